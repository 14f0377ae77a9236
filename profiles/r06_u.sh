#!/bin/bash
# chirp-z transform of the even lengths in ONE LDS tile (the modulated series never goes through the scratch)
O=gpurun_out/r06u; mkdir -p $O
run() { # name, args...
  name=$1; shift
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e "$@" > $O/$name.json 2>$O/$name.err
  python - <<PY
import json
d=json.loads(open('$O/$name.json').read().strip().split('\n')[-1])
print('$name', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms'].items()}, d.get('parity_sample'))
PY
}
run cfg4 --n-series 5000 --ragged 4096:8192 --params efficient
run c2500 --n-series 10000 --ragged 2049:4096 --params efficient
run h1000 --n-series 20000 --length 1000
run c6000 --n-series 5000 --length 6000 --params efficient
timeout 1500 python -m pytest tests/test_spectral_chirpz.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
TSFA_FUZZ_MAXLENS=300,3000,5000,7000,8192 timeout 1200 python profiles/fuzz_parity.py 16 99 > $O/fuzz_long.log 2>&1; tail -1 $O/fuzz_long.log
grep -h "mismatches [1-9]\|UNWRITTEN" $O/*.log | cut -c1-400 | head -20
