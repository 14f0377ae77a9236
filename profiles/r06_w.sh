#!/bin/bash
# 1025 .. 2048 samples through the 48-byte-entry sweep (one workgroup per CU, 6 column parts instead of 22)
O=gpurun_out/r06w; mkdir -p $O
run() { # name, args...
  name=$1; shift
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e "$@" > $O/$name.json 2>$O/$name.err
  python - <<PY
import json
d=json.loads(open('$O/$name.json').read().strip().split('\n')[-1])
print('$name', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms'].items()}, d.get('parity_sample'))
PY
}
run c2048 --n-series 20000 --ragged 1025:2048
run f2048 --n-series 20000 --length 2048
run f1500 --n-series 20000 --length 1500
run c4096 --n-series 10000 --length 4096
run h1024 --n-series 100000 --length 1024
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_entropy_hbits.py tests/test_param_sweep.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
TSFA_FUZZ_MAXLENS=300,1100,1500,2048,2047,3000 timeout 1200 python profiles/fuzz_parity.py 20 57 > $O/fuzz_long.log 2>&1; tail -1 $O/fuzz_long.log
grep -h "mismatches [1-9]\|UNWRITTEN" $O/*.log | cut -c1-400 | head -20
