#!/bin/bash
# k_entropy_bits<.,3>: two rounds' tasks in registers, column parts built once for both (2049 .. 4096 samples)
O=gpurun_out/r06v; mkdir -p $O
run() { # name, args...
  name=$1; shift
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e "$@" > $O/$name.json 2>$O/$name.err
  python - <<PY
import json
d=json.loads(open('$O/$name.json').read().strip().split('\n')[-1])
print('$name', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms'].items()}, d.get('parity_sample'))
PY
}
run c4096 --n-series 10000 --length 4096
run c3000 --n-series 10000 --ragged 2049:4096
run c2048 --n-series 20000 --ragged 1025:2048
run h1024 --n-series 100000 --length 1024
timeout 1500 python -m pytest tests/test_query_similarity.py tests/test_gpu_parity.py tests/test_entropy_hbits.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
TSFA_FUZZ_MAXLENS=300,2100,3000,4096,4000 timeout 1200 python profiles/fuzz_parity.py 16 31 > $O/fuzz_long.log 2>&1; tail -1 $O/fuzz_long.log
grep -h "mismatches [1-9]\|UNWRITTEN" $O/*.log | cut -c1-400 | head -20
