#!/bin/bash
# change_quantiles: corridor edges once (lane = edge), means and variances from ONE pass per sweep (shifted sums)
O=gpurun_out/r06q; mkdir -p $O
run() { # name, args...
  name=$1; shift
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e "$@" > $O/$name.json 2>$O/$name.err
  python - <<PY
import json
d=json.loads(open('$O/$name.json').read().strip().split('\n')[-1])
print('$name', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms'].items()})
PY
}
run h1024 --n-series 100000 --length 1024
run h256 --n-series 125000 --length 256
run cfg4 --n-series 5000 --ragged 4096:8192 --params efficient
run cfg1 --n-series 10000 --length 1024 --params efficient
timeout 1500 python -m pytest tests/test_query_similarity.py tests/test_gpu_parity.py tests/test_frames.py tests/test_nonfinite.py tests/test_param_beyond.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 900 python profiles/fuzz_parity.py 30 977 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
grep -h "mismatches [1-9]\|UNWRITTEN" $O/*.log | cut -c1-400 | head -20
