#!/bin/bash
# lempel_ziv symbol rows in HBM (k_seq<T, true>) + exact-size phrase hash: parity on the device, then A/B at three shapes
O=gpurun_out/r06o; mkdir -p $O
timeout 900 python -m pytest tests/test_seq.py -x -q -m gpu > $O/seq.log 2>&1; tail -3 $O/seq.log
run() { # name, args...
  name=$1; shift
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e "$@" > $O/$name.json 2>$O/$name.err
  python - <<PY
import json
d=json.loads(open('$O/$name.json').read().strip().split('\n')[-1])
print('$name', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms'].items()})
PY
}
for rows in 0 1 -1; do
  run cfg4_rows$rows --n-series 5000 --ragged 4096:8192 --params efficient --plan-option seq_rows=$rows
  run h1024_rows$rows --n-series 100000 --length 1024 --plan-option seq_rows=$rows
  run h256_rows$rows --n-series 125000 --length 256 --plan-option seq_rows=$rows
done
