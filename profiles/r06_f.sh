#!/bin/bash
O=gpurun_out/r06f; mkdir -p $O
timeout 1200 python -m pytest tests/test_entropy_hbits.py -x -q > $O/hbits.log 2>&1; tail -4 $O/hbits.log
for shape in "2000 16384" "5000 8192"; do
  set -- $shape
  python bench.py --n-series $1 --length $2 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $O/bench$2.json 2>$O/err$2.log
  python -c "
import json,sys;d=json.loads(open('$O/bench$2.json').read().strip().split('\n')[-1]);print('$2',round(d['ms_per_step'],3),{k:round(v,3) for k,v in d['kernel_ms'].items()})"
done
