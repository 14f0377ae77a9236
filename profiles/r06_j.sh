#!/bin/bash
O=gpurun_out/r06j; mkdir -p $O
timeout 900 python -m pytest tests/test_seq.py -x -q -m gpu > $O/seq.log 2>&1; tail -3 $O/seq.log
for shape in "100000 1024 comprehensive" "125000 256 comprehensive"; do
  set -- $shape
  python bench.py --n-series $1 --length $2 --params $3 --steps 5 --warmup 1 --no-cpu-baseline --no-e2e > $O/bench$2.json 2>$O/err$2.log
  python -c "
import json,sys;d=json.loads(open('$O/bench$2.json').read().strip().split('\n')[-1]);print('$2',round(d['ms_per_step'],3),{k:round(v,3) for k,v in d['kernel_ms'].items()})"
done
python bench.py --n-series 5000 --ragged 4096:8192 --params efficient --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > $O/bench_cfg4.json 2>$O/err_cfg4.log
python -c "
import json,sys;d=json.loads(open('$O/bench_cfg4.json').read().strip().split('\n')[-1]);print('cfg4',round(d['ms_per_step'],3),{k:round(v,3) for k,v in d['kernel_ms'].items()})"
