#!/bin/bash
# round 6, call d: new tests (composite upload-once, forced exchange over RCCL, strong+ragged bench, row form) and the 256 shape with
# two register budgets of the row kernels
O=gpurun_out/r06d; mkdir -p $O
timeout 900 python -m pytest tests/test_row_form.py tests/test_composite_plan.py tests/test_distributed_rccl.py tests/test_bench_launch.py tests/test_nonfinite.py -x -q -m gpu > $O/tests.log 2>&1; tail -5 $O/tests.log
for lib in libtsfresh_amd.so libtsfresh_amd_b.so; do
  TSFA_LIB=$PWD/tsfresh_amd/$lib python bench.py --n-series 125000 --length 256 --steps 5 --warmup 1 --no-cpu-baseline --no-e2e > $O/bench256_$lib.json 2>$O/err_$lib.log
  python -c "
import json,sys;d=json.loads(open('$O/bench256_$lib.json').read().strip().split('\n')[-1]);print('$lib',round(d['ms_per_step'],3),{k:round(v,3) for k,v in d['kernel_ms'].items()})"
done
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e > $O/bench1024.json 2>$O/err1024.log
python -c "
import json,sys;d=json.loads(open('$O/bench1024.json').read().strip().split('\n')[-1]);print('1024',round(d['ms_per_step'],3),{k:round(v,3) for k,v in d['kernel_ms'].items()})"
