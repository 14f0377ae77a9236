#!/bin/bash
# quick GPU check while optimising: (TESTS=1: the gpu parity tests) + per-kernel times of the default bench workload
#   gpurun -- 'TESTS=1 bash profiles/quick_bench.sh'
export TMPDIR=/tmp
mkdir -p gpurun_out/exp
{
if [ -n "$TESTS" ]; then timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3; fi
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms'].items()})"
} > gpurun_out/exp/q.txt 2>&1
cat gpurun_out/exp/q.txt
