#!/bin/bash
# round 3: gpu tests + the long-series entropy measurement + configs 2 / 4 / 5 lines
export TMPDIR=/tmp
O=gpurun_out/r03_c
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
run() { timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', '|', round(d['ms_per_step'],2), 'ms |', round(d['value']), 'series/s |', d.get('parity_sample'), {k: round(v,2) for k,v in d['kernel_ms'].items()})"; }
{
run --n-series 10000 --length 4096
run --n-series 2000 --length 16384
run --n-series 10000 --length 2048
run --n-series 10000 --length 1024 --params efficient
run --n-series 125000 --length 256
run --n-series 5000 --ragged 4096:8192 --params efficient
run --params minimal
} > $O/shapes.txt 2>&1
cat $O/shapes.txt
