#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/exp
run() { timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), {k: round(v,1) for k,v in d['kernel_ms'].items()})"; }
{
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for nt in 64 128 192 256; do echo "ar nt=$nt"; TSFA_NT_3=$nt run; done
} > gpurun_out/exp/log30.txt 2>&1
