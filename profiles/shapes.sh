#!/bin/bash
# the other BASELINE shapes (Efficient 10k x 1024, Comprehensive 125k x 256, ragged 4096..8192, Minimal): DESIGN.md section 5
export TMPDIR=/tmp
mkdir -p gpurun_out/exp
run() { timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), round(d['value']), {k: round(v,2) for k,v in d['kernel_ms'].items()})"; }
{
echo default; run
echo efficient; run --params efficient --n-series 10000
echo len256; run --length 256 --n-series 125000
echo ragged; run --params efficient --n-series 5000 --ragged 4096:8192
echo minimal; run --params minimal
} > gpurun_out/exp/shapes.txt 2>&1
cat gpurun_out/exp/shapes.txt
