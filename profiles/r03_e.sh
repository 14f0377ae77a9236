#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03_e
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "entropy or long or lds or config5 or structured" 2>&1 | tail -4
run() { timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', '|', round(d['ms_per_step'],2), 'ms |', d.get('parity_sample'), {k: round(v,2) for k,v in d['kernel_ms'].items()})"; }
{
run --n-series 10000 --length 4096
run --n-series 10000 --length 2048
run --n-series 10000 --length 3000
TSFA_ENT_PAIRS=1 run --n-series 10000 --length 3000
} > $O/long_entropy.txt 2>&1
cat $O/long_entropy.txt
