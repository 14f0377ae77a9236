#!/bin/bash
# round 4, call J: shared per-series statistics: gpu suite + the steps
export TMPDIR=/tmp
O=gpurun_out/r04_j; rm -rf $O; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
q() { timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms'].items()}, d.get('parity_sample'))"; }
{ q "headline"; TSFA_NO_STATS_SHARE=1 q "headline-noshare"; q "walk" --walk; q "256" "--n-series 125000 --length 256"; q "4096" "--n-series 10000 --length 4096"; } > $O/quick.txt 2>&1; cat $O/quick.txt
