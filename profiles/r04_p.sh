#!/bin/bash
# round 4, call P: permutation_entropy for all dimensions of one stride from one sweep; parity + steps
export TMPDIR=/tmp
O=gpurun_out/r04_p; rm -rf $O; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -x -k "perm or golden or config or ragged or shared or known or frames" > $O/pytest_some.log 2>&1; echo "pytest rc=$?" >> $O/pytest_some.log; tail -3 $O/pytest_some.log
q() { timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-e2e $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms'].items()}, d.get('parity_sample'))"; }
{ q "headline"; q "256" "--n-series 125000 --length 256"; q "4096" "--n-series 10000 --length 4096"; } > $O/quick.txt 2>&1; cat $O/quick.txt
