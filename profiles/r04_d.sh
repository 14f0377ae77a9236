#!/bin/bash
# round 4, call D: the wavefront form of k_seq -- parity on the device, then the headline step with S = 1 / 2 / 4 series
# per wavefront against the workgroup form
export TMPDIR=/tmp
O=gpurun_out/r04_d; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_seq.py -m gpu -x -q > $O/pytest_seq.log 2>&1; tail -3 $O/pytest_seq.log
q() { timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms'].items()}, d.get('parity_sample'))"; }
{
TSFA_SEQ_WG=1 q "workgroup form  "
TSFA_SEQ_S=1 q "wave form S=1   "
TSFA_SEQ_S=2 q "wave form S=2   "
TSFA_SEQ_S=4 q "wave form S=4   "
TSFA_SEQ_S=2 q "S=2 walk        " --walk
TSFA_SEQ_WG=1 q "wg walk         " --walk
TSFA_SEQ_S=2 q "S=2 256         " "--n-series 125000 --length 256"
TSFA_SEQ_S=4 q "S=4 256         " "--n-series 125000 --length 256"
TSFA_SEQ_WG=1 q "wg 256          " "--n-series 125000 --length 256"
} > $O/seq_ab.txt 2>&1
cat $O/seq_ab.txt
