#!/bin/bash
# round 4, call S: permutation_entropy from one sweep, results scattered lane = column, one block sum; clocks + parity + steps
export TMPDIR=/tmp
O=gpurun_out/r04_s; rm -rf $O; mkdir -p $O
TSFA_LIB=$PWD/tsfresh_amd/libtsfresh_amd_ticks.so timeout 300 python profiles/phase_ticks.py 2> $O/fused.err | grep -i "permutation" > $O/fused.md; cat $O/fused.md
timeout 600 python -m pytest tests -m gpu -q -x -k "perm or golden" > $O/pytest_some.log 2>&1; echo "pytest rc=$?" >> $O/pytest_some.log; tail -3 $O/pytest_some.log
q() { timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-e2e $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms'].items()}, d.get('parity_sample'))"; }
{ q "headline"; TSFA_NO_PE_FUSED=1 q "headline, one dimension at a time"; } > $O/quick.txt 2>&1; cat $O/quick.txt
