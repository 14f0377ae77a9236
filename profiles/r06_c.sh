#!/bin/bash
# round 6, call c: row-form tests, the extreme-magnitude device fuzz (VERDICT r5 #4c: >= 100 rounds), issue counters of the
# configs[3] shard shape (125 000 x 256) with the row form
O=gpurun_out/r06c; mkdir -p $O
timeout 600 python -m pytest tests/test_row_form.py -x -q > $O/rows.log 2>&1; tail -3 $O/rows.log
TSFA_FUZZ_EXTREME=1 TSFA_FUZZ_MAXLENS=40,300,1024 timeout 1500 python profiles/fuzz_parity.py 100 61 > $O/fuzz_extreme.log 2>&1; grep -c "^round" $O/fuzz_extreme.log; tail -1 $O/fuzz_extreme.log
PMC_BENCH_ARGS="--n-series 125000 --length 256" bash profiles/pmc_issue.sh > $O/pmc.log 2>&1
cp gpurun_out/pmc_issue/summary.md $O/pmc_issue_256.md; cat $O/pmc_issue_256.md | cut -c1-250
