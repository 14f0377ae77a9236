#!/bin/bash
# round 3, last build: rocprofv3 kernel stats of the bench command, issue counters, then the bench line itself (with cpu_baseline / e2e)
TAG=${1:-r03_p}
export TMPDIR=/tmp
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python bench.py --no-cpu-baseline --no-e2e > $O/prof_bench.json 2> $O/prof.err
DB=$(ls $O/prof/*/*.db $O/prof/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python profiles/summarize_rocpd.py $DB "$TAG: rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-e2e" > $O/kernel_stats.md && rm -f $DB
head -14 $O/kernel_stats.md
bash profiles/pmc_issue.sh > $O/pmc_issue.log 2>&1; cp gpurun_out/pmc_issue/summary.md $O/pmc_issue.md; cp gpurun_out/pmc_issue/valu_issue.json $O/valu_issue.json; tail -3 $O/pmc_issue.md
cp $O/valu_issue.json profiles/valu_issue.json
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; head -c 1800 $O/bench.json; echo
rm -rf $O/prof
