#!/bin/bash
# One GPU-box pass: gpu tests, bench line, rocprofv3 kernel stats of the same bench command, HBM counters, boundary timings.
# usage: bash profiles/run_profile.sh TAG   -> gpurun_out/TAG/*
TAG=${1:-run}
export TMPDIR=/tmp
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
if [ -z "$SKIP_TESTS" ]; then timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log; fi
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python bench.py --no-cpu-baseline --no-e2e > $O/prof_bench.json 2> $O/prof.err
DB=$(ls $O/prof/*/*.db $O/prof/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python profiles/summarize_rocpd.py $DB "$TAG: rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-e2e" > $O/kernel_stats.md && rm -f $DB
cat $O/kernel_stats.md
if [ -z "$SKIP_PMC" ]; then bash profiles/pmc_hbm.sh > $O/pmc_hbm.log 2>&1; cp gpurun_out/hbm/traffic.json $O/hbm_traffic.json 2>/dev/null; tail -12 $O/pmc_hbm.log; fi
if [ -z "$SKIP_E2E" ]; then timeout 600 python profiles/e2e_timing.py > $O/e2e.json 2> $O/e2e.err; tail -3 $O/e2e.json; fi
rm -rf $O/prof gpurun_out/hbm/*/
