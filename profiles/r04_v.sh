#!/bin/bash
# round 4, call V (last): k_perm with one atomic per wavefront for the patterns that occur once (build -DTSFA_PE_BALLOT,
# libtsfresh_amd_exp.so) against the plain build; then the rocprofv3 kernel stats of the bench command with the better one
export TMPDIR=/tmp
O=gpurun_out/r04_v; rm -rf $O; mkdir -p $O
EXP=$PWD/tsfresh_amd/libtsfresh_amd_exp.so
TSFA_LIB=$EXP timeout 60 python -m pytest tests -m gpu -q -k "perm or several_devices" > $O/pytest_exp.log 2>&1; RC=$?; echo "pytest rc=$RC" >> $O/pytest_exp.log; tail -2 $O/pytest_exp.log
TSFA_LIB=$EXP timeout 60 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 > $O/bench_exp.json
KP=$(python -c "import json; d=json.load(open('$O/bench_exp.json')); print(round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms'].items()}, d.get('parity_sample'), file=open('$O/quick_exp.txt','w')); print(1 if d['kernel_ms'].get('k_perm', 9) < 0.84 and d.get('parity_sample') == 'ok' else 0)")
cat $O/quick_exp.txt
if [ "$RC" = "0" ] && [ "$KP" = "1" ]; then export TSFA_LIB=$EXP; echo "stats: ballot build" | tee $O/which.txt; else echo "stats: plain build" | tee $O/which.txt; fi
timeout 60 rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python bench.py --no-cpu-baseline --no-e2e > $O/prof_bench.json 2> $O/prof.err
DB=$(ls $O/prof/*/*.db $O/prof/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python profiles/summarize_rocpd.py $DB "r04_v ($(cat $O/which.txt)): rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-e2e" > $O/kernel_stats.md && rm -f $DB
head -16 $O/kernel_stats.md; rm -rf $O/prof
