#!/bin/bash
# round 4, last call (5 GPU-minutes left): the build with k_perm.  Call T ran the golden / ragged / every_cell /
# longer_than_lds gpu tests on the same device code; this runs the REST of the gpu suite (+ the permutation_entropy
# tests), the rocprofv3 kernel stats of the bench command, and the bench line.  usage: bash profiles/r04_final2.sh TAG
TAG=${1:-r04_zz}
export TMPDIR=/tmp
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
TSFA_PARITY_SKIPS_MD=$O/parity_skips.md timeout 200 python -m pytest tests -m gpu -q -k "perm or not (golden or ragged or every_cell or longer_than_lds)" > $O/pytest_gpu_rest.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_rest.log; tail -3 $O/pytest_gpu_rest.log
timeout 90 rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python bench.py --no-cpu-baseline --no-e2e > $O/prof_bench.json 2> $O/prof.err
DB=$(ls $O/prof/*/*.db $O/prof/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python profiles/summarize_rocpd.py $DB "$TAG: rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-e2e" > $O/kernel_stats.md && rm -f $DB
cat $O/kernel_stats.md; rm -rf $O/prof
timeout 150 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json; echo
