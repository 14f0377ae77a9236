#!/bin/bash
# round 6, the measurement set from ONE build: gpu tests, the bench line (with cpu_baseline on every physical core), rocprofv3
# kernel stats of the same command, issue / HBM / I-cache counters (documents stamped with the library's build id), the other
# BASELINE configs as bench lines.   usage: bash profiles/r05_final.sh TAG [notests]
TAG=${1:-r06_z}
export TMPDIR=/tmp
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
sha256sum tsfresh_amd/libtsfresh_amd.so | cut -c1-16 > $O/lib_sha16.txt
if [ "$2" != "notests" ]; then
  TSFA_PARITY_SKIPS_MD=$O/parity_skips.md timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
fi
bash profiles/pmc_issue.sh > $O/pmc_issue.log 2>&1; cp gpurun_out/pmc_issue/summary.md $O/pmc_issue.md; cp gpurun_out/pmc_issue/valu_issue.json $O/valu_issue.json; tail -3 $O/pmc_issue.md
bash profiles/pmc_hbm.sh > $O/pmc_hbm.log 2>&1; cp gpurun_out/hbm/traffic.json $O/hbm_traffic.json 2>/dev/null
# the bench line reads the counter documents of THIS build (profiles/ on the box is the snapshot's: stage them there)
cp $O/valu_issue.json profiles/valu_issue.json; cp $O/hbm_traffic.json profiles/hbm_traffic.json
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json; echo
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python bench.py --no-cpu-baseline --no-e2e > $O/prof_bench.json 2> $O/prof.err
DB=$(ls $O/prof/*/*.db $O/prof/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python profiles/summarize_rocpd.py $DB "$TAG: rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-e2e" > $O/kernel_stats.md && rm -f $DB
head -20 $O/kernel_stats.md
for cfg in "--walk" "--n-series 10000 --length 1024 --params efficient" "--n-series 125000 --length 256" "--n-series 5000 --ragged 4096:8192 --params efficient" "--params minimal --steps 20 --warmup 3" "--n-series 10000 --length 4096" "--n-series 20000 --length 1000"; do
  timeout 600 python bench.py --no-cpu-baseline --no-e2e $cfg 2>/dev/null | tail -1 >> $O/configs.jsonl
done
python - <<PY
import json
for l in open("$O/configs.jsonl"):
    d = json.loads(l)
    print(d["config"]["workload"][:90], "|", round(d["ms_per_step"], 3), "ms |", round(d["value"]), "series/s | roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"], 4), d.get("parity_sample"))
PY
rm -rf $O/prof gpurun_out/hbm/*/
# device fuzz of this build: standard + random parameter sets + extreme magnitudes
timeout 900 python profiles/fuzz_parity.py 40 4242 > $O/fuzz_std.log 2>&1; tail -1 $O/fuzz_std.log
TSFA_FUZZ_EXTREME=1 timeout 600 python profiles/fuzz_parity.py 25 777 > $O/fuzz_extreme.log 2>&1; tail -1 $O/fuzz_extreme.log
grep -h "mismatches [1-9]\|UNWRITTEN" $O/fuzz_*.log | cut -c1-300 | head
