#!/bin/bash
# round 4, the measurement set from ONE build: gpu tests, the bench line (with cpu_baseline), rocprofv3 kernel stats of
# the same command, issue / MFMA / HBM counters, the other BASELINE configs as bench lines.   usage: bash profiles/r04_final.sh TAG
TAG=${1:-r04_z}
export TMPDIR=/tmp
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
TSFA_PARITY_SKIPS_MD=$O/parity_skips.md timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1200 $O/bench.json; echo
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python bench.py --no-cpu-baseline --no-e2e > $O/prof_bench.json 2> $O/prof.err
DB=$(ls $O/prof/*/*.db $O/prof/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python profiles/summarize_rocpd.py $DB "$TAG: rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-e2e" > $O/kernel_stats.md && rm -f $DB
cat $O/kernel_stats.md
bash profiles/pmc_issue.sh > $O/pmc_issue.log 2>&1; cp gpurun_out/pmc_issue/summary.md $O/pmc_issue.md; cp gpurun_out/pmc_issue/valu_issue.json $O/valu_issue.json; tail -5 $O/pmc_issue.md
bash profiles/pmc_hbm.sh > $O/pmc_hbm.log 2>&1; cp gpurun_out/hbm/traffic.json $O/hbm_traffic.json 2>/dev/null; tail -12 $O/pmc_hbm.log
for cfg in "--walk" "--n-series 10000 --length 1024 --params efficient" "--n-series 125000 --length 256" "--n-series 5000 --ragged 4096:8192 --params efficient" "--params minimal --steps 20 --warmup 3" "--n-series 10000 --length 4096"; do
  timeout 600 python bench.py --no-cpu-baseline --no-e2e $cfg 2>/dev/null | tail -1 >> $O/configs.jsonl
done
python - <<PY
import json
for l in open("$O/configs.jsonl"):
    d = json.loads(l)
    print(d["config"]["workload"][:90], "|", round(d["ms_per_step"], 3), "ms |", round(d["value"]), "series/s | roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"], 4), d.get("parity_sample"))
PY
rm -rf $O/prof gpurun_out/hbm/*/
# parity fuzz on the device with the final build: random calculator subsets x random ragged batches against the oracle
( timeout 900 python profiles/fuzz_parity.py 40 4101; timeout 600 python profiles/fuzz_parity.py 30 4102 ) > $O/fuzz_gpu.log 2>&1; tail -3 $O/fuzz_gpu.log
# the long-series check of VERDICT r3 #8: 2 000 x 16 384 Comprehensive (the run that "did not finish in 600 s")
( time timeout 900 python bench.py --n-series 2000 --length 16384 --steps 1 --warmup 0 --no-cpu-baseline --no-e2e ) > $O/long_2000x16384.json 2> $O/long_2000x16384.err; tail -c 600 $O/long_2000x16384.json; tail -4 $O/long_2000x16384.err
