#!/bin/bash
# round 4, call M: LDS-staged k_cwt_gemm: parity tests, step, matrix-pipe counters
export TMPDIR=/tmp
O=gpurun_out/r04_m; rm -rf $O; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -x -k "golden or frames or config or ragged or every_cell or longer_than" > $O/pytest_some.log 2>&1; echo "pytest rc=$?" >> $O/pytest_some.log; tail -3 $O/pytest_some.log
q() { timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-e2e $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2), {k: round(v,3) for k,v in d['kernel_ms'].items()}, d.get('parity_sample'))"; }
{ q "headline"; q "256" "--n-series 125000 --length 256"; } > $O/quick.txt 2>&1; cat $O/quick.txt
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-e2e > $O/pmc.log 2>&1
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r04_m/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_cwt_gemm" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
print("k_cwt_gemm counters (mean per launch):", m)
if "GRBM_GUI_ACTIVE" in m:
    print("MFMA busy = %.3f" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)))
PY
rm -rf $O/pmc
