#!/bin/bash
# Issue-side counters of every kernel of the default bench workload (separate --pmc passes, no tracing next to them):
# how busy the VALU / LDS / scalar pipes are per family.  Writes gpurun_out/pmc_issue/summary.md
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_issue; mkdir -p gpurun_out/pmc_issue
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set -d gpurun_out/pmc_issue/p$i -o p --output-format csv -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/pmc_issue/p$i.log 2>&1
done
python - <<'PY' > gpurun_out/pmc_issue/summary.md
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmc_issue/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        if k.startswith("k_"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
cols = ["GRBM_GUI_ACTIVE", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU",
        "SQ_ACTIVE_INST_SCA", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_ANY",
        "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_SMEM"]
print("per-launch means (rocprofv3 --pmc, bench.py default workload); SQ_* cycle counters are summed over all SIMDs/waves in quad-cycles\n")
print("| kernel | " + " | ".join(cols) + " | VALU busy = ACTIVE_INST_VALU*4 / (GUI_ACTIVE/8 XCDs * 1024 SIMDs) |")
print("|---|" + "---|" * (len(cols) + 1))
for k, v in agg.items():
    m = {c: (sum(v[c]) / len(v[c]) if v.get(c) else float("nan")) for c in cols}
    busy = m["SQ_ACTIVE_INST_VALU"] * 4.0 / (m["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)  # GUI_ACTIVE is summed over the 8 XCDs if m["GRBM_GUI_ACTIVE"] == m["GRBM_GUI_ACTIVE"] else float("nan")
    print("| %s | " % k + " | ".join("%.4g" % m[c] for c in cols) + " | %.3f |" % busy)
PY
cat gpurun_out/pmc_issue/summary.md
