#!/bin/bash
# Issue-side counters of every kernel of the default bench workload (separate --pmc passes, no tracing next to them):
# how busy the VALU / LDS / scalar pipes are per family, and the MFMA pipe for k_cwt_gemm.
# Writes gpurun_out/pmc_issue/summary.md and gpurun_out/pmc_issue/valu_issue.json (bench.py reads the committed copy
# profiles/valu_issue.json for its roofline.valu field).
# PMC_BENCH_ARGS: other workloads (round 6: "--n-series 125000 --length 256", the configs[3] shard)
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_issue; mkdir -p gpurun_out/pmc_issue
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F64"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set -d gpurun_out/pmc_issue/p$i -o p --output-format csv -- python bench.py $PMC_BENCH_ARGS --steps 1 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/pmc_issue/p$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections, json, hashlib
LIBSHA = hashlib.sha256(open('tsfresh_amd/libtsfresh_amd.so', 'rb').read()).hexdigest()[:16]
# one dict per kernel NAME and counter; every pass is aggregated by the kernel name of ITS OWN rows (round 2's table
# printed k_sort's GRBM_GUI_ACTIVE in the k_trend row: the template arguments were cut at the first '<' of 'void k<..>')
def kname(raw):
    raw = raw.strip()
    if raw.startswith("void "):
        raw = raw[5:]
    return raw.split("<")[0].split("(")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmc_issue/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = kname(r["Kernel_Name"])
        if k.startswith("k_") or k.startswith("kl_"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
cols = ["GRBM_GUI_ACTIVE", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU",
        "SQ_ACTIVE_INST_SCA", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT",
        "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F64"]
line = json.loads([l for l in open("gpurun_out/pmc_issue/p1.log") if l.startswith('{"metric"')][-1])
SIMDS = 1024.0
# cycles per wave64 VALU instruction per kernel: the kernel's instruction mix weighted by the issue rates measured on the box
# (profiles/lab/issue_rates.hip -> profiles/r04_issue_rates.jsonl -> profiles/lab/valu_mix.py -> profiles/valu_mix.json);
# the shader clock is the one measured in the same microbenchmark (s_memtime / s_memrealtime)
try:
    MIX = json.load(open("profiles/valu_mix.json"))
except (OSError, ValueError):
    MIX = {"kernels": {}, "shader_ghz_measured": 2.4}
CLOCK = MIX["shader_ghz_measured"] * 1e9
def cyc_per_inst(k):
    return MIX["kernels"].get(k, {}).get("cycles_per_valu_inst", 4.0)
doc = {"workload": line["config"], "units": "means per launch (= per step) of each kernel; SQ_* instruction counters are "
       "wave-instructions, *ACTIVE* / *BUSY* counters quad-cycles summed over the SIMDs; GRBM_GUI_ACTIVE cycles summed over the 8 XCDs",
       "kernels": {}, "step": None}
with open("gpurun_out/pmc_issue/summary.md", "w") as out:
    out.write("means per launch of each kernel = per step (rocprofv3 --pmc, bench.py default workload: 4 launches per pass, one pass per counter set)\n\n")
    out.write("| kernel | launches | " + " | ".join(cols) + " | VALU busy | ms at full VALU issue | MFMA busy |\n")
    out.write("|---|---|" + "---|" * (len(cols) + 3) + "\n")
    tot_insts = tot_active = tot_gui = tot_full = 0.0
    for k, v in sorted(agg.items()):
        n_launch = max(len(x) for x in v.values())
        m = {c: (sum(v[c]) / len(v[c]) if v.get(c) else float("nan")) for c in cols}   # mean per launch
        gui = m["GRBM_GUI_ACTIVE"] / 8.0
        busy = m["SQ_ACTIVE_INST_VALU"] * 4.0 / (gui * SIMDS) if gui == gui and gui > 0 else float("nan")
        full_ms = m["SQ_INSTS_VALU"] * cyc_per_inst(k) / (SIMDS * CLOCK) * 1e3
        # MfmaUtil as rocprofv3's gfx94x formula: MFMA-busy cycles (counted in cycles per SIMD) over all SIMD-cycles of the kernel
        mfma = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * SIMDS) if gui == gui and gui > 0 else float("nan")
        out.write("| %s | %d | " % (k, n_launch) + " | ".join("%.4g" % m[c] for c in cols) + " | %.3f | %.3f | %.4f |\n" % (busy, full_ms, mfma))
        doc["kernels"][k] = {"launches": n_launch, "insts_valu": m["SQ_INSTS_VALU"], "active_inst_valu_quadcycles": m["SQ_ACTIVE_INST_VALU"],
                             "gui_active_cycles_per_xcd": gui, "valu_busy": busy, "ms_at_full_issue": full_ms,
                             "cycles_per_valu_inst": cyc_per_inst(k),
                             "mfma_busy_quadcycles": m["SQ_VALU_MFMA_BUSY_CYCLES"], "mfma_mops_f64": m["SQ_INSTS_VALU_MFMA_MOPS_F64"],
                             "mfma_busy_frac": mfma}
        if m["SQ_INSTS_VALU"] == m["SQ_INSTS_VALU"]:
            tot_insts += m["SQ_INSTS_VALU"]; tot_active += m["SQ_ACTIVE_INST_VALU"]; tot_gui += gui; tot_full += full_ms
    doc["step"] = {"insts": tot_insts, "busy": tot_active * 4.0 / (tot_gui * SIMDS) if tot_gui else None,
                   "ms_at_full_issue": tot_full,
                   "basis": "SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU / GRBM_GUI_ACTIVE summed over the kernels of one step; full issue of a "
                            "kernel = its VALU wave-instructions x the cycles per instruction of ITS instruction mix (3.5 .. 4.1: "
                            "profiles/valu_mix.json, weights measured on the box by profiles/lab/issue_rates.hip -- 2.25 cycles for "
                            "and/or/xor/add/sub/lshr/mov/f32 add-mul, 4.05 for every float64, VOP3, DPP, compare, min/max and "
                            "shift-left form) on 1024 SIMDs at the measured %.2f GHz" % (CLOCK / 1e9),
                   "source": "profiles/valu_issue.json (builder-measured rocprofv3 --pmc passes of this command, profiles/pmc_issue.sh; "
                             "replayed, not re-measured in this run)"}
    out.write("\nstep: %.4g VALU wave-instructions, %.3f ms at full issue, time-weighted VALU busy %.3f\n" % (
        tot_insts, doc["step"]["ms_at_full_issue"], doc["step"]["busy"] or float("nan")))
doc["lib_sha16"] = LIBSHA   # bench.py replays this document only for the library it was measured on
json.dump(doc, open("gpurun_out/pmc_issue/valu_issue.json", "w"), indent=1)
print(open("gpurun_out/pmc_issue/summary.md").read())
PY
rm -rf gpurun_out/pmc_issue/p*/
