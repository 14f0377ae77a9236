#!/bin/bash
# round 4, call E: full gpu suite on the tree (closed-form LZ edges, constant tables, no pre-fill), the headline step, the
# MFMA DFT experiment (time + matrix-pipe counters)
export TMPDIR=/tmp
O=gpurun_out/r04_e; rm -rf $O; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
q() { timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms'].items()}, d.get('parity_sample'))"; }
{ q "headline"; q "walk" --walk; q "256" "--n-series 125000 --length 256"; } > $O/quick.txt 2>&1; cat $O/quick.txt
timeout 120 profiles/lab/build/mfma_dft > $O/mfma_dft.json 2> $O/mfma_dft.err; cat $O/mfma_dft.json
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/mfma_pmc -o p -- profiles/lab/build/mfma_dft 20000 > $O/mfma_pmc.log 2>&1
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r04_e/mfma_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_dft_mfma" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
print("mfma_dft counters (mean per launch, 20000 series):", m)
if "GRBM_GUI_ACTIVE" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
    gui = m["GRBM_GUI_ACTIVE"] / 8.0
    print("MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) = %.3f" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024.0)))
PY
rm -rf $O/mfma_pmc
