#!/bin/bash
# round 5, call E: why the Ricker convolutions on the matrix cores do not pay -- (1) does v_mfma_f64_16x16x4_f64 run beside
# another wavefront's vector instructions or in their place (profiles/lab/mfma_f64_coissue.hip), (2) k_cwtpeaks alone in both
# forms: time, VALU instructions / busy, matrix-pipe busy
export TMPDIR=/tmp
O=gpurun_out/r05_e; rm -rf $O; mkdir -p $O
timeout 120 profiles/lab/build/mfma_f64_coissue > $O/mfma_f64_coissue.jsonl 2> $O/coissue.err; cat $O/mfma_f64_coissue.jsonl
for v in 0 1; do
  TSFA_NO_CWT_MFMA=$v timeout 120 python profiles/lab/cwtpeaks_only.py >> $O/cwtpeaks_only.jsonl 2>> $O/err.txt
  i=0
  for set in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"; do
    i=$((i+1))
    TSFA_NO_CWT_MFMA=$v timeout 200 rocprofv3 --pmc $set -d $O/p${v}_$i -o p --output-format csv -- python profiles/lab/cwtpeaks_only.py > $O/p${v}_$i.log 2>&1
  done
done
cat $O/cwtpeaks_only.jsonl
python - <<'PY'
import csv, glob, collections
for v in (0, 1):
    agg = collections.defaultdict(list)
    for f in sorted(glob.glob("gpurun_out/r05_e/p%d_*/*counter_collection.csv" % v)):
        for r in csv.DictReader(open(f)):
            if "k_cwtpeaks" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("TSFA_NO_CWT_MFMA=%d" % v, {k: "%.4g" % (sum(x) / len(x)) for k, x in sorted(agg.items())})
PY
rm -rf $O/p*/
