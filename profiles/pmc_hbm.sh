#!/bin/bash
# HBM traffic of every kernel of the bench workload: FETCH_SIZE and WRITE_SIZE in separate passes
# (MI355X_MICROARCH.md: TCC slots).  Writes gpurun_out/hbm/traffic.json (per-launch averages, raw counter units = KiB
# as rocprofv3 reports them; bench.py applies the guide's gfx950 x2 read correction).
export TMPDIR=/tmp
rm -rf gpurun_out/hbm; mkdir -p gpurun_out/hbm
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -d gpurun_out/hbm/$c -o p --output-format csv -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/hbm/$c.log 2>&1
done
python - <<'PY'
import csv, glob, collections, json
out = collections.defaultdict(dict)
for f in sorted(glob.glob("gpurun_out/hbm/*/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        out[k][c] = {"launches": len(v), "mean": sum(v) / len(v)}
json.dump(out, open("gpurun_out/hbm/traffic.json", "w"), indent=1)
for k, v in out.items():
    print(k, {c: round(x["mean"], 1) for c, x in v.items()})
PY
