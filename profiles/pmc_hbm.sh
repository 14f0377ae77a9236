#!/bin/bash
# HBM traffic of every kernel of the default bench workload: FETCH_SIZE and WRITE_SIZE in separate passes
# (MI355X_MICROARCH.md: the TCC counters do not fit one pass; no --kernel-trace/--stats next to --pmc).
# Writes gpurun_out/hbm/traffic.json: per-kernel per-launch means in the counter's own unit (KiB) plus the bytes
# after the guide's gfx950 correction (FETCH_SIZE reports half of a wide coalesced read: x2; WRITE_SIZE uncalibrated,
# taken as reported).  bench.py reads profiles/hbm_traffic.json (a committed copy) for its roofline.traffic field.
export TMPDIR=/tmp
rm -rf gpurun_out/hbm; mkdir -p gpurun_out/hbm
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -d gpurun_out/hbm/$c -o p --output-format csv -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/hbm/$c.log 2>&1
done
python - <<'PY'
import csv, glob, collections, json, hashlib
LIBSHA = hashlib.sha256(open('tsfresh_amd/libtsfresh_amd.so', 'rb').read()).hexdigest()[:16]
out = collections.defaultdict(dict)
for f in sorted(glob.glob("gpurun_out/hbm/*/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        out[k][c] = {"launches": len(v), "mean": sum(v) / len(v)}
for k, v in out.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        v["hbm_bytes_per_launch"] = 2.0 * v["FETCH_SIZE"]["mean"] * 1024.0 + v["WRITE_SIZE"]["mean"] * 1024.0
line = json.loads([l for l in open("gpurun_out/hbm/FETCH_SIZE.log") if l.startswith('{"metric"')][-1])
doc = {"workload": line["config"], "units": "FETCH_SIZE / WRITE_SIZE means in KiB per launch as rocprofv3 reports them; "
       "hbm_bytes_per_launch = 2 * FETCH_SIZE + WRITE_SIZE in bytes (gfx950 read correction)", "kernels": out}
doc["lib_sha16"] = LIBSHA   # bench.py replays this document only for the library it was measured on
json.dump(doc, open("gpurun_out/hbm/traffic.json", "w"), indent=1)
for k, v in out.items():
    print(k, {c: (round(x["mean"], 1) if isinstance(x, dict) else round(x / 1e6, 1)) for c, x in v.items()})
PY
