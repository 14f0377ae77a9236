#!/bin/bash
# Instruction-fetch counters per kernel of the default bench workload (separate --pmc passes, no tracing next to them).
# Why: round 4 measured that code executed ONCE per series can cost far more than its instruction count -- an out-of-line
# call to a 50 KB function inside k_sort cost 40 k cycles per series around its phases, the unchanged one-dimension-at-a-time
# path in the same (larger) kernel went from 85 k to 116 k cycles (profiles/r04_pqrs_perm_in_sort.md) -- and k_sort / k_ar /
# k_basic are each 100+ KB of straight-line code against a 64 KB instruction cache shared by two CUs.  If the miss rate
# and SQ_IFETCH_LEVEL (fetches in flight x cycles -> latency per fetch) confirm it, the next split is chosen by code size.
# NOT RUN YET (written when the round's GPU minutes were spent).  ~1 GPU-minute: three passes of one step each.
#   bash profiles/pmc_icache.sh   ->  gpurun_out/pmc_icache/summary.md
export TMPDIR=/tmp
O=gpurun_out/pmc_icache; rm -rf $O; mkdir -p $O
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d $O/p$i -o p --output-format csv -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-e2e > $O/p$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections
def kname(raw):
    raw = raw.strip()
    if raw.startswith("void "):
        raw = raw[5:]
    return raw.split("<")[0].split("(")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmc_icache/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = kname(r["Kernel_Name"])
        if k.startswith("k_") or k.startswith("kl_"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
def tot(k, c):
    v = agg[k].get(c)
    return sum(v) / max(1, len(v)) if v else float("nan")   # per launch (the warm-up and the timed step)
rows = []
for k in sorted(agg):
    req, hit, miss, dup = (tot(k, c) for c in ("SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQC_ICACHE_MISSES_DUPLICATE"))
    fetch, level, wcyc = tot(k, "SQ_IFETCH"), tot(k, "SQ_IFETCH_LEVEL"), tot(k, "SQ_WAVE_CYCLES")
    wait_inst, valu = tot(k, "SQ_WAIT_INST_ANY"), tot(k, "SQ_INSTS_VALU")
    rows.append((k, req, miss / req if req else float("nan"), dup / req if req else float("nan"), fetch,
                 level / fetch if fetch else float("nan"), wait_inst / wcyc if wcyc else float("nan"), valu))
with open("gpurun_out/pmc_icache/summary.md", "w") as f:
    f.write("| kernel | icache requests | miss rate | duplicate-miss rate | SQ_IFETCH | cycles per fetch (LEVEL / IFETCH) | SQ_WAIT_INST_ANY / wave cycles | VALU insts |\n|---|---|---|---|---|---|---|---|\n")
    for r in rows:
        f.write("| `%s` | %.3g | %.3f | %.3f | %.3g | %.1f | %.3f | %.3g |\n" % r)
print(open("gpurun_out/pmc_icache/summary.md").read())
PY
rm -rf $O/p*/
