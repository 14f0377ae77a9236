#!/bin/bash
# A/B of library builds on the headline bench: bash profiles/ab.sh libA.so libB.so ...   (paths relative to tsfresh_amd/)
export TMPDIR=/tmp
mkdir -p gpurun_out/ab
for lib in "$@"; do
  TSFA_LIB=$PWD/tsfresh_amd/$lib timeout 600 python bench.py --no-cpu-baseline --no-e2e --steps 4 --warmup 1 2>/dev/null | tail -1 > gpurun_out/ab/$lib.json
  python - <<PY
import json
d = json.load(open("gpurun_out/ab/$lib.json"))
print("$lib", round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in d["kernel_ms"].items()}, d.get("parity_sample"))
PY
done
