#!/usr/bin/env python
"""Registers / scratch / occupancy of every kernel as the compiler reports them (no GPU needed):
    python profiles/kernel_resources.py [substring filter]"""
import os
import re
import subprocess
import sys

csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tsfresh_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
       "-Wno-unused-function", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-mllvm", "-disable-machine-licm",
       "-Rpass-analysis=kernel-resource-usage", "-c", "tsfa_kernels.hip", "-o", "/tmp/tsfa_k.o"]
txt = subprocess.run(cmd, cwd=csrc, capture_output=True, text=True).stderr
rows, cur = [], None
for line in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]+\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for r in rows:
    if flt not in r["name"]:
        continue
    print("%-50s VGPR %3d AGPR %3d SGPR %3d scratch %4d occ %d LDS %d" % (
        r["name"][:50], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("SGPRs", -1), r.get("ScratchSize", -1),
        r.get("Occupancy", -1), r.get("LDS Size", -1)))
if "error" in txt:
    print(txt[-3000:])
