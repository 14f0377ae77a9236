#!/usr/bin/env python
"""ms per series of the entropy family (and of the whole Comprehensive step) for series beyond the bit-matrix limit
(VERDICT r3 'Next' #8): -> markdown on stdout.   python profiles/long_entropy.py > gpurun_out/long_entropy.md"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [(4096, 512), (4097, 256), (5000, 256), (8192, 128), (16384, 64), (32768, 16)]


def one(length, n):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-e2e",
           "--n-series", str(n), "--length", str(length)]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=420)
    except subprocess.TimeoutExpired:
        return None
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    return json.loads(lines[-1]) if lines else None


def main():
    print("| samples | series | step ms | ms per series (whole step) | `k_entropy` ms | ms per series (entropy) | parity sample |")
    print("|---|---|---|---|---|---|---|")
    for length, n in CASES:
        d = one(length, n)
        if d is None:
            print("| %d | %d | did not finish in 420 s (2 launches) | | | | |" % (length, n))
            continue
        ke = d["kernel_ms"].get("k_entropy", float("nan"))
        print("| %d | %d | %.1f | %.3f | %.1f | %.3f | %s |" % (length, n, d["ms_per_step"], d["ms_per_step"] / n, ke, ke / n, d.get("parity_sample")))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
