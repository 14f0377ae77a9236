#!/usr/bin/env python
"""SURVEY H2 option (a), reported beside the gate: what the REAL reference returns when the BASELINE dtype (float32
columns) is fed to it as is -- several calculators then run numpy / pandas reductions in float32 -- against the gate of
this repository, the reference's value for x.astype(float64) (oracle/).  Inputs and reference outputs:
tests/golden/ref_frames_{main,conda}.json, cases float32_as_is_* (8 series x 1024: 4 i.i.d., 4 walks).

    python profiles/float32_as_is_report.py  > profiles/r03_float32_as_is.md
"""
import json
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.simplefilter("ignore")
from golden.frame_codec import decode_container, decode_frame  # noqa: E402
from oracle.extract import oracle_matrix  # noqa: E402
from parity import is_integer_feature  # noqa: E402
from tsfresh_amd.feature_extraction import settings  # noqa: E402


def main():
    per_calc = {}
    flips = []
    for f in ("ref_frames_main.json", "ref_frames_conda.json"):
        with open(os.path.join(ROOT, "tests", "golden", f)) as fh:
            cases = json.load(fh)["cases"]
        for case in cases:
            if not case["name"].startswith("float32_as_is"):
                continue
            df = decode_container(case["input"])
            ref32 = decode_frame(case["output"])
            ids = sorted(df["id"].unique())
            series = [df[df.id == i].sort_values("time")["value"].to_numpy() for i in ids]
            assert all(s.dtype == np.float32 for s in series)
            values = np.concatenate(series).astype(np.float64)
            offsets = np.concatenate([[0], np.cumsum([len(s) for s in series])])
            names, gate = oracle_matrix(values, offsets, settings.ComprehensiveFCParameters())
            col = {n: j for j, n in enumerate(names)}
            for c in ref32.columns:
                g = gate[:, col[c]]
                r = ref32[c].to_numpy()
                calc = c.split("__")[1]
                both = np.isfinite(g) & np.isfinite(r)
                if is_integer_feature(c):
                    n_flip = int(np.sum(g[both] != r[both]))
                    if n_flip:
                        flips.append((c, n_flip, len(g)))
                    dev = 0.0
                else:
                    scale = np.maximum(np.abs(g[both]), 1e-12 * max(1.0, float(np.max(np.abs(values)))))
                    dev = float(np.max(np.abs(r[both] - g[both]) / scale)) if both.any() else 0.0
                nan_mismatch = int(np.sum(np.isnan(g) != np.isnan(r)))
                a = per_calc.setdefault(calc, [0.0, 0, 0])
                a[0] = max(a[0], dev)
                a[1] += 1
                a[2] += nan_mismatch
    print("# float32 fed as is (reference) vs the gate (reference on x.astype(float64)) -- 8 series x 1024\n")
    print("| calculator | columns | max relative deviation | NaN mismatches |")
    print("|---|---|---|---|")
    for calc, (dev, ncol, nn) in sorted(per_calc.items(), key=lambda kv: -kv[1][0]):
        print("| %s | %d | %.1e | %d |" % (calc, ncol, dev, nn))
    print("\nInteger / boolean columns that change value:\n")
    for c, k, n in flips:
        print("* `%s`: %d of %d series" % (c, k, n))
    if not flips:
        print("* none")


if __name__ == "__main__":
    main()
