#!/usr/bin/env python
"""Ad-hoc fuzz of feature selection on the GPU box: random matrices (ties, binary / constant / few-valued columns), random
targets (binary, multiclass, regression, with ties) and options, tsfresh_amd against oracle/selection.py.
    python profiles/fuzz_selection.py [rounds] [seed]"""
import math
import os
import sys
import warnings

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.selection import relevance_table  # noqa: E402
from tsfresh_amd.feature_selection import calculate_relevance_table  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    for r in range(rounds):
        n = int(rng.choice([12, 40, 200, 1000, 2049, 6000]))
        m = int(rng.integers(3, 14))
        task = rng.choice(["binary", "multi", "regression", "smir"])
        X = pd.DataFrame(rng.standard_normal((n, m)), columns=["f%d" % i for i in range(m)])
        if task == "regression":
            yv = rng.standard_normal(n)
            if rng.random() < 0.5:
                yv = np.round(yv, 1)
            y = pd.Series(yv)
            sig = yv
            kw = {}
        else:
            C = 2 if task in ("binary", "smir") else int(rng.integers(3, 6))
            yv = rng.integers(0, C, n)
            if len(np.unique(yv)) < 2:
                yv[0] = 1 - yv[0]
            y = pd.Series(yv)
            sig = yv.astype(float)
            kw = {"multiclass": True, "n_significant": int(rng.integers(1, 3))} if task == "multi" else {}
            if task == "smir":
                kw["test_for_binary_target_real_feature"] = "smir"
        X["f0"] += 0.5 * sig
        X["f1"] = np.round(X["f1"] + 0.3 * sig, int(rng.integers(0, 2)))
        X["f2"] = (rng.random(n) < 0.3 + 0.2 * np.tanh(sig - np.mean(sig))) * float(rng.integers(1, 4))
        if m > 5:
            X["f3"] = 1.25
            X["f4"] = rng.integers(0, 3, n).astype(float)
        kw["hypotheses_independent"] = bool(rng.random() < 0.5)
        kw["fdr_level"] = float(rng.choice([0.01, 0.05, 0.2]))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            tab = calculate_relevance_table(X, y, **kw)
            want = relevance_table(X, y, **kw)
        nb = 0
        for f in X.columns:
            for c, w in want[f].items():
                g = tab.loc[f][c]
                if isinstance(w, (bool, np.bool_)):
                    ok = bool(g) == bool(w)
                elif isinstance(w, str):
                    ok = g == w
                elif isinstance(w, float) and math.isnan(w):
                    ok = isinstance(g, float) and math.isnan(g)
                else:
                    ok = abs(g - w) <= 1e-9 * abs(w) + 1e-300
                if not ok:
                    nb += 1
                    print("  MISMATCH", task, n, f, c, g, w)
        bad += nb
        print("round", r, task, "n", n, "m", m, kw, "mismatches", nb)
    print("TOTAL mismatches", bad)


if __name__ == "__main__":
    main()
