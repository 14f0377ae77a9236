#!/usr/bin/env python
"""Feature selection timing: the relevance table of a [n_rows x 783] feature matrix against a binary
target -- tsfresh_amd (one tsfa_relevance_classes sweep + host tails) beside the reference's arithmetic on the host
cores (oracle/selection.py = one scipy call per feature, timed on a sample of the columns and scaled).
    python profiles/selection_timing.py [--rows 100000] [--cols 783]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100000)
    ap.add_argument("--cols", type=int, default=783)
    ap.add_argument("--cpu-cols", type=int, default=24)
    args = ap.parse_args()
    from tsfresh_amd import _native
    from tsfresh_amd.feature_selection import calculate_relevance_table
    n, m = args.rows, args.cols
    rng = np.random.default_rng(42)
    y = pd.Series(rng.integers(0, 2, n))
    X = pd.DataFrame(rng.standard_normal((n, m)), columns=["f%d" % i for i in range(m)])
    X.iloc[:, : m // 8] += 0.02 * y.to_numpy()[:, None]
    X.iloc[:, m // 8: m // 4] = np.round(X.iloc[:, m // 8: m // 4], 1)
    res = {"rows": n, "cols": m}
    codes = y.to_numpy().astype(np.int32)
    vals = np.ascontiguousarray(X.to_numpy())
    for rep in range(3):
        t0 = time.perf_counter()
        _native.relevance_classes(vals, codes, 2)
        res["device_call_s"] = time.perf_counter() - t0  # H2D of the matrix + sort / rank / count kernels + D2H
    for rep in range(2):
        t0 = time.perf_counter()
        tab = calculate_relevance_table(X, y)
        res["relevance_table_s"] = time.perf_counter() - t0
    res["relevant"] = int(tab.relevant.sum())
    from oracle.selection import relevance_table
    sub = X.iloc[:, :: max(1, m // args.cpu_cols)]
    t0 = time.perf_counter()
    relevance_table(sub, y)
    cpu = time.perf_counter() - t0
    res["cpu_oracle_s_per_feature"] = cpu / sub.shape[1]
    res["cpu_oracle_s_scaled"] = cpu / sub.shape[1] * m
    res["cpu_sample"] = "%d of %d columns, 1 core (scipy.stats.mannwhitneyu per feature)" % (sub.shape[1], m)
    res["bytes_matrix"] = int(vals.nbytes)
    # regression target: Kendall's tau / Kolmogorov-Smirnov (tsfa_relevance_real)
    yr = pd.Series(np.round(rng.standard_normal(n), 3))
    Xr = X.copy()
    Xr.iloc[:, 0] = (yr.to_numpy() > 0.3) * 1.0
    valsr = np.ascontiguousarray(Xr.to_numpy())
    for rep in range(3):
        t0 = time.perf_counter()
        _native.relevance_real(valsr, yr.to_numpy())
        res["regression_device_call_s"] = time.perf_counter() - t0
    for rep in range(2):
        t0 = time.perf_counter()
        tabr = calculate_relevance_table(Xr, yr)
        res["regression_relevance_table_s"] = time.perf_counter() - t0
    subr = Xr.iloc[:, :: max(1, m // max(2, args.cpu_cols // 4))]
    t0 = time.perf_counter()
    relevance_table(subr, yr)
    cpu = time.perf_counter() - t0
    res["regression_cpu_oracle_s_per_feature"] = cpu / subr.shape[1]
    res["regression_cpu_oracle_s_scaled"] = cpu / subr.shape[1] * m
    print(json.dumps(res))


if __name__ == "__main__":
    main()
