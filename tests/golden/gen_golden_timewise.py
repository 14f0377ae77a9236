"""Generate tests/golden/ref_timewise.npz: linear_trend_timewise (feature_calculators.py:2274) as the REAL reference
computes it through its own dispatcher `_do_extraction_on_chunk` on series that carry a DatetimeIndex.

Third-party stubs exactly as in gen_golden_main.py (the calculator itself only needs pandas + scipy).

    python tests/golden/gen_golden_timewise.py     # needs /root/reference; writes tests/golden/ref_timewise.npz
"""
import os
import sys
import warnings

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden_main  # noqa: E402,F401  (installs the stubs and puts /root/reference on sys.path)
from tsfresh.feature_extraction.extraction import _do_extraction_on_chunk  # noqa: E402


def timewise_cases():
    """-> list[(label, values float64, timestamps int64 ns)]; irregular, regular, coarse and duplicate stamps."""
    rng = np.random.default_rng(20260923)
    t0 = np.datetime64("2021-03-04T05:06:07", "ns").astype(np.int64)
    cases = []
    for n in (300, 64, 17, 5, 3):
        gaps = rng.integers(1, 7200, size=n).astype(np.int64) * 1_000_000_000 + rng.integers(0, 10**9, size=n)
        cases.append(("irregular_%d" % n, rng.standard_normal(n), t0 + np.cumsum(gaps)))
    cases.append(("hourly_200", np.cumsum(rng.standard_normal(200)), t0 + np.arange(200, dtype=np.int64) * 3600 * 10**9))
    cases.append(("millis_128", rng.standard_normal(128, dtype=np.float32).astype(np.float64),
                  t0 + np.cumsum(rng.integers(1, 50, size=128)).astype(np.int64) * 10**6))
    cases.append(("days_40_trend", 0.5 * np.arange(40) + rng.standard_normal(40),
                  t0 + np.arange(40, dtype=np.int64) * 86400 * 10**9))
    dup = t0 + np.repeat(np.arange(15, dtype=np.int64), 2) * 60 * 10**9  # duplicated stamps
    cases.append(("dup_stamps_30", rng.standard_normal(30), dup))
    cases.append(("two_points", np.array([1.0, 3.5]), t0 + np.array([0, 5400 * 10**9], dtype=np.int64)))
    cases.append(("const_y_20", np.full(20, 2.5), t0 + np.cumsum(rng.integers(1, 999, size=20)).astype(np.int64) * 10**9))
    return cases


def main():
    params = {"linear_trend_timewise": [{"attr": a} for a in ["pvalue", "rvalue", "intercept", "slope", "stderr"]]}
    cases = timewise_cases()
    names, rows = None, []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for label, x, t in cases:
            s = pd.Series(x, index=pd.DatetimeIndex(t))
            res = _do_extraction_on_chunk((label, "value", s), params, None, False)
            cols = [r[1] for r in res]
            names = names or cols
            assert cols == names
            rows.append([float(r[2]) for r in res])
    values = np.concatenate([c[1] for c in cases])
    stamps = np.concatenate([c[2] for c in cases])
    offsets = np.zeros(len(cases) + 1, dtype=np.int64)
    np.cumsum([len(c[1]) for c in cases], out=offsets[1:])
    out = os.path.join(HERE, "ref_timewise.npz")
    np.savez_compressed(out, values=values, stamps_ns=stamps, offsets=offsets, labels=np.array([c[0] for c in cases]),
                        names=np.array(names), matrix=np.asarray(rows, dtype=np.float64))
    print("wrote", out, names, np.asarray(rows).shape)


if __name__ == "__main__":
    main()
