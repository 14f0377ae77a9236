#!/usr/bin/env python
"""Which cells does NO kernel write?  Runs plans with a sentinel pre-fill (plan option "fill") over series of many lengths
and lists, per calculator, the lengths at which a column kept the sentinel -- those calculators would need the NaN
pre-fill (k_fill_nan); every other column is written by its kernel for every series.

    python profiles/fill_audit.py > gpurun_out/fill_audit.json
    TSFA_LIB=tsfresh_amd/libtsfresh_amd_lab.so python profiles/fill_audit.py --control   # the positive control (lab build)
"""
import json
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SENT = 123456.789

from tsfresh_amd import _native  # noqa: E402
from tsfresh_amd.feature_extraction import settings  # noqa: E402
from tsfresh_amd.feature_extraction.plan import compile_fc_parameters  # noqa: E402

LENGTHS = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 15, 16, 17, 20, 21, 22, 23, 24, 31, 32, 33, 45, 63, 64, 65, 100, 127, 128, 129,
           200, 255, 256, 257, 300, 511, 512, 513, 1000, 1023, 1024, 1025, 1251, 2047, 2048, 2049, 3000, 4096, 4097, 5000, 8192]


def run(params, series, dtype, options=None):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fplan = compile_fc_parameters(params)
    plan = _native.Plan(fplan.native_specs(_native.calc_id), device=0)
    plan.set_option("fill", SENT)
    for name, value in (options or {}).items():
        plan.set_option(name, value)
    values = np.concatenate(series).astype(dtype)
    offsets = np.concatenate([[0], np.cumsum([len(s) for s in series])]).astype(np.int64)
    out = plan.extract_host(values, offsets)
    plan.close()
    return fplan.names, out


def positive_control():
    """The sentinel really is what an unwritten cell shows: with one family's launch skipped (option "skip_family", which only
    the LAB build of the library has: TSFA_LIB=.../libtsfresh_amd_lab.so) its columns -- and only its columns -- keep it."""
    rng = np.random.default_rng(2)
    series = [rng.standard_normal(300) for _ in range(4)]
    params = {"lempel_ziv_complexity": [{"bins": 10}], "mean": None, "median": None}
    names, out = run(params, series, np.float32, options={"skip_family": 6})    # TSFA_FAM_SEQ
    col = {n.split("__")[0]: i for i, n in enumerate(names)}
    ok = bool((out[:, col["lempel_ziv_complexity"]] == SENT).all() and (out[:, col["mean"]] != SENT).all()
              and (out[:, col["median"]] != SENT).all())
    names, out = run(params, series, np.float32)
    return ok and bool((out != SENT).all())


def audit(lengths=LENGTHS, sets=("comprehensive", "extra", "minimal")):
    """-> {calculator: {column: [lengths at which a cell kept the sentinel]}}"""
    rng = np.random.default_rng(1)
    kept = {}
    extra = {
        "fft_coefficient": [{"coeff": c, "attr": a} for c in (0, 1, 5, 50, 99, 500, 3000) for a in ("real", "imag", "abs", "angle")],
        "cwt_coefficients": [{"widths": (2, 5, 10, 20), "coeff": c, "w": w} for c in (0, 3, 14) for w in (2, 5, 10, 20)],
        "spkt_welch_density": [{"coeff": c} for c in (0, 2, 5, 8, 100, 129, 200)],
        "ar_coefficient": [{"coeff": c, "k": k} for k in (1, 3, 10) for c in (0, 1, 3, 10, 11)],
        "partial_autocorrelation": [{"lag": l} for l in (0, 1, 5, 9, 20, 40)],
        "autocorrelation": [{"lag": l} for l in (0, 1, 9, 50, 200)],
        "agg_autocorrelation": [{"f_agg": f, "maxlag": m} for f in ("mean", "median", "var") for m in (1, 5, 40, 60)],
        "number_peaks": [{"n": n} for n in (1, 3, 50, 200)],
        "time_reversal_asymmetry_statistic": [{"lag": l} for l in (1, 3, 100)],
        "c3": [{"lag": l} for l in (1, 3, 100)],
        "agg_linear_trend": [{"attr": a, "chunk_len": c, "f_agg": f} for a in ("rvalue", "stderr") for c in (5, 50, 500) for f in ("max", "var")],
        "energy_ratio_by_chunks": [{"num_segments": s, "segment_focus": f} for s, f in ((10, 0), (10, 9), (3, 2), (200, 150))],
        "sample_entropy": None, "approximate_entropy": [{"m": 2, "r": 0.3}, {"m": 3, "r": 0.5}],
        "number_cwt_peaks": [{"n": 1}, {"n": 5}, {"n": 16}],
        "lempel_ziv_complexity": [{"bins": b} for b in (2, 10, 100)],
        "fourier_entropy": [{"bins": b} for b in (2, 100)], "permutation_entropy": [{"tau": t, "dimension": d} for t in (1, 3) for d in (3, 7)],
        "friedrich_coefficients": [{"coeff": c, "m": 3, "r": 30} for c in range(4)], "max_langevin_fixed_point": [{"m": 3, "r": 30}],
        "change_quantiles": [{"ql": 0.2, "qh": 0.8, "isabs": True, "f_agg": "var"}, {"ql": 0.0, "qh": 0.2, "isabs": False, "f_agg": "mean"}],
        "index_mass_quantile": [{"q": 0.5}], "linear_trend": [{"attr": a} for a in ("pvalue", "rvalue", "intercept", "slope", "stderr")],
        "augmented_dickey_fuller": [{"attr": a, "autolag": "AIC"} for a in ("teststat", "pvalue", "usedlag")],
        "fft_aggregated": [{"aggtype": a} for a in ("centroid", "variance", "skew", "kurtosis")],
        "quantile": [{"q": 0.3}], "mean_n_absolute_max": [{"number_of_maxima": 7}, {"number_of_maxima": 1}],
        "ratio_beyond_r_sigma": [{"r": 1}], "symmetry_looking": [{"r": 0.1}], "large_standard_deviation": [{"r": 0.2}],
        "binned_entropy": [{"max_bins": 10}], "benford_correlation": None, "skewness": None, "kurtosis": None,
        "mean_second_derivative_central": None, "cid_ce": [{"normalize": True}, {"normalize": False}],
    }
    all_sets = {"comprehensive": settings.ComprehensiveFCParameters(), "extra": extra, "minimal": settings.MinimalFCParameters()}
    for set_name in sets:
        params = all_sets[set_name]
        for dtype in (np.float32, np.float64):
            shapes = [("ragged", [rng.standard_normal(n) for n in lengths])]
            for n in lengths:
                odd = rng.standard_normal(n)
                odd[rng.integers(0, max(n, 1), size=min(n, 2))] = [np.inf, np.nan][:min(n, 2)]    # non-finite samples
                shapes.append(("len%d" % n, [rng.standard_normal(n), np.cumsum(rng.standard_normal(n)), np.full(n, 1.5), np.zeros(n), odd]))
            for shape_name, series in shapes:
                names, out = run(params, series, dtype)
                hit = np.argwhere(out == SENT)
                for r, c in hit:
                    kept.setdefault(names[c].split("__")[0], {}).setdefault(names[c], set()).add(len(series[r]))
    return {calc: {nm: sorted(v) for nm, v in cols.items()} for calc, cols in kept.items()}


def main():
    if "--control" in sys.argv:
        ok = positive_control()
        print("positive control ok" if ok else "positive control FAILED")
        sys.exit(0 if ok else 1)
    control = None   # (needs the lab build: run with --control under TSFA_LIB)
    sys.stderr.write("positive control (a skipped family keeps the sentinel): %s\n" % control)
    doc = audit()
    print(json.dumps({"sentinel": SENT, "positive_control_ok": control, "lengths_tried": LENGTHS, "calculators_that_leave_cells": doc}, indent=1))
    sys.stderr.write("calculators leaving cells unwritten: %s\n" % sorted(doc))


if __name__ == "__main__":
    main()
