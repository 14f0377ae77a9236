"""The golden fixtures (outputs of the REAL reference, tests/golden/gen_golden_*.py) as (names, matrix, values,
offsets, series, labels) tuples.

  pair            reference run                                   compare(..., simd_golden=)
  main            ref_main.npz + ref_conda.npz                    True  (numpy's SIMD argsort ranked the windows)
  main_nosimd     ref_main_nosimd.npz + ref_conda.npz             False (stable ranks: permutation_entropy ties pinned)
  degenerate      ref_main_degenerate.npz + ref_conda_degenerate  True
  degenerate_nosimd  ref_main_degenerate_nosimd.npz + ...         False
  offset / offset_nosimd   ref_main_offset*.npz + ref_conda_offset.npz   (|mean| >> spread: the rank cuts of np.polyfit / pinv)
  long            ref_main_long.npz + ref_conda_long.npz          True  (1025 .. 8192 samples, rolled windows of one walk)
"""
import os

import numpy as np

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

PAIRS = {
    "main": ("ref_main.npz", "ref_conda.npz", True),
    "main_nosimd": ("ref_main_nosimd.npz", "ref_conda.npz", False),
    "degenerate": ("ref_main_degenerate.npz", "ref_conda_degenerate.npz", True),
    "degenerate_nosimd": ("ref_main_degenerate_nosimd.npz", "ref_conda_degenerate.npz", False),
    "offset": ("ref_main_offset.npz", "ref_conda_offset.npz", True),
    "offset_nosimd": ("ref_main_offset_nosimd.npz", "ref_conda_offset.npz", False),
    # series of 1025 .. 8192 samples (BASELINE configs[4]; golden_cases.long_series): past the reference's FFT switch of
    # agg_autocorrelation, ADF's growing maxlag, the growing CWT-peak noise window, the kernels' length classes
    "long": ("ref_main_long.npz", "ref_conda_long.npz", True),
    "long_nosimd": ("ref_main_long_nosimd.npz", "ref_conda_long.npz", False),
}


# fixtures of OTHER parameter sets than ComprehensiveFCParameters (not part of the PAIRS the golden tests iterate over)
EXTRA = {
    # tests/golden/param_cases.py: parameters away from the Comprehensive grids, the `main` series
    "sweep": ("ref_main_sweep.npz", "ref_conda_sweep.npz", True),
    # the same parameters on the rank-deficient / far-from-zero series: the second passes (k_ar_degenerate, k_langevin_dd)
    # with other AR orders and Langevin (m, r) than the Comprehensive ones
    "degenerate_sweep": ("ref_main_degenerate_sweep.npz", "ref_conda_degenerate_sweep.npz", True),
    "offset_sweep": ("ref_main_offset_sweep.npz", "ref_conda_offset_sweep.npz", True),
    # ... and on the 1025 .. 8192-sample series: lags, peak supports and coefficient indices that only exist there
    "long_sweep": ("ref_main_long_sweep.npz", "ref_conda_long_sweep.npz", True),
    # param_cases.beyond_parameters: values beyond the tuned kernels' tables (k_general, AR orders of the second pass alone)
    "beyond": ("ref_main_beyond.npz", "ref_conda_beyond.npz", True),
    "degenerate_beyond": ("ref_main_degenerate_beyond.npz", "ref_conda_degenerate_beyond.npz", True),
    "offset_beyond": ("ref_main_offset_beyond.npz", "ref_conda_offset_beyond.npz", True),
    "long_beyond": ("ref_main_long_beyond.npz", "ref_conda_long_beyond.npz", True),
}


def load(pair):
    f1, f2, simd = PAIRS[pair] if pair in PAIRS else EXTRA[pair]
    g1 = np.load(os.path.join(G, f1))
    g2 = np.load(os.path.join(G, f2))
    assert np.array_equal(g1["values"], g2["values"]) and np.array_equal(g1["offsets"], g2["offsets"])
    names = list(g1["names"]) + list(g2["names"])
    matrix = np.concatenate([g1["matrix"], g2["matrix"]], axis=1)
    values, offsets = g1["values"], g1["offsets"]
    series = [values[offsets[i]:offsets[i + 1]] for i in range(len(offsets) - 1)]
    ar_sv = {int(k[len("ar_sv_k"):]): g2[k] for k in g2.files if k.startswith("ar_sv_k")}
    return dict(names=names, matrix=matrix, values=values, offsets=offsets, series=series,
                labels=[str(l) for l in g1["labels"]], simd=simd, ar_sv=ar_sv)


def align(names_want, names_got, got):
    assert set(names_want) == set(names_got), (set(names_want) ^ set(names_got))
    idx = [names_got.index(n) for n in names_want]
    return got[:, idx]


def check_engine(engine, pair, fc_parameters, max_len=None):
    """-> (mismatches, skipped cells, total cells) of `engine` against the fixture pair.
    max_len: compare only the series of at most that many samples (the single-thread emulation of the slow general kernel on
    8192-sample series is a minute per test; the device tests compare every row)."""
    from parity import compare
    g = load(pair)
    values, offsets, series, matrix, ar_sv = g["values"], g["offsets"], g["series"], g["matrix"], g["ar_sv"]
    if max_len is not None:
        keep = [i for i, x in enumerate(series) if len(x) <= max_len]
        series = [series[i] for i in keep]
        values = np.concatenate(series)
        offsets = np.concatenate([[0], np.cumsum([len(x) for x in series])]).astype(np.int64)
        matrix = matrix[keep]
        ar_sv = {k: v[keep] for k, v in ar_sv.items()}
    got_names, got = engine(fc_parameters, values, offsets)
    skipped = []
    bad = compare(g["names"], align(g["names"], got_names, got), matrix, series, simd_golden=g["simd"],
                  skipped=skipped, ar_sv=ar_sv or None)
    return bad, skipped, matrix.size
