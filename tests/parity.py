"""The parity bar, written down once (BASELINE.json north_star: integer/count features bit-exact, float features
within 1e-6 relative).

`compare(names, got, want, series)` returns a list of human-readable mismatches.

* integer / boolean / count features (INTEGER_FEATURES) must be EXACTLY equal.
* every other feature:  |got - want| <= 1e-6 * |want| + atol, where atol is a tiny absolute floor scaled by the
  magnitude of the series (results that are mathematically ~0, e.g. the imaginary part of a real FFT bin, have no
  meaningful relative error):  atol = 1e-9 * scale,  scale = max(1, max|x|) ** power(feature)  (sum|x| for FFT bins).
* NaN must match NaN, +-inf must match.

Documented, reference-side non-determinism that is EXCLUDED (see DESIGN.md "Known deviations"):
  - permutation_entropy on a series with tied values inside a window: the reference ranks with numpy's default
    argsort, which is an unstable SIMD sort for ties, so its answer depends on the CPU the reference runs on;
  - fft_coefficient "angle" of a bin whose magnitude is round-off noise (|X_k| < 1e-9 * sum|x|);
  - fourier_entropy / spkt_welch_density of a constant series (the detrended PSD is pure round-off noise);
  - regressions on rank-deficient designs (constant / exactly linear series): ar_coefficient,
    augmented_dickey_fuller, friedrich_coefficients, max_langevin_fixed_point fall back, in the reference, on the
    minimum-norm pseudo-inverse solution of a singular system (LAPACK round-off decides the digits).
"""
import numpy as np

INTEGER_FEATURES = {
    "length", "variance_larger_than_standard_deviation", "large_standard_deviation", "symmetry_looking",
    "has_duplicate_max", "has_duplicate_min", "has_duplicate", "count_above_mean", "count_below_mean", "value_count",
    "range_count", "number_crossing_m", "longest_strike_above_mean", "longest_strike_below_mean", "number_peaks",
    "number_cwt_peaks",
}
RTOL = 1e-6


def feature_of(col):
    return col.split("__")[1]


def is_integer_feature(col):
    f = feature_of(col)
    if f == "augmented_dickey_fuller":
        return 'attr_"usedlag"' in col
    return f in INTEGER_FEATURES


def _rank_deficient(x):
    x = np.asarray(x, dtype=np.float64)
    if len(x) < 3:
        return True
    d2 = np.diff(x, 2)
    return np.ptp(x) == 0 or np.max(np.abs(d2)) <= 1e-12 * max(1.0, np.max(np.abs(x)))


def _has_window_ties(x, dim):
    x = np.asarray(x)
    for lag in range(1, dim):
        if np.any(x[lag:] == x[:-lag]):
            return True
    return False


def excluded(col, x):
    f = feature_of(col)
    if f == "permutation_entropy":
        dim = int(col.split("dimension_")[1].split("__")[0])
        return _has_window_ties(x, dim)
    if f in ("ar_coefficient", "augmented_dickey_fuller", "friedrich_coefficients", "max_langevin_fixed_point"):
        return _rank_deficient(x)
    if f in ("fourier_entropy", "spkt_welch_density"):
        return np.ptp(np.asarray(x, dtype=np.float64)) == 0  # Welch PSD of a constant series is round-off noise
    return False


def atol_for(col, x):
    f = feature_of(col)
    ax = np.abs(np.asarray(x, dtype=np.float64))
    amax = max(1.0, float(ax.max()) if len(ax) else 1.0)
    if f in ("fft_coefficient", "fft_aggregated"):
        return 1e-10 * max(1.0, float(ax.sum()))
    if f in ("c3", "time_reversal_asymmetry_statistic"):
        return 1e-9 * amax ** 3
    if f in ("abs_energy", "variance", "spkt_welch_density"):
        return 1e-9 * amax ** 2
    if f in ("skewness", "kurtosis", "autocorrelation", "agg_autocorrelation", "partial_autocorrelation",
             "approximate_entropy", "sample_entropy", "linear_trend", "agg_linear_trend"):
        return 1e-9
    return 1e-9 * amax


def compare(names, got, want, series, rtol=RTOL, check_excluded=False):
    """names: list[str]; got, want: [n_series, n_cols]; series: list of 1-D arrays.  -> list[str] of mismatches."""
    bad = []
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape == (len(series), len(names)), (got.shape, want.shape, len(series), len(names))
    for i, x in enumerate(series):
        absum = float(np.abs(np.asarray(x, dtype=np.float64)).sum())
        spectrum = None
        for j, col in enumerate(names):
            g, w = got[i, j], want[i, j]
            if not check_excluded and excluded(col, x):
                continue
            if np.isnan(w) or np.isnan(g):
                if np.isnan(w) != np.isnan(g):
                    bad.append("series %d %s: got %r want %r" % (i, col, g, w))
                continue
            if np.isinf(w) or np.isinf(g):
                if g != w:
                    bad.append("series %d %s: got %r want %r" % (i, col, g, w))
                continue
            if is_integer_feature(col):
                if g != w:
                    bad.append("series %d %s: integer feature got %r want %r" % (i, col, g, w))
                continue
            if feature_of(col) == "fft_coefficient" and 'attr_"angle"' in col:
                # magnitude of the bin from the data itself (the "abs" column need not be part of the plan)
                kbin = int(col.split("coeff_")[1].split("__")[0])
                if spectrum is None:
                    spectrum = np.abs(np.fft.rfft(np.asarray(x, dtype=np.float64)))
                if kbin < len(spectrum) and spectrum[kbin] < 1e-9 * max(absum, 1e-300):
                    continue
                d = abs(g - w)
                d = min(d, 360.0 - d)  # -180 == 180
                if d > 1e-6 * 180.0:
                    bad.append("series %d %s: got %r want %r" % (i, col, g, w))
                continue
            if abs(g - w) > rtol * abs(w) + atol_for(col, x):
                bad.append("series %d %s: got %r want %r (rel %.3g)" % (i, col, g, w, abs(g - w) / max(abs(w), 1e-300)))
    return bad
