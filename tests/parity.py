"""The parity bar, written down once (BASELINE.json north_star: integer/count features bit-exact, float features
within 1e-6 relative).

`compare(names, got, want, series)` returns a list of human-readable mismatches.

* integer / boolean / count features (INTEGER_FEATURES) must be EXACTLY equal.
* every other feature:  |got - want| <= 1e-6 * |want| + atol, where atol is a tiny absolute floor scaled by the
  magnitude of the series (results that are mathematically ~0, e.g. the imaginary part of a real FFT bin, have no
  meaningful relative error):  atol = 1e-9 * scale,  scale = max(1, max|x|) ** power(feature)  (sum|x| for FFT bins).
* NaN must match NaN, +-inf must match.

EXCLUSIONS.  A cell is skipped only where the REFERENCE's value is a function of round-off, i.e. where no
implementation -- not even the reference on another CPU or LAPACK build -- reproduces it.  Each predicate below is
computed from the series alone (never from `got`), and `compare(..., skipped=[])` reports what was skipped so a test
can bound it.

  R1  permutation_entropy with tied values inside a window, ONLY against fixtures generated with numpy's SIMD sorts
      (`simd_golden=True`: tests/golden/ref_main*.npz without the _nosimd suffix).  The reference ranks windows with
      np.argsort's default kind, an unstable vectorised sort on AVX-512/AVX2 hosts; numpy's scalar path (fixtures
      *_nosimd.npz, gen_golden_main.py --nosimd) ranks ties stably, as the kernels and the oracle do, and is compared
      without exclusion.
  R2  fft_coefficient "angle" of a bin whose magnitude is round-off (|X_k| < 1e-9 * sum|x|).
  R3  fourier_entropy / spkt_welch_density of a constant series (the detrended PSD is pure round-off).
  R4  ar_coefficient / augmented_dickey_fuller when the regression design (as statsmodels builds it) has a singular
      value inside (5e-16, 1e-8) * s_max: statsmodels' pinv cuts at 1e-15 * s_max, so such a value is either LAPACK
      round-off that gets inverted (const_1024: the reference returns -0.0754 for coefficients whose minimum-norm
      value is 0.0091) or a genuine direction whose solution the float64 SVD resolves to eps / 1e-8 at best, measured: 3e-6 off at 1e-9
      (tiny_noise_ramp_300: exact rational arithmetic gives usedlag 12, the reference 0 -- see DESIGN.md 4.3).
      Exactly rank-deficient designs whose round-off stays below the cut (constant / linear / periodic series up to a
      few hundred samples) are NOT excluded: the minimum-norm solution is well defined and must match.
  R5  augmented_dickey_fuller when a lag-search regression fits perfectly (ssr <= 1e-18 * yy, yy > 0): the AIC is
      n log(round-off) and the t statistic (round-off)/(round-off); the kernels return AIC = -inf and 0/0 = NaN or
      x/0 = +-inf there (tests/test_degenerate.py pins that behaviour).
  R6  max_langevin_fixed_point when the fitted cubic's leading coefficient is round-off (a perfectly linear drift):
      np.roots then returns a root near -c2/c3.
  R7  agg_linear_trend: slope-type attributes when the chunk aggregates differ by round-off only
      (0 < ptp <= 1e-12 * max|agg|), and "stderr" of linear_trend / agg_linear_trend when 1 - r^2 < 1e-9
      (scipy's sqrt((1 - r^2) ...) cancels).
  R9  partial_autocorrelation from the lag at which the Levinson-Durbin innovation variance has dropped below
      1e-9 * acov[0] (an exactly predictable series: periodic, linear): the next coefficient divides by round-off.
  R10 fourier_entropy when a normalised Welch density lies on an edge of np.histogram's bins up to round-off (exactly
      periodic series: a Hann-leaked bin of exactly 1/4 of the peak sits on the edge 0.25 of 100 bins).
  R11 friedrich_coefficients / max_langevin_fixed_point when a singular value of np.polyfit's scaled Vandermonde design
      lies within 10 % of its cut (rcond = len(x) * eps, tests/polyfit_mp.py): the RANK the reference's float64
      SVD sees -- and with it which minimum-norm cubic it returns -- is decided by LAPACK round-off (the computed small
      singular values carry an absolute error of ~eps * s_max).  Outside the band the columns are COMPARED, with the
      tolerance the reference's own arithmetic has there (tolerance_for): 2 eps kappa (|c| + kappa |resid| |v_min|), kappa =
      s_max / the smallest kept singular value (measured against 60-digit arithmetic: the reference deviates by up to
      1.2 eps kappa (...), the kernels' double-double pass by < 1e-2 of that; tests/test_offset.py).  The largest real root
      moves with the well-conditioned VALUES of the cubic, not with its monomial coefficients.
  R8  number_cwt_peaks when a CWT row has two neighbouring values on top of a hump that are equal up to round-off
      (1e-12 of the row's magnitude; symmetric integer-valued or periodic data): which one is the STRICT relative
      maximum that starts a ridge line depends on the summation order of scipy's convolution.
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


def _param(col, key, cast=float):
    return cast(col.split(key + "_")[1].split("__")[0].strip('"'))


def _has_window_ties(x, dim):
    x = np.asarray(x)
    for lag in range(1, dim):
        if np.any(x[lag:] == x[:-lag]):
            return True
    return False


def _pinv_unstable(X):
    if X.size == 0 or X.shape[1] == 0:
        return False
    s = np.linalg.svd(X, compute_uv=False)
    if not np.isfinite(s).all() or s[0] == 0:
        return False
    r = s / s[0]
    return bool(np.any((r > 5e-16) & (r < 1e-8)))


def _ar_design(x, k):
    n = len(x)
    if n < 2 * k + 2:
        return None
    rows = np.arange(k, n)
    return np.column_stack([np.ones(n - k)] + [x[rows - j] for j in range(1, k + 1)])


def _adf_state(x):
    """-> (unstable, perfect): properties of the lag-search design of statsmodels.adfuller(x, autolag="AIC")."""
    from oracle.third_party import _add_const
    n = len(x)
    maxlag = min(n // 2 - 2, int(np.ceil(12.0 * np.power(n / 100.0, 0.25))))
    if maxlag < 0:
        return False, False
    d = np.diff(x)
    rows = np.arange(maxlag, len(d))
    Z = np.column_stack([x[rows]] + [d[rows - j] for j in range(1, maxlag + 1)])
    y = d[rows]
    full = _add_const(Z, prepend=True)
    yy = float(y @ y)
    perfect = False
    if yy > 0:
        beta = np.linalg.lstsq(full, y, rcond=None)[0]
        r = y - full @ beta
        perfect = float(r @ r) <= 1e-18 * yy
    return _pinv_unstable(full), perfect


def _langevin_noise_cubic(x, m, r):
    from oracle.calculators import friedrich_coefficients_of
    c = friedrich_coefficients_of(np.asarray(x, dtype=np.float64), m, r)
    if c is None or not np.all(np.isfinite(c)):
        return False
    s = max(float(np.max(np.abs(x))), 1e-300)
    mag = np.abs(c) * s ** np.arange(len(c) - 1, -1, -1)
    return bool(mag[0] <= 1e-8 * mag.max())


def _langevin_fit_facts(x, m, r):
    """Conditioning of the np.polyfit call of fc.py:131-173 for this series, or None when the reference does not fit
    (qcut fails / fewer than two samples).  See tolerance_for."""
    import warnings
    import polyfit_mp
    x = np.asarray(x, dtype=np.float64)
    if len(x) < 2 or not np.all(np.isfinite(x)):
        return None
    bm = polyfit_mp.bin_means(x, r)
    if bm is None or len(bm[0]) == 0:
        return None
    xm, ym = bm
    A, scale = polyfit_mp.scaled_design(xm, m)
    if not (np.all(np.isfinite(A)) and np.all(scale > 0)):
        return None
    _, S, Vt = np.linalg.svd(A, full_matrices=False)
    rcond = len(xm) * polyfit_mp.EPS
    ratio = S / S[0]
    kept = np.where(ratio > rcond)[0]
    in_band = bool(np.any((ratio > rcond / polyfit_mp.BAND) & (ratio < rcond * polyfit_mp.BAND)))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        coef = np.polyfit(xm, ym, m)
    # residual of the fit, evaluated about the centre of the bins (the monomial form cancels catastrophically)
    c0 = float(np.mean(xm))
    shifted = np.poly1d(coef)(np.poly1d([1.0, c0]))       # p(u + c0) as a polynomial in u
    resid = float(np.linalg.norm(ym - shifted(xm - c0)))
    return {"kappa": float(1.0 / ratio[kept[-1]]), "in_band": in_band, "resid": resid, "coef": coef, "centre": c0,
            "shifted": shifted, "halfwidth": float(np.max(np.abs(xm - c0))), "scaled_norm": float(np.linalg.norm(coef * scale)),
            "noise_dir": np.abs(Vt[kept[-1]]) / scale,      # |v_min| in coefficient units
            "dmax": float(np.max(np.abs(np.diff(x))))}


def _chunk_aggs(x, f_agg, chunk_len):
    x = np.asarray(x, dtype=np.float64)
    return np.array([getattr(x[i * chunk_len:(i + 1) * chunk_len], f_agg)() for i in range(int(np.ceil(len(x) / chunk_len)))])


def _cwt_peaks_ambiguous(x, n):
    """A CWT row (scipy.signal.find_peaks_cwt's convolution with the Ricker wavelet) holds two neighbouring values that
    are equal up to round-off on top of a hump: which of them -- or neither -- is the strict relative maximum that
    starts a ridge line depends on the summation order of the convolution."""
    from scipy.signal._wavelets import _cwt
    from oracle.calculators import _ricker
    x = np.asarray(x, dtype=np.float64)
    if len(x) < 4 or np.ptp(x) == 0:
        return False
    rows = _cwt(x, _ricker, np.arange(1, n + 1))
    for c in rows:
        tol = 1e-12 * np.max(np.abs(c))
        near = np.abs(c[1:] - c[:-1]) <= tol                       # c[i] ~ c[i+1]
        left = np.concatenate([[True], c[1:-1] >= c[:-2] - tol])   # c[i] >= c[i-1]
        right = np.concatenate([c[1:-1] >= c[2:] - tol, [True]])   # c[i+1] >= c[i+2]
        if np.any(near & left & right):
            return True
    return False


def _psd_on_bin_edge(x, bins):
    """fourier_entropy (feature_calculators.py:2216): binned_entropy(pxx / max(pxx), bins) -- is a value within round-off
    of an interior edge of np.histogram(., bins)?"""
    from scipy.signal import welch
    if len(x) < 2:
        return False
    _, pxx = welch(x, nperseg=min(len(x), 256))
    if not np.all(np.isfinite(pxx)) or np.max(pxx) <= 0:
        return False
    v = pxx / np.max(pxx)
    edges = np.histogram_bin_edges(v, bins=bins)[1:-1]
    if len(edges) == 0:
        return False
    d = np.abs(v[:, None] - edges[None, :]).min()
    return bool(d < 1e-9)


def _r_of_chunks(agg):
    agg = np.asarray(agg, dtype=np.float64)
    if len(agg) < 3 or np.ptp(agg) == 0 or not np.all(np.isfinite(agg)):
        return None
    return float(np.corrcoef(np.arange(len(agg), dtype=np.float64), agg)[0, 1])


def _r2_cancels(rvalue):
    # two-point fits are exact (r = +-1, stderr = 0 by scipy's special case) and stay compared
    return rvalue is not None and np.isfinite(rvalue) and 1.0 - rvalue * rvalue < 1e-9


def _pacf_noise_lag(x):
    """First lag whose Levinson-Durbin step divides by an innovation variance at round-off level (or a large number)."""
    from oracle.third_party import acovf_adjusted
    n = len(x)
    nlags = min(40, n // 2 - 1)
    if nlags < 1:
        return 10 ** 9
    acv = acovf_adjusted(x, nlags)
    if not acv[0] > 0:
        return 10 ** 9
    phi_prev = np.zeros(nlags + 1)
    phi_prev[1] = acv[1] / acv[0]
    sig = acv[0] - phi_prev[1] * acv[1]
    for k in range(2, nlags + 1):
        if not abs(sig) > 1e-9 * acv[0]:
            return k
        pkk = (acv[k] - np.dot(phi_prev[1:k], acv[1:k][::-1])) / sig
        cur = phi_prev.copy()
        for j in range(1, k):
            cur[j] = phi_prev[j] - pkk * phi_prev[k - j]
        cur[k] = pkk
        sig = sig * (1 - pkk * pkk)
        phi_prev = cur
    return 10 ** 9


class _SeriesFacts:
    """Lazily evaluated facts about one series (SVDs and fits are only computed if a column asks)."""

    def __init__(self, x):
        self.x = np.asarray(x, dtype=np.float64)
        self._c = {}

    def get(self, key, fn):
        if key not in self._c:
            self._c[key] = fn()
        return self._c[key]


def excluded(col, x, simd_golden=False, facts=None, rvalue=None):
    f = feature_of(col)
    facts = facts or _SeriesFacts(x)
    xv = facts.x
    if f == "permutation_entropy":
        return simd_golden and _has_window_ties(xv, _param(col, "dimension", int))                       # R1
    if f in ("fourier_entropy", "spkt_welch_density"):
        if len(xv) > 0 and np.ptp(xv) == 0:
            return True                                                                                   # R3
        if f == "fourier_entropy":
            bins = _param(col, "bins", int)
            return facts.get(("fe_edge", bins), lambda: _psd_on_bin_edge(xv, bins))                      # R10
        return False
    if f == "ar_coefficient":
        k = _param(col, "k", int)
        X = facts.get(("ar", k), lambda: _ar_design(xv, k))
        return X is not None and facts.get(("ar_unst", k), lambda: _pinv_unstable(X))                    # R4
    if f == "augmented_dickey_fuller":
        unstable, perfect = facts.get("adf", lambda: _adf_state(xv))
        return unstable or perfect                                                                       # R4, R5
    if f in ("max_langevin_fixed_point", "friedrich_coefficients"):
        m, r = _param(col, "m", int), _param(col, "r", float)
        r = int(r) if float(r).is_integer() else r
        fit = facts.get(("langfit", m, r), lambda: _langevin_fit_facts(xv, m, r))
        if fit is not None and fit["in_band"]:
            return True                                                                                   # R11
        if f == "friedrich_coefficients":
            return False
        return facts.get(("lang", m, r), lambda: _langevin_noise_cubic(xv, m, r))                        # R6
    if f == "agg_linear_trend":
        attr = col.split('attr_"')[1].split('"')[0]
        f_agg, cl = col.split('f_agg_"')[1].split('"')[0], _param(col, "chunk_len", int)
        agg = facts.get(("agg", f_agg, cl), lambda: _chunk_aggs(xv, f_agg, cl))
        if attr != "intercept" and len(agg) > 1 and 0 < np.ptp(agg) <= 1e-12 * np.max(np.abs(agg)):
            return True                                                                                   # R7
        if attr == "stderr" and len(agg) > 2 and rvalue is None:  # the plan does not hold the sibling column
            rvalue = _r_of_chunks(agg)
        return attr == "stderr" and len(agg) > 2 and _r2_cancels(rvalue)
    if f == "linear_trend":
        if 'attr_"stderr"' in col and len(xv) > 2 and rvalue is None:
            rvalue = _r_of_chunks(xv)
        return 'attr_"stderr"' in col and len(xv) > 2 and _r2_cancels(rvalue)
    if f == "partial_autocorrelation":
        return _param(col, "lag", int) >= facts.get("pacf", lambda: _pacf_noise_lag(xv))                   # R9
    if f == "number_cwt_peaks":
        nn = _param(col, "n", int)
        return facts.get(("cwtp", nn), lambda: _cwt_peaks_ambiguous(xv, nn))                             # R8
    return False


COND_FACTOR = 2.0
EPS = float(np.finfo(np.float64).eps)


def tolerance_for(col, x, want, facts):
    """-> (rtol, atol) of one cell: 1e-6 and a tiny dimension-aware floor, except where the reference's own float64
    arithmetic is measurably worse than that (R11): a least-squares solution computed by a backward-stable float64
    solver deviates from the exact one by  eps * kappa * (|c| + kappa * |resid| * |v_min|)  (Wedin; measured against
    60-digit arithmetic on 600 random offset series: at most 1.2 x that, tests/polyfit_mp.py) -- v_min the right singular
    vector of the smallest kept singular value.  The values of the fitted cubic move by eps * (kappa |resid| + |c_scaled|)
    on the range of the bin means and by the Chebyshev factor T_m(d / halfwidth) beyond it; a simple root moves by that
    over |p'(root)|."""
    f = feature_of(col)
    if f in ("max_langevin_fixed_point", "friedrich_coefficients"):
        m, r = _param(col, "m", int), _param(col, "r", float)
        r = int(r) if float(r).is_integer() else r
        fit = facts.get(("langfit", m, r), lambda: _langevin_fit_facts(facts.x, m, r))
        if fit is None or not np.isfinite(fit["kappa"]):
            return RTOL, atol_for(col, x)
        noise = COND_FACTOR * EPS * fit["kappa"]
        if f == "friedrich_coefficients":
            # dimension of coefficient j: delta / x^(m - j); the floor only matters for coefficients that are ~0
            j = _param(col, "coeff", int)
            amax = max(float(np.max(np.abs(facts.x))), 1e-300)
            atol = 1e-9 * fit["dmax"] / amax ** max(m - j, 0)
            if 0 <= j <= m:
                atol += noise * fit["kappa"] * fit["resid"] * fit["noise_dir"][j]
            return max(RTOL, noise), atol
        atol = atol_for(col, x)
        if np.isfinite(want):
            u = want - fit["centre"]
            slope = abs(float(fit["shifted"].deriv()(u)))
            t = max(1.0, abs(u) / fit["halfwidth"]) if fit["halfwidth"] > 0 else 1.0
            grow = float(np.polynomial.chebyshev.Chebyshev.basis(m)(t))
            if slope > 0:
                atol += COND_FACTOR * EPS * grow * (fit["kappa"] * fit["resid"] + fit["scaled_norm"]) / slope
        return RTOL, atol
    return RTOL, atol_for(col, x)


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


def _rvalue_lookup(names, want_row):
    """rvalue of the sibling column of a linear_trend / agg_linear_trend "stderr" column, if the plan holds it."""
    idx = {n: j for j, n in enumerate(names)}

    def find(col):
        sib = col.replace('attr_"stderr"', 'attr_"rvalue"')
        j = idx.get(sib)
        return None if j is None else float(want_row[j])
    return find


def compare(names, got, want, series, rtol=RTOL, check_excluded=False, simd_golden=False, skipped=None):
    """names: list[str]; got, want: [n_series, n_cols]; series: list of 1-D arrays.  -> list[str] of mismatches.
    skipped: optional list that receives (series index, column) of every excluded cell."""
    bad = []
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape == (len(series), len(names)), (got.shape, want.shape, len(series), len(names))
    for i, x in enumerate(series):
        absum = float(np.abs(np.asarray(x, dtype=np.float64)).sum())
        spectrum = None
        facts = _SeriesFacts(x)
        rv = _rvalue_lookup(names, want[i])
        for j, col in enumerate(names):
            g, w = got[i, j], want[i, j]
            if not check_excluded:
                r = rv(col) if 'attr_"stderr"' in col else None
                if excluded(col, x, simd_golden=simd_golden, facts=facts, rvalue=r):
                    if skipped is not None:
                        skipped.append((i, col))
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
                if kbin < len(spectrum) and spectrum[kbin] < 1e-9 * max(absum, 1e-300):   # R2
                    continue
                d = abs(g - w)
                d = min(d, 360.0 - d)  # -180 == 180
                if d > 1e-6 * 180.0:
                    bad.append("series %d %s: got %r want %r" % (i, col, g, w))
                continue
            rt, at = tolerance_for(col, x, w, facts)
            if abs(g - w) > max(rtol, rt) * abs(w) + at:
                bad.append("series %d %s: got %r want %r (rel %.3g)" % (i, col, g, w, abs(g - w) / max(abs(w), 1e-300)))
    return bad
