"""ORACLE (test infrastructure only): restatement of the third-party numerics the reference calls but that are NOT
importable in the main interpreter of this image.

  * statsmodels >= 0.13 (setup.cfg:42; floor only, not vendored under /root/reference)
        acf / acovf / pacf("ld") / levinson_durbin / adfuller(autolag=...) / mackinnonp / AutoReg(trend="c")
        call sites: feature_calculators.py:429 (acf), :490 (pacf), :521 (adfuller), :1493-1494 (AutoReg)
  * PyWavelets (setup.cfg:44, unpinned)
        pywt.cwt(x, scales, "mexh")     call site: feature_calculators.py:1402

  * stumpy >= 1.11.1 (setup.cfg:47; NOT installed in either interpreter of this image, not vendored)
        stumpy.core.mass(Q, T) / stumpy.core.mass_absolute(Q, T)    call sites: feature_calculators.py:2514, :2516
        restated from the library's published definition (core.py: _calculate_squared_distance, _mass_absolute /
        _p_norm_distance_profile); PARITY UNPINNED against the library itself -- the only vectors are the four known
        answers of the reference's own unit test (test_feature_calculations.py:2017-2037), checked in
        tests/test_query_similarity.py.

Pinning: tests/golden/gen_golden_conda.py runs the REAL libraries (statsmodels 0.12.2 with two import shims,
pywt 1.1.1) under /opt/conda/bin/python3.9 in the build container and commits their outputs under tests/golden/;
tests/test_oracle_golden.py checks this file against those vectors and against the reference's own known-answer
tests (tests/units/feature_extraction/test_feature_calculations.py:238-412, 1055-1127).  statsmodels 0.12.2 is
below the reference's floor (0.13); the algorithms restated here did not change between the two, but the
short-series error behaviour of AutoReg/adfuller is taken from 0.12.2.
"""
import numpy as np
from scipy import stats


# ------------------------------------------------------------------------------------------------
# statsmodels.tsa.stattools
# ------------------------------------------------------------------------------------------------
def acovf_adjusted(x, nlags=None):
    """acovf(x, adjusted=True, demean=True, fft=False): sum_t xo[t] xo[t+k] / (n - k)."""
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    xo = x - x.mean()
    kmax = n - 1 if nlags is None else min(nlags, n - 1)
    out = np.empty(kmax + 1)
    for k in range(kmax + 1):
        out[k] = np.dot(xo[: n - k], xo[k:]) / (n - k)
    return out


def acf_adjusted(x, nlags):
    """acf(x, adjusted=True, nlags=nlags): avf[:nlags + 1] / avf[0]."""
    avf = acovf_adjusted(x, nlags)
    return avf / avf[0]


def levinson_durbin_pacf(acv, nlags):
    """levinson_durbin(acv, nlags, isacov=True)[2]: the diagonal of phi with pacf[0] = 1."""
    order = nlags
    phi = np.zeros((order + 1, order + 1))
    sig = np.zeros(order + 1)
    phi[1, 1] = acv[1] / acv[0]
    sig[1] = acv[0] - phi[1, 1] * acv[1]
    for k in range(2, order + 1):
        phi[k, k] = (acv[k] - np.dot(phi[1:k, k - 1], acv[1:k][::-1])) / sig[k - 1]
        for j in range(1, k):
            phi[j, k] = phi[j, k - 1] - phi[k, k] * phi[k - j, k - 1]
        sig[k] = sig[k - 1] * (1 - phi[k, k] ** 2)
    pacf = np.diag(phi).copy()
    pacf[0] = 1.0
    return pacf


def pacf_ld(x, nlags):
    """pacf(x, method="ld", nlags): adjusted autocovariance + Levinson-Durbin; ValueError as statsmodels raises."""
    x = np.asarray(x, dtype=np.float64)
    if nlags >= x.shape[0] // 2:
        raise ValueError("Can only compute partial correlations for lags up to 50% of the sample size.")
    return levinson_durbin_pacf(acovf_adjusted(x, nlags), nlags)


_TAU_MAX_C, _TAU_MIN_C, _TAU_STAR_C = 2.74, -18.83, -1.61
_TAU_C_SMALLP = np.array([2.1659, 1.4412, 3.8269]) * np.array([1, 1, 1e-2])
_TAU_C_LARGEP = np.array([1.7339, 9.3202, -1.2745, -1.0368]) * np.array([1, 1e-1, 1e-1, 1e-2])


def mackinnonp_c(teststat):
    """adfvalues.mackinnonp(teststat, regression="c", N=1)."""
    if teststat > _TAU_MAX_C:
        return 1.0
    if teststat < _TAU_MIN_C:
        return 0.0
    coef = _TAU_C_SMALLP if teststat <= _TAU_STAR_C else _TAU_C_LARGEP
    return stats.norm.cdf(np.polyval(coef[::-1], teststat))


def _ols(y, X, probe=None):
    """statsmodels OLS(y, X).fit(method="pinv") (regression/linear_model.py:300-338, tools/tools.py:398 pinv_extended):
    params = pinv(X) y with singular values <= 1e-15 * s_max zeroed; rank = matrix_rank(diag(s)) (tolerance
    s_max * p * eps); normalized_cov_params = pinv(X) pinv(X)^T; ssr from the residuals.
    -> (params, ssr, rank, normalized_cov_params)
    probe (tests/parity.py only; .rng, .relative_eps): the rows (observations) and columns (regressors) are permuted
    before the solve and the result is mapped back -- mathematically the same regression, numerically another run of
    the same LAPACK routine: the spread over a few probes IS the reference's round-off on this design."""
    X = np.asarray(X, dtype=np.float64)
    if X.shape[1] == 0:
        return np.zeros(0), float(y @ y), 0, np.zeros((0, 0))
    if probe is not None and getattr(probe, "relative_eps", 0.0):
        # ... or the design with every entry moved by a few ulp (the reference's answer for an input that differs from
        # the real one by less than its own representation error)
        X = X * (1.0 + probe.relative_eps * probe.rng.uniform(-1.0, 1.0, size=X.shape))
    elif probe is not None:
        rows, cols = probe.rng.permutation(X.shape[0]), probe.rng.permutation(X.shape[1])
        b2, ssr, rank, nc2 = _ols(np.asarray(y)[rows], X[rows][:, cols])
        beta = np.empty_like(b2)
        beta[cols] = b2
        ncov = np.empty_like(nc2)
        ncov[np.ix_(cols, cols)] = nc2
        return beta, ssr, rank, ncov
    u, s, vt = np.linalg.svd(X, False)
    s_orig = s.copy()
    cutoff = 1e-15 * s.max()
    sinv = np.where(s > cutoff, 1.0 / np.where(s > cutoff, s, 1.0), 0.0)
    pinv = vt.T @ (sinv[:, None] * u.T)
    rank = int(np.linalg.matrix_rank(np.diag(s_orig)))
    beta = pinv @ y
    resid = y - X @ beta
    return beta, float(resid @ resid), rank, pinv @ pinv.T


def _add_const(X, prepend):
    """tsatools.add_trend(X, "c", prepend, has_constant="skip"): the constant column is NOT added when X already holds
    an exactly constant, non-zero column (tsatools.py:112-136)."""
    if X.shape[1]:
        ptp0 = np.ptp(X, axis=0)
        if np.any((ptp0 == 0) & (X[0] != 0)):
            return X
    ones = np.ones((X.shape[0], 1))
    return np.column_stack([ones, X]) if prepend else np.column_stack([X, ones])


ADF_TSTAT_STOP = 1.6448536269514722   # stats.norm.ppf(.95), stattools._autolag "t-stat"


def adfuller(x, autolag="AIC", probe=None):
    """adfuller(x, autolag=autolag) -> (teststat, pvalue, usedlag); raises ValueError like statsmodels
    (stattools.py:160-380, _autolag :63-147).  autolag "AIC" / "BIC": the lag minimising -2 llf + 2 rank / -2 llf +
    log(nobs) rank (linear_model.py:1827, :1843 with df_model = rank - k_constant; llf of OLS.loglike :896-903); "t-stat":
    from maxlag downwards, the first lag whose LAST coefficient has |t| >= norm.ppf(.95), else the smallest; None: maxlag.
    t value = params[j] / sqrt(ssr / (nobs - rank) * ncov[j, j]).
    probe (tests/parity.py only): a random generator handed to every `_ols` call (row / column permutations of the same
    regression: how far the reference's own round-off moves the result)."""
    mode = None if autolag is None else str(autolag).lower()
    if mode not in (None, "aic", "bic", "t-stat"):
        raise ValueError("autolag must be one of 'AIC', 'BIC', 't-stat' or None")
    x = np.asarray(x, dtype=np.float64)
    nobs = x.shape[0]
    ntrend = 1
    maxlag = int(np.ceil(12.0 * np.power(nobs / 100.0, 1 / 4.0)))
    maxlag = min(nobs // 2 - ntrend - 1, maxlag)
    if maxlag < 0:
        raise ValueError("sample size is too short to use selected regression component")
    d = np.diff(x)

    def design(lags):
        rows = np.arange(lags, len(d))  # t
        cols = [x[rows]] + [d[rows - j] for j in range(1, lags + 1)]
        return np.column_stack(cols), d[rows]

    Z, y = design(maxlag)
    n1 = len(y)
    with np.errstate(divide="ignore", invalid="ignore"):
        if mode is None:
            usedlag = maxlag
        else:
            full = _add_const(Z, prepend=True)
            startlag = full.shape[1] - Z.shape[1] + 1
            if mode == "t-stat":
                bestlag = startlag + maxlag
                for lag in range(startlag + maxlag, startlag - 1, -1):
                    beta, ssr, rank, ncov = _ols(y, full[:, :lag], probe)
                    tlast = beta[-1] / np.sqrt(ssr / (n1 - rank) * ncov[-1, -1])
                    bestlag = lag
                    if np.abs(tlast) >= ADF_TSTAT_STOP:
                        break
            else:
                best = None
                for lag in range(startlag, startlag + maxlag + 1):
                    _, ssr, rank, _ = _ols(y, full[:, :lag], probe)
                    llf = -n1 / 2.0 * np.log(2 * np.pi) - n1 / 2.0 * np.log(ssr / n1) - n1 / 2.0
                    ic = -2 * llf + (2 if mode == "aic" else np.log(n1)) * rank
                    if best is None or (ic, lag) < best:
                        best = (ic, lag)
                bestlag = best[1]
            usedlag = bestlag - startlag
        Z, y = design(usedlag)
        n2 = len(y)
        X = _add_const(Z[:, : usedlag + 1], prepend=False)
        beta, ssr, rank, ncov = _ols(y, X, probe)
        sigma2 = ssr / (n2 - rank)
        tstat = beta[0] / np.sqrt(sigma2 * ncov[0, 0])
    return tstat, mackinnonp_c(tstat), usedlag


def adfuller_aic(x, probe=None):
    return adfuller(x, "AIC", probe)


def autoreg_params(x, k, probe=None):
    """AutoReg(x, lags=k, trend="c").fit().params = conditional OLS; raises ValueError/ZeroDivisionError when
    statsmodels (0.12.2) cannot estimate the model (n < 2k + 2)."""
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    if k >= n:
        raise ValueError("maxlag should be < nobs")
    nobs = n - k
    if nobs < k + 1:
        raise ValueError("The model specification cannot be estimated.")
    if nobs == k + 1:
        raise ZeroDivisionError("division by zero")
    rows = np.arange(k, n)
    X = np.column_stack([np.ones(nobs)] + [x[rows - j] for j in range(1, k + 1)])
    return _ols(x[rows], X, probe)[0]


# ------------------------------------------------------------------------------------------------
# pywt.cwt(x, scales, "mexh")   (pywt/_cwt.py, pywt/_functions.py; PyWavelets 1.1.1)
# ------------------------------------------------------------------------------------------------
_MEXH_CACHE = {}


def _mexh_int_psi():
    if "v" not in _MEXH_CACHE:
        x = np.linspace(-8.0, 8.0, 1024)
        psi = (1.0 - x ** 2) * np.exp(-(x ** 2) / 2.0) * 2.0 / (np.sqrt(3.0) * np.sqrt(np.sqrt(np.pi)))
        step = x[1] - x[0]
        _MEXH_CACHE["v"] = (np.cumsum(psi) * step, x, step)
    return _MEXH_CACHE["v"]


def cwt_mexh(data, scales):
    """-> array [len(scales), len(data)] of float64 coefficients."""
    data = np.asarray(data, dtype=np.float64)
    int_psi, x, step = _mexh_int_psi()
    out = np.empty((len(scales), data.shape[0]))
    for i, scale in enumerate(scales):
        j = np.arange(scale * (x[-1] - x[0]) + 1) / (scale * step)
        j = j.astype(int)
        if j[-1] >= int_psi.size:
            j = np.extract(j < int_psi.size, j)
        int_psi_scale = int_psi[j][::-1]
        conv = np.convolve(data, int_psi_scale)
        coef = -np.sqrt(scale) * np.diff(conv)
        d = (coef.shape[-1] - data.shape[-1]) / 2.0
        if d > 0:
            coef = coef[int(np.floor(d)): -int(np.ceil(d))]
        elif d < 0:
            raise ValueError("Selected scale of {} too small.".format(scale))
        out[i] = coef
    return out


# ------------------------------------------------------------------------------------------------
# stumpy.core  (distance profiles of a query against every window of a series)
# ------------------------------------------------------------------------------------------------
STUMPY_D_SQUARED_THRESHOLD = 1e-14   # stumpy/config.py: a squared distance below it is an exact match (0.0)


def _stumpy_windows(Q, T):
    Q = np.asarray(Q, dtype=np.float64)
    T = np.asarray(T, dtype=np.float64)
    m, n = len(Q), len(T)
    if m < 3:   # core.check_window_size
        raise ValueError("All window sizes must be greater than or equal to three")
    if m > n:
        raise ValueError("The window size must be less than or equal to {}".format(n))
    return Q, np.lib.stride_tricks.sliding_window_view(T, m), m


def stumpy_mass(Q, T):
    """stumpy.core.mass(Q, T): z-normalised Euclidean distance of Q to every window of T.
    core._calculate_squared_distance: both constant -> 0; one constant -> m; else D^2 = |2 m (1 - min(rho, 1))| with rho
    the Pearson correlation of the two subsequences; D^2 < 1e-14 -> 0; a window with a non-finite sample -> inf."""
    Q, W, m = _stumpy_windows(Q, T)
    mq, sq = Q.mean(), Q.std()
    mt, st = W.mean(axis=1), W.std(axis=1)
    qconst = Q.max() == Q.min()
    tconst = W.max(axis=1) == W.min(axis=1)
    with np.errstate(all="ignore"):
        rho = ((W - mt[:, None]) * (Q - mq)[None, :]).sum(axis=1) / (m * sq * st)
        d2 = np.abs(2.0 * m * (1.0 - np.minimum(rho, 1.0)))
    d2 = np.where(tconst | qconst, float(m), d2)
    d2 = np.where(tconst & qconst, 0.0, d2)
    d2 = np.where(d2 < STUMPY_D_SQUARED_THRESHOLD, 0.0, d2)
    bad = ~np.isfinite(W).all(axis=1) | (not np.isfinite(Q).all())
    d2 = np.where(bad, np.inf, d2)
    return np.sqrt(d2)


def stumpy_mass_absolute(Q, T):
    """stumpy.core.mass_absolute(Q, T): Euclidean distance of Q to every window of T (p = 2)."""
    Q, W, m = _stumpy_windows(Q, T)
    with np.errstate(all="ignore"):
        d2 = ((W - Q[None, :]) ** 2).sum(axis=1)
    bad = ~np.isfinite(W).all(axis=1) | (not np.isfinite(Q).all())
    d2 = np.where(bad, np.inf, d2)
    return np.sqrt(d2)
