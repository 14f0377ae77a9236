"""TEST INFRASTRUCTURE: statsmodels' adfuller(autolag="AIC") and AutoReg(trend="c") evaluated in 60-digit arithmetic
(mpmath) on the float64 data -- "what the reference's algorithm returns when its SVD does not run out of digits".
Used to adjudicate between the kernels' double-double pass and the oracle's float64 SVD on ill-conditioned designs.
Full-rank designs only (a dependent column raises)."""
import math

import mpmath as mp
import numpy as np

mp.mp.dps = 60


def _ls(X, y):
    """-> (beta, ssr, (X^T X)^-1) by Cholesky of the normal equations in 60 digits."""
    n, p = len(X), len(X[0])
    G = mp.matrix(p, p)
    g = mp.matrix(p, 1)
    for a in range(p):
        for c in range(a + 1):
            G[a, c] = G[c, a] = mp.fsum(X[t][a] * X[t][c] for t in range(n))
        g[a] = mp.fsum(X[t][a] * y[t] for t in range(n))
    Ginv = G ** -1
    beta = Ginv * g
    ssr = mp.fsum((y[t] - mp.fsum(X[t][a] * beta[a] for a in range(p))) ** 2 for t in range(n))
    return beta, ssr, Ginv


def adfuller_aic_mp(x):
    x = [mp.mpf(float(v)) for v in x]
    n = len(x)
    maxlag = min(n // 2 - 2, int(math.ceil(12.0 * (n / 100.0) ** 0.25)))
    d = [x[i + 1] - x[i] for i in range(n - 1)]
    d = [mp.mpf(float(v)) for v in d]  # np.diff rounds to float64

    def design(lags, const_first):
        rows = range(lags, len(d))
        Z = [[x[t]] + [d[t - j] for j in range(1, lags + 1)] for t in rows]
        y = [d[t] for t in rows]
        cols = list(zip(*Z))
        has_const = any(all(v == c[0] for v in c) and c[0] != 0 for c in cols)
        if not has_const:
            Z = [([mp.mpf(1)] + r) if const_first else (r + [mp.mpf(1)]) for r in Z]
        return Z, y, (0 if has_const else 1)

    Z, y, hc = design(maxlag, True)
    startlag = hc + 1
    nobs = len(y)
    best = None
    for lag in range(startlag, startlag + maxlag + 1):
        _, ssr, _ = _ls([r[:lag] for r in Z], y)
        aic = nobs * mp.log(ssr / nobs) + nobs * (mp.log(2 * mp.pi) + 1) + 2 * lag
        if best is None or (aic, lag) < best:
            best = (aic, lag)
    usedlag = best[1] - startlag
    Z, y, hc = design(usedlag, False)
    beta, ssr, Ginv = _ls(Z, y)
    sigma2 = ssr / (len(y) - len(Z[0]))
    return float(beta[0] / mp.sqrt(sigma2 * Ginv[0, 0])), usedlag


def autoreg_params_mp(x, k):
    x = [mp.mpf(float(v)) for v in x]
    n = len(x)
    X = [[mp.mpf(1)] + [x[t - j] for j in range(1, k + 1)] for t in range(k, n)]
    beta, _, _ = _ls(X, [x[t] for t in range(k, n)])
    return np.array([float(b) for b in beta])
