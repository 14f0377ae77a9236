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


# ---------------------------------------------------------------------------------------------------------------------
# statsmodels' pinv semantics in many digits: singular values <= 1e-15 s_max dropped, rank = #{s > s_max p eps}
# ---------------------------------------------------------------------------------------------------------------------
def pinv_ols_mp(X, y, dps=120):
    """OLS(y, X).fit(method="pinv") evaluated in `dps` digits from the eigen-decomposition of X^T X (the squared
    condition number needs the digits).  -> (beta, ssr, rank, cov00, s / s_max) as floats / float arrays."""
    old = mp.mp.dps
    mp.mp.dps = dps
    try:
        n, p = len(X), len(X[0])
        Xm = [[mp.mpf(float(v)) for v in r] for r in X]
        ym = [mp.mpf(float(v)) for v in y]
        G = mp.matrix(p, p)
        g = mp.matrix(p, 1)
        for a in range(p):
            for c in range(a + 1):
                G[a, c] = G[c, a] = mp.fsum(Xm[t][a] * Xm[t][c] for t in range(n))
            g[a] = mp.fsum(Xm[t][a] * ym[t] for t in range(n))
        lam, V = mp.eigsy(G)
        lmax = max(lam)
        smax = mp.sqrt(lmax)
        eps = mp.mpf(2) ** -52
        beta = [mp.mpf(0)] * p
        cov00 = mp.mpf(0)
        rank = 0
        ratios = []
        for i in range(p):
            s = mp.sqrt(lam[i]) if lam[i] > 0 else mp.mpf(0)
            ratios.append(float(s / smax))
            if s > smax * p * eps:
                rank += 1
            if s > mp.mpf("1e-15") * smax:
                w = mp.fsum(V[a, i] * g[a] for a in range(p)) / lam[i]
                for a in range(p):
                    beta[a] += V[a, i] * w
                cov00 += V[0, i] ** 2 / lam[i]
        ssr = mp.fsum((ym[t] - mp.fsum(Xm[t][a] * beta[a] for a in range(p))) ** 2 for t in range(n))
        return [float(v) for v in beta], float(ssr), rank, float(cov00), sorted(ratios, reverse=True)
    finally:
        mp.mp.dps = old


def autoreg_params_pinv_mp(x, k):
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    rows = np.arange(k, n)
    X = np.column_stack([np.ones(n - k)] + [x[rows - j] for j in range(1, k + 1)])
    return pinv_ols_mp(X.tolist(), x[rows].tolist())
