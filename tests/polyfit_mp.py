"""TEST INFRASTRUCTURE: 60-digit adjudicator for np.polyfit on ill-conditioned bin means (mpmath).

`np.polyfit` (numpy/lib/_polynomial_impl.py) scales the Vandermonde columns to unit norm and solves with LAPACK gelsd,
rcond = len(x) * eps: singular values <= rcond * s_max are dropped and the minimum-norm solution of the rest is
returned.  `exact_polyfit` evaluates exactly that definition -- the float64 scaled design numpy forms, its SVD, the cut,
the minimum-norm solution -- in 60-digit arithmetic.  The reference's float64 result deviates from it by about
eps * s_max / s_min(kept) (`kappa_kept`), the kernels' double-double pass (fam_langevin_dd.h) by far less; a singular
value within a factor `BAND` of the cut makes the reference's RANK decision a function of round-off (measured on 1525
random series with a singular value within 4x of the cut: gelsd's rank differed from the 60-digit one twice, both
within 0.3 % of the cut; BAND = 1.1 leaves a margin for a last-bit difference in the bin means).
"""
import mpmath as mp
import numpy as np
import pandas as pd

EPS = float(np.finfo(np.float64).eps)
BAND = 1.1


def bin_means(x, r):
    """x_mean / y_mean of fc.py:131-173 (_estimate_friedrich_coefficients), or None when qcut fails."""
    x = np.asarray(x, dtype=np.float64)
    df = pd.DataFrame({"signal": x[:-1], "delta": np.diff(x)})
    try:
        df["quantiles"] = pd.qcut(df.signal, r)
    except (ValueError, IndexError):
        return None
    g = df.groupby("quantiles", observed=False)
    res = pd.DataFrame({"x_mean": g.signal.mean(), "y_mean": g.delta.mean()}).dropna()
    return res.x_mean.to_numpy(), res.y_mean.to_numpy()


def scaled_design(xm, deg):
    lhs = np.vander(xm, deg + 1)
    scale = np.sqrt((lhs * lhs).sum(axis=0))
    return lhs / scale, scale


def singular_ratios(xm, deg):
    """s_i / s_max of the scaled design (float64 SVD: good to ~eps absolute, which is all the callers need)."""
    A, _ = scaled_design(np.asarray(xm, dtype=np.float64), deg)
    s = np.linalg.svd(A, compute_uv=False)
    return s / s[0]


def exact_polyfit(xm, ym, deg, dps=60):
    """-> (coefficients, s / s_max in 60 digits, rank)."""
    old = mp.mp.dps
    mp.mp.dps = dps
    try:
        A, scale = scaled_design(np.asarray(xm, dtype=np.float64), deg)
        k = len(xm)
        rcond = k * EPS
        U, S, V = mp.svd_r(mp.matrix(A.tolist()))
        smax = max(S)
        y = mp.matrix([float(v) for v in ym])
        c = [mp.mpf(0)] * (deg + 1)
        rank = 0
        for i in range(len(S)):
            if S[i] > rcond * smax:
                rank += 1
                w = sum(U[j, i] * y[j] for j in range(k)) / S[i]
                for j in range(deg + 1):
                    c[j] += V[i, j] * w
        coef = np.array([float(c[j] / mp.mpf(float(scale[j]))) for j in range(deg + 1)])
        return coef, np.array([float(s / smax) for s in S]), rank
    finally:
        mp.mp.dps = old


def polyfit_conditioning(xm, deg):
    """-> (kappa_kept, in_band): s_max / smallest KEPT singular value, and whether any singular value lies within
    BAND of the cut (then the reference's rank is decided by LAPACK round-off)."""
    s = singular_ratios(xm, deg)
    rcond = len(xm) * EPS
    kept = s[s > rcond]
    in_band = bool(np.any((s > rcond / BAND) & (s < rcond * BAND)))
    return (1.0 / kept[-1] if len(kept) else np.inf), in_band
