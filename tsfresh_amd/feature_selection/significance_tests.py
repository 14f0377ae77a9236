"""p-value tails of the reference's univariate tests, from the statistics the device returns.

tsfresh/feature_selection/significance_tests.py calls scipy once per feature; the data-dependent part of those calls
(sorting, mid-ranks, tie groups, the 2 x 2 table) is what `tsfa_relevance_classes` computes for all features at
once.  The functions here restate the remaining scalar arithmetic of
  scipy.stats.mannwhitneyu(x1, x0, use_continuity=True, alternative="two-sided")   (method="auto")
  scipy.stats.fisher_exact(table, alternative="two-sided")
and the FDR procedures of statsmodels.stats.multitest.multipletests ("fdr_bh", "fdr_by") with numpy / math only.
The reference-named single-feature functions are provided on top of the same device call.
"""
import math
import warnings

import numpy as np
import pandas as pd


def mannwhitney_pvalue(rank_sum_1, n1, n2, tie_term):
    """Two-sided p-value of scipy.stats.mannwhitneyu from the mid-rank sum of sample 1, the sample sizes and
    sum(t^3 - t) over the tie groups.  scipy's method="auto": exact distribution when a sample has <= 8 members and
    there are no ties, normal approximation with tie and continuity correction otherwise."""
    n1, n2 = int(n1), int(n2)
    u1 = float(rank_sum_1) - n1 * (n1 + 1) / 2.0
    u2 = n1 * n2 - u1
    u = max(u1, u2)
    if (n1 > 8 and n2 > 8) or tie_term > 0:
        n = n1 + n2
        mu = n1 * n2 / 2.0
        var = n1 * n2 / 12.0 * ((n + 1) - tie_term / (n * (n - 1.0)))
        s = math.sqrt(var) if var > 0 else 0.0
        num = u - mu - 0.5
        if s == 0.0:
            z = math.copysign(math.inf, num) if num != 0 else math.nan
        else:
            z = num / s
        p = math.erfc(z / math.sqrt(2.0)) if not math.isnan(z) else math.nan  # 2 * norm.sf(z)
    else:
        p = 2.0 * _mwu_exact_sf(int(u), n1, n2)
    return min(max(p, 0.0), 1.0) if not math.isnan(p) else p


def _mwu_exact_sf(k, n1, n2):
    """P(U >= k) for the Mann-Whitney statistic of samples of n1, n2 distinct values: the number of arrangements with
    statistic u is the coefficient of q^u in the Gaussian binomial [n1 + n2 choose n1]_q (exact integers)."""
    m, n = min(n1, n2), max(n1, n2)
    top = m * n
    coef = [1] + [0] * top
    for i in range(1, m + 1):
        a = n + i  # multiply by (1 - q^a)
        for d in range(top, a - 1, -1):
            coef[d] -= coef[d - a]
        for d in range(i, top + 1):  # divide by (1 - q^i)
            coef[d] += coef[d - i]
    total = sum(coef)
    k = max(0, min(int(k), top + 1))
    return sum(coef[k:]) / total


def fisher_exact_pvalue(a, b, c, d):
    """Two-sided p-value of scipy.stats.fisher_exact([[a, b], [c, d]]): the probability of all tables with the same
    margins that are at most as likely as the observed one (hypergeometric probabilities by their ratio recurrence)."""
    a, b, c, d = int(a), int(b), int(c), int(d)
    n1, n2, n = a + b, c + d, a + c
    if n1 == 0 or n2 == 0 or n == 0 or (b + d) == 0:
        return 1.0
    lo, hi = max(0, n - n2), min(n1, n)
    ks = np.arange(lo, hi + 1, dtype=np.float64)
    # log pmf(lo) by log-gamma, the rest by pmf(k + 1) / pmf(k) = (n1 - k)(n - k) / ((k + 1)(n2 - n + k + 1))
    lg = math.lgamma
    log0 = (lg(n1 + 1) - lg(lo + 1) - lg(n1 - lo + 1) + lg(n2 + 1) - lg(n - lo + 1) - lg(n2 - n + lo + 1)
            - (lg(n1 + n2 + 1) - lg(n + 1) - lg(n1 + n2 - n + 1)))
    k = ks[:-1]
    ratios = np.log((n1 - k) * (n - k)) - np.log((k + 1.0) * (n2 - n + k + 1.0))
    logp = np.concatenate([[log0], log0 + np.cumsum(ratios)])
    pm = np.exp(logp)
    pm /= pm.sum()  # removes the common error of the log-gamma anchor
    pobs = pm[a - lo]
    return float(min(pm[pm <= pobs * (1.0 + 1e-7)].sum(), 1.0))


def kendall_pvalue(n, dis, xtie, ntie, x0, x1, ytie, y0, y1):
    """Two-sided p-value of scipy.stats.kendalltau(x, y, method="asymptotic") (tau-b) from the discordant pairs and
    the tie statistics: con - dis is approximately normal with the variance of Kendall (1970), as scipy evaluates it."""
    n = int(n)
    tot = n * (n - 1) // 2
    if xtie == tot or ytie == tot:
        return math.nan
    con_minus_dis = tot - int(xtie) - int(ytie) + int(ntie) - 2 * int(dis)
    m = n * (n - 1.0)
    var = ((m * (2 * n + 5) - float(x1) - float(y1)) / 18.0 + (2.0 * int(xtie) * int(ytie)) / m
           + float(x0) * float(y0) / (9.0 * m * (n - 2)))
    z = con_minus_dis / math.sqrt(var)
    return math.erfc(abs(z) / math.sqrt(2.0))


def target_tie_statistics(y_rank):
    """(ytie, y0, y1) of scipy's count_rank_tie for the dense ranks of the target."""
    cnt = np.bincount(np.asarray(y_rank)).astype(np.int64)
    cnt = cnt[cnt > 1]
    return (int((cnt * (cnt - 1) // 2).sum()), int((cnt * (cnt - 1.0) * (cnt - 2)).sum()),
            int((cnt * (cnt - 1.0) * (2 * cnt + 5)).sum()))


def ks_2samp_pvalue(n1, n2, d):
    """Two-sided p-value of scipy.stats.ks_2samp(method="auto") from the sample sizes and the statistic: the exact
    lattice-path probability for samples of <= 10 000, the one-sample Kolmogorov distribution at n1 n2 / (n1 + n2)
    (scipy.stats.kstwo) beyond."""
    n1, n2 = int(n1), int(n2)
    if max(n1, n2) <= 10000:
        g = math.gcd(n1, n2)
        lcm = (n1 // g) * n2
        h = int(round(d * lcm))
        if h == 0:
            return 1.0
        if n1 == n2:
            prob, k = 0.0, n1 // h
            while k >= 0:  # Horner form of 2 * sum (-1)^(k-1) binom(2n, n - k h) / binom(2n, n)
                p1 = 1.0
                for j in range(h):
                    p1 = (n1 - k * h - j) * p1 / (n1 + k * h + j + 1)
                prob = p1 * (1.0 - prob)
                k -= 1
            prob = 2.0 * prob
        else:
            from tsfresh_amd import _native
            prob = _native.ks_outer_prob(n1, n2, g, h)
        if 0.0 <= prob <= 1.0:
            return prob
    from scipy.stats import kstwo  # a special function of two scalars (the reference's own dependency)
    big, small = max(float(n1), float(n2)), min(float(n1), float(n2))
    en = big * small / (big + small)
    return float(min(max(kstwo.sf(d, np.round(en)), 0.0), 1.0))


def fdr_reject(pvalues, alpha, independent):
    """statsmodels.stats.multitest.multipletests(pvalues, alpha, "fdr_bh" if independent else "fdr_by")[0]."""
    p = np.asarray(pvalues, dtype=np.float64)
    m = len(p)
    if m == 0:
        return np.zeros(0, dtype=bool)
    order = np.argsort(p)
    ps = p[order]
    factor = np.arange(1, m + 1) / float(m)
    if not independent:
        factor = factor / np.sum(1.0 / np.arange(1, m + 1))
    ok = ps <= factor * alpha
    rej_sorted = np.zeros(m, dtype=bool)
    if ok.any():
        rej_sorted[: np.nonzero(ok)[0].max() + 1] = True
    out = np.empty(m, dtype=bool)
    out[order] = rej_sorted
    return out


# ---- the reference's single-feature functions (significance_tests.py:43-188), on the same device call ----

def _check_series(x, y):
    if not isinstance(x, pd.Series):
        raise TypeError("x should be a pandas Series")
    if not isinstance(y, pd.Series):
        raise TypeError("y should be a pandas Series")
    if not y.index.equals(x.index):
        raise ValueError("X and y need to have the same index!")
    if np.isnan(np.asarray(x.values, dtype=float)).any():
        raise ValueError("Feature {} contains NaN values".format(x.name))
    if np.isnan(np.asarray(y.values, dtype=float)).any():
        raise ValueError("Target contains NaN values")


def _check_binary_target(y):
    vals = set(y)
    if vals != {0, 1}:
        if len(vals) > 2:
            raise ValueError("Target is not binary!")
        warnings.warn("The binary target should have values 1 and 0 (or True and False). Instead found" + str(vals),
                      RuntimeWarning)


def _single(x, y, device):
    from tsfresh_amd import _native
    yv = np.asarray(y.values)
    y0, y1 = np.unique(yv)
    codes = (yv == y1).astype(np.int32)
    nu, lo, hi, tie, rs, hc = _native.relevance_classes(np.asarray(x.values, dtype=np.float64).reshape(-1, 1), codes, 2,
                                                        device=device)
    n1 = int(codes.sum())
    return nu[0], tie[0], rs[0], hc[0], n1, len(codes) - n1


def target_binary_feature_real_test(x, y, test="mann", device=0):
    """significance_tests.py:84 -- Mann-Whitney U of the feature split by the binary target."""
    _check_series(x, y)
    _check_binary_target(y)
    if test not in ("mann", "smir"):
        raise ValueError("Please use a valid entry for test_for_binary_target_real_feature. "
                         "Valid entries are 'mann' and 'smir'.")
    if test == "smir":
        from tsfresh_amd import _native
        yv = np.asarray(y.values)
        codes = (yv == np.unique(yv)[1]).astype(np.int32)
        ks_d = _native.relevance_classes(np.asarray(x.values, dtype=np.float64).reshape(-1, 1), codes, 2, device=device,
                                         with_ks=True)[6]
        n1 = int(codes.sum())
        return ks_2samp_pvalue(n1, len(codes) - n1, ks_d[0, 1])
    _, tie, rs, _, n1, n0 = _single(x, y, device)
    return mannwhitney_pvalue(rs[1], n1, n0, tie)


def target_binary_feature_binary_test(x, y, device=0):
    """significance_tests.py:43 -- Fisher's exact test of the 2 x 2 table."""
    _check_series(x, y)
    vals = set(x)
    if vals != {0, 1}:
        if len(vals) > 2:
            raise ValueError("[target_binary_feature_binary_test] Feature is not binary!")
        warnings.warn("A binary feature should have only values 1 and 0 (incl. True and False). Instead found "
                      + str(vals) + " in feature ''" + str(x.name) + "''.", RuntimeWarning)
    _check_binary_target(y)
    _, _, _, hc, n1, n0 = _single(x, y, device)
    return fisher_exact_pvalue(hc[1], n1 - hc[1], hc[0], n0 - hc[0])
