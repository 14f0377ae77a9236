"""TEST INFRASTRUCTURE ONLY (imported by tests/ only; the product never imports oracle/).

CPU restatement of the reference's relevance table
(tsfresh/feature_selection/relevance.py:31-350, significance_tests.py:43-188): one scipy call per feature and label,
exactly like the reference, plus statsmodels' FDR procedures (stats/multitest.py: fdrcorrection, "indep" / "negcorr")
restated in numpy.  Pinned by tests/golden/ref_selection.json, which the REAL reference produced
(tests/golden/gen_golden_selection.py).
"""
import numpy as np
import pandas as pd
from scipy import stats


def fdr(pvals, alpha, independent):
    p = np.asarray(pvals, dtype=float)
    m = len(p)
    srt = np.argsort(p)
    ps = p[srt]
    ecdf = np.arange(1, m + 1) / float(m)
    if not independent:  # Benjamini-Yekutieli
        ecdf = ecdf / np.sum(1.0 / np.arange(1, m + 1))
    rej = ps <= ecdf * alpha
    if rej.any():
        rej[: max(np.nonzero(rej)[0])] = True
    out = np.empty(m, dtype=bool)
    out[srt] = rej
    return out


def _regression_table(X, y, types, fdr_level, hypotheses_independent):
    # relevance.py:303-316 with significance_tests.py:135 (ks_2samp) and :170 (kendalltau, asymptotic)
    tested = [f for f in X.columns if types[f] == "real"] + [f for f in X.columns if types[f] == "binary"]
    pv = []
    for f in tested:
        x = X[f]
        if types[f] == "real":
            pv.append(stats.kendalltau(x, y, method="asymptotic")[1])
        else:
            x0, x1 = np.unique(x.values)
            pv.append(stats.ks_2samp(y[x == x1], y[x == x0])[1])
    rej = fdr(pv, fdr_level, hypotheses_independent) if tested else []
    out = {f: {"type": types[f]} for f in X.columns}
    for i, f in enumerate(tested):
        out[f]["p_value"] = float(pv[i])
        out[f]["relevant"] = bool(rej[i])
    for f in X.columns:
        if types[f] == "constant":
            out[f]["p_value"] = np.nan
            out[f]["relevant"] = False
    return out


def relevance_table(X, y, multiclass=False, n_significant=1, fdr_level=0.05, hypotheses_independent=False, ml_task="auto",
                    test_for_binary_target_real_feature="mann", **_):
    """-> dict feature -> dict(type, p_value..., relevant...) with the reference's column names."""
    y = y.sort_index()
    X = X.sort_index()
    if ml_task == "auto":
        ml_task = "classification" if (y.dtype.kind in "iub" or y.dtype == object) else "regression"
    if ml_task == "regression":
        types = {}
        for f in X.columns:
            nu = len(set(X[f].values))
            types[f] = "constant" if nu == 1 else ("binary" if nu == 2 else "real")
        return _regression_table(X, y, types, fdr_level, hypotheses_independent)
    labels = list(y.unique())
    if multiclass and len(labels) <= 2:
        multiclass = False
    types = {}
    for f in X.columns:
        nu = len(set(X[f].values))
        types[f] = "constant" if nu == 1 else ("binary" if nu == 2 else "real")
    tested = [f for f in X.columns if types[f] == "real"] + [f for f in X.columns if types[f] == "binary"]
    out = {f: {"type": types[f]} for f in X.columns}
    per_label = {}
    for label in labels:
        yb = (y == label)
        pv = []
        for f in tested:
            x = X[f]
            if types[f] == "real":
                if test_for_binary_target_real_feature == "smir":
                    pv.append(stats.ks_2samp(x[yb], x[~yb])[1])
                else:
                    pv.append(stats.mannwhitneyu(x[yb], x[~yb], use_continuity=True, alternative="two-sided").pvalue)
            else:
                x0, x1 = np.unique(x.values)
                a = int(np.sum(yb[x == x1])); b = int(np.sum(yb[x == x0]))
                c = int(np.sum(x == x1)) - a; d = int(np.sum(x == x0)) - b
                pv.append(stats.fisher_exact([[a, b], [c, d]], alternative="two-sided")[1])
        rej = fdr(pv, fdr_level, hypotheses_independent) if tested else []
        per_label[label] = {f: (pv[i], bool(rej[i])) for i, f in enumerate(tested)}
    for f in X.columns:
        if types[f] == "constant":
            if multiclass:
                for label in labels:
                    out[f]["relevant_" + str(label)] = False
                out[f]["n_significant"] = 0
            else:
                out[f]["p_value"] = np.nan
            out[f]["relevant"] = False
            continue
        if multiclass:
            ns = 0
            for label in labels:
                p, r = per_label[label][f]
                out[f]["p_value_" + str(label)] = p
                out[f]["relevant_" + str(label)] = r
                ns += int(r)
            out[f]["n_significant"] = ns
            out[f]["relevant"] = ns >= n_significant
        else:
            out[f]["p_value"] = min(per_label[label][f][0] for label in labels)
            out[f]["relevant"] = any(per_label[label][f][1] for label in labels)
    return out
