"""Feature selection.  CPU: the oracle and the host-side p-value tails against the golden tables of
the real reference and against scipy.  GPU: `calculate_relevance_table` / `select_features` through the C-ABI against
the same golden tables and against the oracle on larger matrices."""
import json
import math
import os
import sys

import numpy as np
import pandas as pd
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, os.path.dirname(HERE))

from selection_cases import CASES, make_case  # noqa: E402

GOLD = {r["name"]: r for r in json.load(open(os.path.join(HERE, "golden", "ref_selection.json")))}


def _frames(rec):
    X = pd.DataFrame({c: v for c, v in rec["X"].items()}, index=rec["index"])
    y = pd.Series(rec["y"], index=rec["y_index"])
    return X, y


def _check_against_golden(rec, got_rows):
    """got_rows: feature -> dict of column values."""
    cols = [c for c in rec["columns"] if c != "feature"]
    assert sorted(got_rows) == sorted(rec["table_index"])
    for r, feat in enumerate(rec["table_index"]):
        for c in cols:
            want = rec["table"][c][r]
            got = got_rows[feat].get(c)
            if isinstance(want, bool):
                assert bool(got) == want, (rec["name"], feat, c, got, want)
            elif want is None:
                assert got is None or (isinstance(got, float) and math.isnan(got)), (rec["name"], feat, c, got)
            elif isinstance(want, float):
                ks_col = (rec["table"]["type"][r] == "binary" and rec["name"].startswith("regression")) or \
                         (rec["table"]["type"][r] == "real" and rec["name"].endswith("smir"))
                if c.startswith("p_value") and ks_col:
                    # the golden tables come from scipy 1.7.1, whose exact Kolmogorov-Smirnov tail is 1 - P(inside): an
                    # absolute rounding error of a few 1e-16 (scipy >= 1.9 and this package compute the complement
                    # directly and keep the relative accuracy of small tails)
                    assert abs(got - want) <= 1e-9 * want + 5e-15, (rec["name"], feat, c, got, want)
                else:
                    assert got == pytest.approx(want, rel=1e-9, abs=1e-300), (rec["name"], feat, c, got, want)
            else:
                assert str(got) == want, (rec["name"], feat, c, got, want)


@pytest.mark.parametrize("name", CASES)
def test_case_inputs_are_reproducible(name):
    X, y, _ = make_case(name)
    gX, gy = _frames(GOLD[name])
    assert list(X.index) == list(gX.index) and list(y.index) == list(gy.index)
    assert np.array_equal(X.to_numpy(dtype=float), gX[list(X.columns)].to_numpy(dtype=float))
    assert (np.array_equal(y.to_numpy(), gy.to_numpy()) if y.dtype.kind == "f" else [str(v) for v in y] == [str(v) for v in gy])


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_the_reference_tables(name):
    from oracle.selection import relevance_table
    rec = GOLD[name]
    X, y = _frames(rec)
    _check_against_golden(rec, relevance_table(X, y, **rec["kwargs"]))


def test_host_tails_match_scipy():
    from scipy import stats
    from tsfresh_amd.feature_selection.significance_tests import fdr_reject, fisher_exact_pvalue, mannwhitney_pvalue
    from oracle.selection import fdr
    rng = np.random.default_rng(1)
    for trial in range(200):
        n1, n2 = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        x, z = rng.standard_normal(n1), rng.standard_normal(n2)
        if trial % 2:
            x, z = np.round(x, 0), np.round(z, 0)
        allv = np.concatenate([x, z])
        _, cnt = np.unique(allv, return_counts=True)
        p = mannwhitney_pvalue(stats.rankdata(allv)[:n1].sum(), n1, n2, float(np.sum(cnt.astype(float) ** 3 - cnt)))
        q = stats.mannwhitneyu(x, z, use_continuity=True, alternative="two-sided").pvalue
        assert (math.isnan(p) and math.isnan(q)) or p == pytest.approx(q, rel=1e-12, abs=1e-300)
        a, b, c, d = [int(v) for v in rng.integers(0, 50, 4)]
        if trial % 5 == 0:
            a, c = b, d
        assert fisher_exact_pvalue(a, b, c, d) == pytest.approx(stats.fisher_exact([[a, b], [c, d]])[1], rel=1e-10)
        pv = rng.random(int(rng.integers(1, 30))) ** 3
        for ind in (True, False):
            assert np.array_equal(fdr_reject(pv, 0.1, ind), fdr(pv, 0.1, ind))


def test_ks_tail_matches_scipy():
    from scipy import stats
    from tsfresh_amd.feature_selection.significance_tests import kendall_pvalue, ks_2samp_pvalue, target_tie_statistics
    rng = np.random.default_rng(2)
    for trial in range(60):
        n1, n2 = int(rng.integers(2, 400)), int(rng.integers(2, 400))
        if trial % 4 == 0:
            n2 = n1
        a, b = rng.standard_normal(n1) + 0.3 * (trial % 3), rng.standard_normal(n2)
        r = stats.ks_2samp(a, b)
        assert ks_2samp_pvalue(n1, n2, float(r.statistic)) == pytest.approx(r.pvalue, rel=1e-10, abs=1e-300)
    r = stats.ks_2samp(rng.standard_normal(12000), rng.standard_normal(9000) + 0.02)  # beyond the exact range
    assert ks_2samp_pvalue(12000, 9000, float(r.statistic)) == pytest.approx(r.pvalue, rel=1e-12)
    # Kendall tail from brute-force statistics
    for trial in range(20):
        n = int(rng.integers(5, 60))
        x, y = np.round(rng.standard_normal(n), 1), np.round(rng.standard_normal(n), 1)
        dis = sum(1 for i in range(n) for j in range(n) if x[i] < x[j] and y[i] > y[j])
        def ties(v):
            _, cnt = np.unique(v, return_counts=True)
            cnt = cnt[cnt > 1].astype(float)
            return int((cnt * (cnt - 1) // 2).sum()), float((cnt * (cnt - 1) * (cnt - 2)).sum()), float((cnt * (cnt - 1) * (2 * cnt + 5)).sum())
        xtie, x0, x1 = ties(x)
        _, cj = np.unique(np.stack([x, y], 1), axis=0, return_counts=True)
        ntie = int((cj * (cj - 1) // 2).sum())
        ytie, y0, y1 = target_tie_statistics(np.unique(y, return_inverse=True)[1])
        want = stats.kendalltau(x, y, method="asymptotic").pvalue
        got = kendall_pvalue(n, dis, xtie, ntie, x0, x1, ytie, y0, y1)
        assert (math.isnan(want) and math.isnan(got)) or got == pytest.approx(want, rel=1e-12)


def test_invalid_options_raise_the_reference_errors():
    from tsfresh_amd.feature_selection import calculate_relevance_table
    X = pd.DataFrame({"a": [1.0, 2.0, 3.0, 4.0]})
    with pytest.raises(ValueError):
        calculate_relevance_table(X, pd.Series([0, 1, 0, 1]), test_for_binary_target_real_feature="other")
    with pytest.raises(ValueError):
        calculate_relevance_table(X, pd.Series([0, 1, 0, 1]), ml_task="ranking")


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_relevance_table_matches_the_reference(name):
    from tsfresh_amd.feature_selection import calculate_relevance_table, select_features
    rec = GOLD[name]
    X, y = _frames(rec)
    tab = calculate_relevance_table(X, y, **rec["kwargs"])
    assert list(tab.columns) == rec["columns"]
    rows = {str(f): {c: (None if (isinstance(v, float) and math.isnan(v)) else v) for c, v in row.items()}
            for f, row in tab.set_index(tab.index.astype(str)).iterrows()}
    _check_against_golden(rec, rows)
    # row order: ascending p-value inside the tested block (ties may permute), constants last
    if "p_value" in tab.columns and name != "all_constant":
        p = tab["p_value"].to_numpy()
        k = int(np.sum(~np.isnan(p)))
        assert np.all(np.diff(p[:k]) >= 0) and np.all(np.isnan(p[k:]))
    if name != "all_constant":
        kw = {k: v for k, v in rec["kwargs"].items()}
        sel = select_features(X, y, **kw)
        want = [f for f, r in zip(rec["table_index"], rec["table"]["relevant"]) if r]
        assert sorted(sel.columns) == sorted(want)
        assert sel.index.equals(X.index)


@pytest.mark.gpu
@pytest.mark.parametrize("n,m,classes", [(5000, 40, 2), (20000, 25, 5), (70000, 12, 3)])
def test_gpu_relevance_table_matches_the_oracle_at_scale(n, m, classes):
    from oracle.selection import relevance_table
    from tsfresh_amd.feature_selection import calculate_relevance_table
    rng = np.random.default_rng(n)
    yv = rng.integers(0, classes, n)
    X = pd.DataFrame(rng.standard_normal((n, m)), columns=["f%d" % i for i in range(m)])
    X["f0"] += 0.05 * yv
    X["f1"] = np.round(X["f1"] + 0.1 * yv, 1)          # ties
    X["f2"] = (rng.random(n) < 0.3 + 0.02 * yv) * 1.0   # binary
    X["f3"] = 2.0                                       # constant
    X["f4"] = rng.integers(0, 4, n).astype(float)       # four values
    y = pd.Series(yv)
    kw = {"multiclass": classes > 2, "n_significant": 2} if classes > 2 else {}
    tab = calculate_relevance_table(X, y, **kw)
    want = relevance_table(X, y, **kw)
    for f in X.columns:
        row = tab.loc[f]
        for c, w in want[f].items():
            g = row[c]
            if isinstance(w, (bool, np.bool_)):
                assert bool(g) == bool(w), (f, c, g, w)
            elif isinstance(w, str):
                assert g == w
            elif isinstance(w, float) and math.isnan(w):
                assert math.isnan(g)
            else:
                assert g == pytest.approx(w, rel=1e-9, abs=1e-300), (f, c, g, w)


@pytest.mark.gpu
def test_gpu_extract_relevant_features_end_to_end():
    """extract -> impute -> select on series whose level depends on the class: the level features must survive, and
    the selected frame must equal select_features(extract_features(...))."""
    from tsfresh_amd import MinimalFCParameters, extract_features, extract_relevant_features, select_features
    from tsfresh_amd.utilities.dataframe_functions import impute
    rng = np.random.default_rng(4)
    n_ids, L = 60, 40
    y = pd.Series(rng.integers(0, 2, n_ids), index=np.arange(n_ids) + 100)
    rows = []
    for i, sid in enumerate(y.index):
        rows.append(pd.DataFrame({"id": sid, "time": np.arange(L), "a": rng.standard_normal(L) + 2.0 * y.iloc[i],
                                  "b": rng.standard_normal(L)}))
    df = pd.concat(rows, ignore_index=True)
    Xr = extract_relevant_features(df, y, column_id="id", column_sort="time", default_fc_parameters=MinimalFCParameters())
    Xe = extract_features(df, column_id="id", column_sort="time", default_fc_parameters=MinimalFCParameters(),
                          impute_function=impute)
    Xs = select_features(Xe, y)
    assert list(Xr.columns) == list(Xs.columns) and Xr.equals(Xs)
    assert "a__mean" in Xr.columns and "a__median" in Xr.columns
    assert not any(c.startswith("b__") for c in Xr.columns if c not in ("b__length",))
    with pytest.raises(ValueError):
        extract_relevant_features(df, y.iloc[:-1], column_id="id", column_sort="time",
                                  default_fc_parameters=MinimalFCParameters())


@pytest.mark.gpu
def test_gpu_relevance_statistics_are_exact_with_infinities_and_ties():
    """The C-ABI statistics against scipy.stats.rankdata / np.unique: heavy ties, +-inf values (which tie with the
    padding of the device sort), one row, non-power-of-two row counts."""
    from scipy.stats import rankdata
    from tsfresh_amd import _native
    rng = np.random.default_rng(9)
    for n, m, C in [(1, 2, 1), (7, 3, 2), (2049, 6, 3), (5000, 5, 2), (33000, 4, 4)]:
        X = rng.standard_normal((n, m))
        X[:, 0] = np.round(X[:, 0], 0)
        if n > 4:
            X[rng.choice(n, max(1, n // 10), replace=False), 1] = np.inf
            X[rng.choice(n, max(1, n // 10), replace=False), 1] = -np.inf
        y = rng.integers(0, C, n).astype(np.int32)
        nu, lo, hi, tie, rs, hc = _native.relevance_classes(X, y, C)
        for c in range(m):
            r = rankdata(X[:, c])
            u, cnt = np.unique(X[:, c], return_counts=True)
            assert nu[c] == len(u) and lo[c] == u[0] and hi[c] == u[-1]
            assert tie[c] == float(np.sum(cnt.astype(float) ** 3 - cnt))
            for k in range(C):
                assert rs[c, k] == r[y == k].sum()
                assert hc[c, k] == np.sum((X[:, c] == u[-1]) & (y == k))


@pytest.mark.gpu
@pytest.mark.parametrize("n,m", [(3000, 30), (12000, 16), (60000, 10)])
def test_gpu_regression_table_matches_the_oracle_at_scale(n, m):
    from oracle.selection import relevance_table
    from tsfresh_amd.feature_selection import calculate_relevance_table
    rng = np.random.default_rng(n + 1)
    yv = np.round(rng.standard_normal(n), 2)
    X = pd.DataFrame(rng.standard_normal((n, m)), columns=["f%d" % i for i in range(m)])
    X["f0"] += 0.03 * yv
    X["f1"] = np.round(X["f1"] + 0.05 * yv, 1)
    X["f2"] = (rng.random(n) < 0.4 + 0.02 * np.tanh(yv)) * 1.0
    X["f3"] = -1.0
    X["f4"] = rng.integers(0, 3, n).astype(float)
    X["f5"] = (yv > 0.1).astype(float)  # strongly dependent binary feature: a tiny Kolmogorov-Smirnov tail
    y = pd.Series(yv)
    tab = calculate_relevance_table(X, y)
    want = relevance_table(X, y)
    for f in X.columns:
        for c, w in want[f].items():
            g = tab.loc[f][c]
            if isinstance(w, (bool, np.bool_)):
                assert bool(g) == bool(w), (f, c, g, w)
            elif isinstance(w, str):
                assert g == w
            elif isinstance(w, float) and math.isnan(w):
                assert math.isnan(g)
            else:
                assert g == pytest.approx(w, rel=1e-9, abs=1e-300), (f, c, g, w)


@pytest.mark.gpu
def test_gpu_relevance_column_batches(monkeypatch):
    """The column loop of the C-ABI calls (sort scratch bounded to a few GB): forced to batches of 3 columns, every
    statistic must equal the single-batch result."""
    from tsfresh_amd import _native
    rng = np.random.default_rng(12)
    n, m = 3000, 11
    X = rng.standard_normal((n, m))
    X[:, 4] = (X[:, 4] > 0) * 1.0
    X[:, 7] = np.round(X[:, 7], 1)
    yc = rng.integers(0, 3, n).astype(np.int32)
    yr = np.round(rng.standard_normal(n), 2)
    _native.set_library_option("relevance_batch", 0)
    a = _native.relevance_classes(X, yc, 3, with_ks=True)
    ar, _ = _native.relevance_real(X, yr)
    _native.set_library_option("relevance_batch", 3)
    try:
        b = _native.relevance_classes(X, yc, 3, with_ks=True)
        br, _ = _native.relevance_real(X, yr)
    finally:
        _native.set_library_option("relevance_batch", 0)
    for u, v in zip(a, b):
        assert np.array_equal(u, v)
    for f in ar.dtype.names:
        assert np.array_equal(ar[f], br[f]), f


@pytest.mark.gpu
@pytest.mark.parametrize("task", ["classification", "regression"])
def test_device_resident_chain_equals_the_host_chain(gpu, task):
    """extract_relevant_features(device_resident=True): the feature matrix stays in HBM from tsfa_extract through
    tsfa_impute and tsfa_relevance_* and only the selected columns come back (tsfa_gather_columns) -- same frame as the
    default chain (relevant_extraction.py:18: extract -> impute -> select), two kinds, NaN / inf cells for impute."""
    from tsfresh_amd import EfficientFCParameters, extract_relevant_features
    rng = np.random.default_rng(8)
    n_ids = 120
    frames = []
    label = rng.integers(0, 2, n_ids)
    for i in range(n_ids):
        n = int(rng.integers(3, 40)) if i % 7 == 0 else int(rng.integers(60, 160))  # short series: NaN / inf features
        shift = 0.8 * label[i]
        frames.append(pd.DataFrame({"id": i, "t": np.arange(n), "a": rng.standard_normal(n) + shift,
                                    "b": np.cumsum(rng.standard_normal(n)) * (1.0 + shift)}))
    df = pd.concat(frames, ignore_index=True)
    if task == "classification":
        y = pd.Series(label, index=np.arange(n_ids))
    else:
        y = pd.Series(label * 1.5 + 0.3 * rng.standard_normal(n_ids), index=np.arange(n_ids))
    params = EfficientFCParameters()
    kw = dict(column_id="id", column_sort="t", default_fc_parameters=params, fdr_level=0.2)
    want = extract_relevant_features(df, y, **kw)
    got = extract_relevant_features(df, y, device_resident=True, **kw)
    assert list(got.columns) == list(want.columns) and len(want.columns) > 5
    assert list(got.index) == list(want.index) and got.index.dtype == want.index.dtype
    assert np.array_equal(got.to_numpy(), want.to_numpy())
    # and through X (the merge of the reference's signature)
    Xin = pd.DataFrame({"extra": rng.standard_normal(n_ids)}, index=np.arange(n_ids))
    got2 = extract_relevant_features(df, y, X=Xin, device_resident=True, **kw)
    assert list(got2.columns) == ["extra"] + list(want.columns)
