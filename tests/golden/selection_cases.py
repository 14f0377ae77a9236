"""Inputs of the feature-selection golden cases (shared by gen_golden_selection.py and the tests)."""
import numpy as np
import pandas as pd

CASES = ["binary_basic", "binary_ties", "binary_small_class_exact", "binary_independent_fdr", "multiclass3",
         "multiclass_strings", "all_constant", "shuffled_index", "regression_basic", "regression_ties",
         "regression_equal_halves", "binary_smir", "multiclass_smir"]


def _features(rng, n, y_signal):
    X = pd.DataFrame(index=np.arange(n))
    X["rel_real"] = y_signal + 0.8 * rng.standard_normal(n)
    X["rel_real2"] = -2.0 * y_signal + rng.standard_normal(n)
    X["noise1"] = rng.standard_normal(n)
    X["noise2"] = rng.standard_normal(n)
    X["ties"] = np.round(y_signal * 0.7 + rng.standard_normal(n), 0)
    X["rel_binary"] = ((y_signal + 0.5 * rng.standard_normal(n)) > 0.5).astype(float)
    X["noise_binary"] = (rng.random(n) > 0.6).astype(float) * 3.0 - 1.0
    X["const"] = 1.5
    X["three_values"] = rng.integers(0, 3, n).astype(float)
    return X


def make_case(name):
    seed = CASES.index(name) + 11
    rng = np.random.default_rng(seed)
    kw = {}
    if name == "binary_basic":
        n = 200
        y = pd.Series(rng.integers(0, 2, n))
        X = _features(rng, n, y.to_numpy().astype(float))
    elif name == "binary_ties":
        n = 333
        y = pd.Series(rng.integers(0, 2, n))
        X = _features(rng, n, y.to_numpy().astype(float))
        for c in ("rel_real", "noise1"):
            X[c] = np.round(X[c], 1)
    elif name == "binary_small_class_exact":
        n = 40
        yv = np.zeros(n, dtype=int)
        yv[rng.choice(n, 6, replace=False)] = 1
        y = pd.Series(yv)
        X = _features(rng, n, yv.astype(float))
    elif name == "binary_independent_fdr":
        n = 150
        y = pd.Series(rng.integers(0, 2, n))
        X = _features(rng, n, y.to_numpy().astype(float))
        kw = {"hypotheses_independent": True, "fdr_level": 0.1}
    elif name == "multiclass3":
        n = 300
        y = pd.Series(rng.integers(0, 3, n))
        X = _features(rng, n, y.to_numpy().astype(float))
        kw = {"multiclass": True, "n_significant": 2, "ml_task": "classification"}
    elif name == "multiclass_strings":
        n = 240
        codes = rng.integers(0, 4, n)
        y = pd.Series(np.array(["a", "b", "c", "d"], dtype=object)[codes])
        X = _features(rng, n, codes.astype(float))
        kw = {"multiclass": True, "n_significant": 1}
    elif name == "all_constant":
        n = 50
        y = pd.Series(rng.integers(0, 2, n))
        X = pd.DataFrame({"c1": np.ones(n), "c2": np.zeros(n)})
    elif name == "shuffled_index":
        n = 120
        y = pd.Series(rng.integers(0, 2, n))
        X = _features(rng, n, y.to_numpy().astype(float))
        perm = rng.permutation(n)
        X = X.iloc[perm]
        y = y.iloc[rng.permutation(n)]
    elif name == "regression_basic":
        n = 250
        yv = rng.standard_normal(n)
        y = pd.Series(yv)
        X = _features(rng, n, yv)
    elif name == "regression_ties":
        n = 400
        yv = np.round(rng.standard_normal(n), 1)
        y = pd.Series(yv)
        X = _features(rng, n, yv)
        X["rel_real"] = np.round(X["rel_real"], 1)
    elif name == "regression_equal_halves":
        n = 300
        yv = rng.standard_normal(n)
        y = pd.Series(yv)
        X = _features(rng, n, yv)
        half = np.zeros(n)
        half[rng.choice(n, n // 2, replace=False)] = 1.0   # n1 == n2: the square formula of ks_2samp
        X["half_binary"] = half
        X["rel_half"] = (yv > np.median(yv)).astype(float)
    elif name == "binary_smir":
        n = 260
        y = pd.Series(rng.integers(0, 2, n))
        X = _features(rng, n, 0.5 * y.to_numpy().astype(float))
        kw = {"test_for_binary_target_real_feature": "smir"}
    elif name == "multiclass_smir":
        n = 280
        y = pd.Series(rng.integers(0, 3, n))
        X = _features(rng, n, 0.4 * y.to_numpy().astype(float))
        kw = {"multiclass": True, "n_significant": 1, "test_for_binary_target_real_feature": "smir"}
    else:
        raise KeyError(name)
    return X, y, kw
