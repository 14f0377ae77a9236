"""`calculate_relevance_table` (tsfresh/feature_selection/relevance.py:31) on the GPU.

The reference maps a scipy test over the columns of X (`_calculate_relevance_table_for_implicit_target`,
relevance.py:325) once per class label.  Here ONE call of `tsfa_relevance_classes` yields, for all columns and all
labels together, the feature types (relevance.py:396), the mid-rank sums and tie terms (Mann-Whitney U) and the
2 x 2 counts (Fisher); the tables are then assembled exactly as the reference does (column names, row order,
multiclass merge, constant features, warnings).
"""
import functools
import warnings

import numpy as np
import pandas as pd

from tsfresh_amd import _native
from tsfresh_amd.feature_selection.significance_tests import (fdr_reject, fisher_exact_pvalue, kendall_pvalue,
                                                             ks_2samp_pvalue, mannwhitney_pvalue, target_tie_statistics)


def infer_ml_task(y):
    """relevance.py:353: integer / object / string targets -> 'classification', everything else 'regression'."""
    if y.dtype.kind in np.typecodes["AllInteger"] or y.dtype == object or isinstance(y.dtype, pd.StringDtype):
        return "classification"
    return "regression"


def get_feature_type(feature_column):
    """relevance.py:396: 'constant', 'binary' or 'real' by the number of distinct values (host helper for one column;
    calculate_relevance_table takes the counts of all columns from the device)."""
    n_unique = len(set(feature_column.values))
    return "constant" if n_unique == 1 else ("binary" if n_unique == 2 else "real")


def combine_relevance_tables(relevance_tables):
    """relevance.py:377: a feature is relevant if any table says so; its p-value is the smallest one."""
    def _combine(a, b):
        a.relevant |= b.relevant
        a.p_value = a.p_value.combine(b.p_value, min, 1)
        return a
    return functools.reduce(_combine, relevance_tables)


def _default_device():
    from tsfresh_amd.feature_extraction.extraction import _default_device as d
    return d()


def calculate_relevance_table(X, y, ml_task="auto", multiclass=False, n_significant=1, n_jobs=None, show_warnings=False,
                              chunksize=None, test_for_binary_target_binary_feature="fisher",
                              test_for_binary_target_real_feature="mann", test_for_real_target_binary_feature="mann",
                              test_for_real_target_real_feature="kendall", fdr_level=0.05, hypotheses_independent=False,
                              device=None):
    """Signature, return frame and warnings of the reference (relevance.py:31-322); `n_jobs` / `chunksize` steer the
    reference's process pool and are ignored."""
    y = y.sort_index()
    X = X.sort_index()
    assert list(y.index) == list(X.index), "The index of X and y need to be the same"
    return _relevance_table(X, X.columns, y, ml_task, multiclass, n_significant, show_warnings,
                            test_for_binary_target_real_feature, fdr_level, hypotheses_independent, device)


def _relevance_table(X, columns, y, ml_task, multiclass, n_significant, show_warnings, test_for_binary_target_real_feature,
                     fdr_level, hypotheses_independent, device):
    """The body of calculate_relevance_table.  X: a DataFrame, or a `_native.DeviceMatrix` whose rows follow y's sorted
    index and whose columns are named by `columns` (the device-resident chain of extract_relevant_features: imputed on
    the device, hence free of NaN)."""
    if ml_task not in ["auto", "classification", "regression"]:
        raise ValueError("ml_task must be one of: 'auto', 'classification', 'regression'")
    elif ml_task == "auto":
        ml_task = infer_ml_task(y)
    if multiclass:
        assert ml_task == "classification", "ml_task must be classification for multiclass problem"
        assert len(y.unique()) >= n_significant, "n_significant must not exceed the total number of classes"
        if len(y.unique()) <= 2:
            warnings.warn("Two or fewer classes, binary feature selection will be used (multiclass = False)")
            multiclass = False
    if ml_task == "classification" and test_for_binary_target_real_feature not in ("mann", "smir"):
        raise ValueError("Please use a valid entry for test_for_binary_target_real_feature. "
                         "Valid entries are 'mann' and 'smir'.")
    smir = (test_for_binary_target_real_feature == "smir")

    columns = pd.Index(columns)
    if isinstance(X, _native.DeviceMatrix):
        values = X
        if device is None:
            device = X.device
    else:
        values = np.ascontiguousarray(X.to_numpy(dtype=np.float64))
        if np.isnan(values).any():
            bad = columns[np.isnan(values).any(axis=0)][0]
            raise ValueError("Feature {} contains NaN values".format(bad))
    if y.dtype.kind == "f" and np.isnan(y.to_numpy()).any():
        raise ValueError("Target contains NaN values")
    if device is None:
        device = _default_device()
    n = len(y)
    if ml_task == "classification":
        labels = list(y.unique())  # order of first appearance, as the reference iterates
        codes = pd.Categorical(y, categories=labels).codes.astype(np.int32)
        stats_ = _native.relevance_classes(values, codes, len(labels), device=device, with_ks=smir)
        n_unique, _, _, tie_term, rank_sums, hi_counts = stats_[:6]
        ks_d = stats_[6] if smir else None
        class_n = np.bincount(codes, minlength=len(labels))
        hi_total = hi_counts.sum(axis=1)
    else:
        real_cols, y_rank = _native.relevance_real(values, y.to_numpy(dtype=np.float64), device=device)
        n_unique = real_cols["n_unique"]
        ytie, y0, y1 = target_tie_statistics(y_rank)

    with warnings.catch_warnings():
        warnings.simplefilter("default" if show_warnings else "ignore")
        relevance_table = pd.DataFrame(index=pd.Series(columns, name="feature"))
        relevance_table["feature"] = relevance_table.index
        relevance_table["type"] = pd.Series(
            np.where(n_unique == 1, "constant", np.where(n_unique == 2, "binary", "real")), index=relevance_table.index)
        pos = {f: i for i, f in enumerate(columns)}
        table_real = relevance_table[relevance_table.type == "real"].copy()
        table_binary = relevance_table[relevance_table.type == "binary"].copy()
        table_const = relevance_table[relevance_table.type == "constant"].copy()
        table_const["p_value"] = np.nan
        table_const["relevant"] = False
        if not table_const.empty:
            warnings.warn("[test_feature_significance] Constant features: {}".format(
                ", ".join(map(str, table_const.feature))), RuntimeWarning)
        if len(table_const) == len(relevance_table):
            return table_const

        if ml_task == "regression":
            # relevance.py:303-316: Kendall's tau for the real features, Kolmogorov-Smirnov of the target split by a
            # binary feature; one table
            t_real, t_bin = table_real.copy(), table_binary.copy()
            t_real["p_value"] = pd.Series(
                [kendall_pvalue(n, real_cols["dis"][pos[f]], real_cols["xtie"][pos[f]], real_cols["ntie"][pos[f]],
                                real_cols["x0"][pos[f]], real_cols["x1"][pos[f]], ytie, y0, y1) for f in t_real.index],
                index=t_real.index, dtype=float)
            t_bin["p_value"] = pd.Series(
                [ks_2samp_pvalue(real_cols["n_hi"][pos[f]], n - real_cols["n_hi"][pos[f]], real_cols["ks_d"][pos[f]])
                 for f in t_bin.index], index=t_bin.index, dtype=float)
            relevance_table = pd.concat([t_real, t_bin])
            relevance_table["relevant"] = fdr_reject(relevance_table.p_value.to_numpy(), fdr_level, hypotheses_independent)
            relevance_table = relevance_table.sort_values("p_value")
            labels = []
        tables = []
        for k, label in enumerate(labels):
            n1, n0 = int(class_n[k]), int(n - class_n[k])
            t_real, t_bin = table_real.copy(), table_binary.copy()
            if smir:  # significance_tests.py:121: ks_2samp(x[y == label], x[y != label])
                t_real["p_value"] = pd.Series([ks_2samp_pvalue(n1, n0, ks_d[pos[f], k]) for f in t_real.index],
                                              index=t_real.index, dtype=float)
            else:
                t_real["p_value"] = pd.Series(
                    [mannwhitney_pvalue(rank_sums[pos[f], k], n1, n0, tie_term[pos[f]]) for f in t_real.index],
                    index=t_real.index, dtype=float)
            # [[y1 & x1, y1 & x0], [y0 & x1, y0 & x0]], x1 = the larger of the feature's two values (np.unique order)
            t_bin["p_value"] = pd.Series(
                [fisher_exact_pvalue(hi_counts[pos[f], k], n1 - hi_counts[pos[f], k], hi_total[pos[f]] - hi_counts[pos[f], k],
                                     n0 - (hi_total[pos[f]] - hi_counts[pos[f], k])) for f in t_bin.index],
                index=t_bin.index, dtype=float)
            tmp = pd.concat([t_real, t_bin])
            tmp["relevant"] = fdr_reject(tmp.p_value.to_numpy(), fdr_level, hypotheses_independent)
            tmp = tmp.sort_values("p_value")
            if multiclass:
                tmp = tmp.reset_index(drop=True)
                tmp.columns = tmp.columns.map(lambda c: c + "_" + str(label) if c != "feature" and c != "type" else c)
            tables.append(tmp)

        if ml_task == "regression":
            pass
        elif multiclass:
            relevance_table = functools.reduce(
                lambda left, right: pd.merge(left, right, on=["feature", "type"], how="outer"), tables)
            relevance_table["n_significant"] = relevance_table.filter(regex="^relevant_", axis=1).sum(axis=1)
            relevance_table["relevant"] = relevance_table["n_significant"] >= n_significant
            relevance_table.index = relevance_table["feature"]
            for column in relevance_table.filter(regex="^relevant_", axis=1).columns:
                table_const[column] = False
            table_const["n_significant"] = 0
            table_const.drop(columns=["p_value"], inplace=True)
        else:
            relevance_table = combine_relevance_tables(tables)
        relevance_table = pd.concat([relevance_table, table_const], axis=0)
        if sum(relevance_table["relevant"]) == 0:
            warnings.warn(
                "No feature was found relevant for {} for fdr level = {} (which corresponds to the maximal percentage "
                "of irrelevant features, consider using an higher fdr level or add other features.".format(
                    ml_task, fdr_level), RuntimeWarning)
    return relevance_table
