"""`select_features` (tsfresh/feature_selection/selection.py:17): keep the columns of X the relevance table accepts."""
import numpy as np
import pandas as pd

from tsfresh_amd.feature_selection.relevance import calculate_relevance_table


def select_features(X, y, test_for_binary_target_binary_feature="fisher", test_for_binary_target_real_feature="mann",
                    test_for_real_target_binary_feature="mann", test_for_real_target_real_feature="kendall",
                    fdr_level=0.05, hypotheses_independent=False, n_jobs=None, show_warnings=False, chunksize=None,
                    ml_task="auto", multiclass=False, n_significant=1, device=None):
    """Same arguments, checks and result as the reference (selection.py:150-181)."""
    assert isinstance(X, pd.DataFrame), "Please pass features in X as pandas.DataFrame."
    assert isinstance(y, (pd.Series, np.ndarray)), \
        "The type of target vector y must be one of: pandas.Series, numpy.ndarray"
    assert len(y) > 1, "y must contain at least two samples."
    assert len(X) == len(y), "X and y must contain the same number of samples."
    assert len(set(y)) > 1, "Feature selection is only possible if more than 1 label/class is provided"
    if isinstance(y, pd.Series) and set(X.index) != set(y.index):
        raise ValueError("Index of X and y must be identical if provided")
    if isinstance(y, np.ndarray):
        y = pd.Series(y, index=X.index)
    relevance_table = calculate_relevance_table(
        X, y, ml_task=ml_task, multiclass=multiclass, n_significant=n_significant, n_jobs=n_jobs,
        show_warnings=show_warnings, chunksize=chunksize,
        test_for_binary_target_real_feature=test_for_binary_target_real_feature, fdr_level=fdr_level,
        hypotheses_independent=hypotheses_independent, device=device)
    relevant_features = relevance_table[relevance_table.relevant].feature
    return X.loc[:, relevant_features]
