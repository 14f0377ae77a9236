"""Imputation helpers against the REAL reference's output (tests/golden/ref_impute.json)."""
import json
import os
import sys
import warnings

import numpy as np
import pandas as pd
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, G)

from tsfresh_amd.utilities import dataframe_functions as ours  # noqa: E402


def _frame():
    rng = np.random.default_rng(11)
    x = rng.standard_normal((12, 6)).round(3)
    x[1, 0] = np.nan; x[5, 0] = np.inf; x[7, 0] = -np.inf
    x[:, 1] = np.nan
    x[2, 2] = np.inf; x[3, 2] = np.inf
    x[0, 3] = -np.inf; x[4, 3] = np.nan; x[9, 3] = np.nan
    x[:, 4] = [np.inf, -np.inf, np.nan] * 4
    return pd.DataFrame(x, columns=list("abcdef"), index=np.arange(100, 112))


def test_impute_matches_the_reference():
    golden = json.load(open(os.path.join(G, "ref_impute.json")))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        mx, mn, med = ours.get_range_values_per_column(_frame())
    assert any("did not have any finite values" in str(x.message) for x in w)
    for name, got in (("max", mx), ("min", mn), ("median", med)):
        assert {k: float(v) for k, v in got.items()} == golden["range"][name]
    df = _frame()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = ours.impute(df)
    assert res is df and df.values.tolist() == golden["impute"] and np.isfinite(df.values).all()
    df = _frame()
    assert ours.impute_dataframe_zero(df).values.tolist() == golden["impute_zero"]


def test_impute_errors():
    df = _frame()
    with pytest.raises(ValueError, match="more or less keys"):
        ours.impute_dataframe_range(df, {"a": 1.0}, {"a": 0.0}, {"a": 0.5})
    full = {c: 0.0 for c in df.columns}
    with pytest.raises(ValueError, match="non finite values"):
        ours.impute_dataframe_range(df, dict(full, a=np.inf), full, full)
    with pytest.raises(ValueError, match="must not contain NaN"):
        ours.check_for_nans_in_columns(df)
    assert len(ours.impute(df.iloc[:0])) == 0
