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


@pytest.mark.gpu
def test_device_impute_matches_the_reference_and_the_numpy_path(gpu):
    """tsfa_impute (column sort in HBM: max / min / median of the finite values, in-place patch) against the real
    reference's output (ref_impute.json) and, on a matrix large enough to take the device route through impute(),
    against the numpy restatement."""
    from tsfresh_amd import _native
    golden = json.load(open(os.path.join(G, "ref_impute.json")))
    x = np.ascontiguousarray(_frame().to_numpy())
    mx, mn, med, cnt = _native.impute_matrix(x, device=0)
    cols = list("abcdef")
    assert dict(zip(cols, mx.tolist())) == golden["range"]["max"]
    assert dict(zip(cols, mn.tolist())) == golden["range"]["min"]
    assert dict(zip(cols, med.tolist())) == golden["range"]["median"]
    assert x.tolist() == golden["impute"] and cnt.tolist()[1] == 0
    rng = np.random.default_rng(5)
    big = rng.standard_normal((5000, 40))
    big[rng.random(big.shape) < 0.02] = np.nan
    big[rng.random(big.shape) < 0.01] = np.inf
    big[rng.random(big.shape) < 0.01] = -np.inf
    big[:, 7] = np.nan
    big[:, 9] = 3.0
    df_dev = pd.DataFrame(big.copy(), columns=["c%d" % i for i in range(40)])
    df_cpu = df_dev.copy()
    assert df_dev.size >= ours._DEVICE_IMPUTE_MIN_CELLS
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        ours.impute(df_dev)
    assert any("did not have any finite values" in str(m.message) for m in w)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mx, mn, med = ours.get_range_values_per_column(df_cpu)
        ours.impute_dataframe_range(df_cpu, mx, mn, med)
    assert np.isfinite(df_dev.to_numpy()).all()
    assert np.array_equal(df_dev.to_numpy(), df_cpu.to_numpy())


@pytest.mark.gpu
def test_device_impute_on_a_row_strided_view_touches_only_its_own_cells(gpu):
    """a row-strided host view (ld > n_cols) that ENDS with its last row -- tsfa_impute used to copy
    n_rows * ld doubles in both directions: out of bounds behind the last row, and stale data over the cells between
    the rows.  Now the rows are staged dense (hipMemcpy2D)."""
    from tsfresh_amd import _native
    rng = np.random.default_rng(9)
    n_rows, n_cols, ld = 300, 7, 12
    buf = np.full((n_rows - 1) * ld + n_cols, 777.0)
    view = np.lib.stride_tricks.as_strided(buf, shape=(n_rows, n_cols), strides=(8 * ld, 8))
    data = rng.standard_normal((n_rows, n_cols))
    data[rng.random(data.shape) < 0.05] = np.nan
    data[rng.random(data.shape) < 0.03] = np.inf
    data[rng.random(data.shape) < 0.03] = -np.inf
    view[:] = data
    want = pd.DataFrame(data.copy())
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mx, mn, med = ours.get_range_values_per_column(want)
        ours.impute_dataframe_range(want, mx, mn, med)
    _native.impute_matrix(view, device=0)
    assert np.array_equal(view, want.to_numpy())
    between = np.ones(len(buf), dtype=bool)
    for r in range(n_rows):
        between[r * ld:r * ld + n_cols] = False
    assert np.all(buf[between] == 777.0)
