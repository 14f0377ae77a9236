"""The packer's pre-sorted fast path (tsfresh_amd/feature_extraction/data.py:_pack_presorted) must produce exactly what
the general factorize + lexsort path produces, and must decline every layout it does not cover."""
import numpy as np
import pandas as pd
import pytest

from tsfresh_amd.feature_extraction import data as D


def _both(df, **kw):
    fast, _, _ = D.pack_timeseries(df, **kw)
    orig = D._pack_presorted
    D._pack_presorted = lambda *a, **k: None
    try:
        gen, _, _ = D.pack_timeseries(df, **kw)
    finally:
        D._pack_presorted = orig
    return fast, gen


def _same(a, b):
    assert len(a) == len(b)
    for p, q in zip(a, b):
        assert p.kind == q.kind
        assert np.array_equal(p.ids, q.ids) and np.array_equal(p.offsets, q.offsets)
        assert p.values.dtype == q.values.dtype and np.array_equal(p.values, q.values)
        assert (p.sort is None) == (q.sort is None) and (p.sort is None or np.array_equal(p.sort, q.sort))
        assert (p.times is None) == (q.times is None) and (p.times is None or np.array_equal(p.times, q.times))


@pytest.mark.parametrize("case", ["sorted", "ragged", "float_ids", "unsorted_within", "unsorted_ids", "string_ids",
                                  "single_id", "datetime"])
def test_presorted_fast_path_equals_general_path(case):
    rng = np.random.default_rng(3)
    lens = rng.integers(1, 9, size=17) if case == "ragged" else np.full(17, 6)
    ids = np.repeat(np.arange(17) * 3, lens)
    t = np.concatenate([np.arange(m) for m in lens])
    df = pd.DataFrame({"id": ids, "time": t, "value": rng.standard_normal(len(ids)).astype(np.float32)})
    if case == "float_ids":
        df["id"] = df["id"].astype(float) / 2
    if case == "unsorted_within":
        df.loc[3, "time"] = 99
    if case == "unsorted_ids":
        df = df.iloc[::-1].reset_index(drop=True)
    if case == "string_ids":
        df["id"] = df["id"].map(lambda v: "s%03d" % v)
    if case == "single_id":
        df["id"] = 7
        df["time"] = np.arange(len(df))
    if case == "datetime":
        df.index = pd.to_datetime("2020-01-01") + pd.to_timedelta(np.arange(len(df)) * 37, unit="min")
    fast, gen = _both(df, column_id="id", column_sort="time")
    _same(fast, gen)
    fast, gen = _both(df.drop(columns="time"), column_id="id")
    _same(fast, gen)


def test_presorted_declines_other_layouts():
    ids = np.array([1, 1, 2, 2, 1])
    assert D._pack_presorted("v", ids, np.zeros(5), None, None) is None           # id comes back later
    ids = np.array([1, 1, 2, 2])
    assert D._pack_presorted("v", ids, np.zeros(4), np.array([0, 1, 1, 0]), None) is None  # descent inside a group
    assert D._pack_presorted("v", ids, np.zeros(4), np.array([0, 1, 0, 1]), None) is not None
    assert D._pack_presorted("v", np.array(["a", "a", "b"]), np.zeros(3), None, None) is None


def test_arrow_tables_pack_like_frames():
    pa = pytest.importorskip("pyarrow")
    rng = np.random.default_rng(8)
    lens = rng.integers(1, 7, size=11)
    df = pd.DataFrame({"id": np.repeat(np.arange(11), lens), "time": np.concatenate([np.arange(m) for m in lens]),
                       "value": rng.standard_normal(int(lens.sum())).astype(np.float32)})
    want, _, _ = D.pack_timeseries(df, column_id="id", column_sort="time")
    for container in (pa.Table.from_pandas(df, preserve_index=False),
                      pa.Table.from_pandas(df, preserve_index=False).to_batches()[0]):
        got, id_dtype, has_dt = D.pack_timeseries(container, column_id="id", column_sort="time")
        _same(got, want)
        assert id_dtype == df["id"].dtype and not has_dt


@pytest.mark.parametrize("case", ["sorted", "ragged", "float_ids", "unsorted_within", "unsorted_ids", "datetime_sort",
                                  "int32", "no_sort", "int_values", "int_values_unsorted", "float64_values"])
def test_native_scan_equals_numpy_passes(case, monkeypatch):
    """tsfa_pack_scan (one multi-threaded C++ pass: layout proof + group boundaries + NaN check) against the numpy
    passes it replaces, on frames large enough to split over several scan threads."""
    rng = np.random.default_rng(5)
    n_ids = 3000
    lens = rng.integers(1, 700, size=n_ids) if case == "ragged" else np.full(n_ids, 300)
    ids = np.repeat(np.arange(n_ids) * 2, lens)
    t = np.concatenate([np.arange(m) for m in lens])
    df = pd.DataFrame({"id": ids, "time": t, "value": rng.standard_normal(len(ids)).astype(np.float32)})
    if case == "float_ids":
        df["id"] = df["id"].astype(float) / 2
    if case == "unsorted_within":
        df.loc[len(df) // 2 + 1, "time"] = 10 ** 6 if df.loc[len(df) // 2 + 2, "id"] == df.loc[len(df) // 2 + 1, "id"] else -1
    if case == "unsorted_ids":
        df = df.iloc[::-1].reset_index(drop=True)
    if case == "datetime_sort":
        df["time"] = pd.to_datetime("2021-03-04") + pd.to_timedelta(df["time"].to_numpy(), unit="s")
    if case == "int32":
        df["id"] = df["id"].astype(np.int32)
        df["time"] = df["time"].astype(np.int32)
    if case.startswith("int_values"):   # no NaN check to run, converted only once the layout is proven
        df["value"] = rng.integers(-50, 50, size=len(df))
        if case.endswith("unsorted"):
            df = df.iloc[::-1].reset_index(drop=True)
    if case == "float64_values":        # taken as they are: no copy
        df["value"] = df["value"].astype(np.float64)
    kw = dict(column_id="id") if case == "no_sort" else dict(column_id="id", column_sort="time")
    if case == "no_sort":
        df = df.drop(columns="time")
    assert len(df) >= D._NATIVE_SCAN_MIN_ROWS
    native, _, _ = D.pack_timeseries(df, **kw)
    monkeypatch.setattr(D, "_NATIVE_SCAN_MIN_ROWS", 1 << 62)
    plain, _, _ = D.pack_timeseries(df, **kw)
    _same(native, plain)


def test_native_scan_reports_nan_values_like_the_reference():
    n = D._NATIVE_SCAN_MIN_ROWS + 10
    df = pd.DataFrame({"id": np.arange(n) // 100, "time": np.arange(n) % 100, "value": np.zeros(n, dtype=np.float32)})
    df.loc[n - 3, "value"] = np.nan
    with pytest.raises(ValueError, match="Column must not contain NaN values: value"):
        D.pack_timeseries(df, column_id="id", column_sort="time")
    small = df.iloc[-50:].reset_index(drop=True)
    with pytest.raises(ValueError, match="Column must not contain NaN values: value"):
        D.pack_timeseries(small, column_id="id", column_sort="time")


@pytest.mark.parametrize("dtype", [np.float16, np.longdouble])
def test_native_scan_sees_nan_in_float16_and_longdouble_columns(dtype):
    """Round-3 ADVICE (low): element types other than float32 / float64 that CAN hold a NaN are converted before the
    native scan, so a large frame raises the reference's ValueError (data.py:124-178) instead of yielding NaN features."""
    n = D._NATIVE_SCAN_MIN_ROWS + 10
    v = np.zeros(n, dtype=dtype)
    v[n - 3] = np.nan
    df = pd.DataFrame({"id": np.arange(n) // 100, "time": np.arange(n) % 100, "value": v})
    with pytest.raises(ValueError, match="Column must not contain NaN values: value"):
        D.pack_timeseries(df, column_id="id", column_sort="time")
    df.loc[n - 3, "value"] = 1.5
    packed, _, _ = D.pack_timeseries(df, column_id="id", column_sort="time")
    assert packed[0].values.dtype == np.float64 and packed[0].values[n - 3] == 1.5


def test_arrow_wide_table_is_packed_without_a_dataframe(monkeypatch):
    pa = pytest.importorskip("pyarrow")
    rng = np.random.default_rng(9)
    lens = rng.integers(1, 9, size=40)
    df = pd.DataFrame({"id": np.repeat(np.arange(40), lens), "time": np.concatenate([np.arange(m) for m in lens]),
                       "a": rng.standard_normal(int(lens.sum())).astype(np.float32), "b": rng.standard_normal(int(lens.sum()))})
    want, _, _ = D.pack_timeseries(df, column_id="id", column_sort="time")
    monkeypatch.setattr(D, "_arrow_to_frame", lambda t: (_ for _ in ()).throw(AssertionError("took the pandas route")))
    table = pa.Table.from_pandas(df, preserve_index=False)
    got, id_dtype, has_dt = D.pack_timeseries(table, column_id="id", column_sort="time")
    _same(got, want)
    assert got[0].values.base is not None  # a view of the Arrow buffer, not a copy
    assert id_dtype == df["id"].dtype and not has_dt
    bad = pa.table({"id": df["id"].to_numpy(), "time": df["time"].to_numpy(), "a": np.where(np.arange(len(df)) == 3, np.nan, df["a"].to_numpy())})
    with pytest.raises(ValueError, match="Column must not contain NaN values: a"):
        D.pack_timeseries(bad, column_id="id", column_sort="time")
