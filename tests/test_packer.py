"""The packer's pre-sorted fast path (tsfresh_amd/feature_extraction/data.py:_pack_presorted) must produce exactly what
the general factorize + lexsort path produces, and must decline every layout it does not cover."""
import numpy as np
import pandas as pd
import pytest

from tsfresh_amd.feature_extraction import data as D


def _both(df, **kw):
    fast, _, _ = D.pack_timeseries(df, **kw)
    orig = D._pack_presorted
    D._pack_presorted = lambda *a: None
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
