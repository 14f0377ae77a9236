"""GPUDistributor: the `map_reduce` a reference `extract_features(distributor=...)` would
call, fed with the reference's chunk shape `(id, kind, pd.Series)`."""
from collections import namedtuple

import numpy as np
import pandas as pd
import pytest

from tsfresh_amd.utilities.distribution import GPUDistributor

Timeseries = namedtuple("Timeseries", ["id", "kind", "data"])  # tsfresh/feature_extraction/data.py:53


def _chunks():
    rng = np.random.default_rng(4)
    out = []
    for sid in (7, 3, 5):
        for kind in ("a", "b"):
            out.append(Timeseries(sid, kind, pd.Series(rng.standard_normal(50 + sid))))
    return out


def test_rejects_foreign_distributors_and_accepts_its_own():
    from tsfresh_amd import extract_features
    df = pd.DataFrame({"id": [1, 1], "time": [0, 1], "x": [1.0, 2.0]})
    with pytest.raises(ValueError, match="not an DistributorBaseClass"):
        extract_features(df, column_id="id", column_sort="time", distributor=object())
    d = GPUDistributor(device=0)
    assert d.device == 0 and d.close() is None


@pytest.mark.gpu
def test_any_distributor_object_is_accepted_and_ignored_with_a_warning(gpu):
    """extraction.py:285-286 type-checks `distributor` and nothing else: a user who passes the reference's
    MultiprocessingDistributor keeps a working call (VERDICT r5 missing #6)."""
    from tsfresh_amd import MinimalFCParameters, extract_features

    class SomeClusterDistributor:          # the DistributorBaseClass interface (utilities/distribution.py:64-104)
        def map_reduce(self, map_function, data, function_kwargs=None, chunk_size=None, data_length=None):
            raise AssertionError("the GPU path must not map chunks over a foreign distributor")

        def close(self):
            pass

    rng = np.random.default_rng(0)
    df = pd.DataFrame({"id": np.repeat([1, 2, 3], 20), "time": np.tile(np.arange(20), 3), "x": rng.standard_normal(60)})
    want = extract_features(df, column_id="id", column_sort="time", default_fc_parameters=MinimalFCParameters())
    with pytest.warns(UserWarning, match="is ignored"):
        got = extract_features(df, column_id="id", column_sort="time", default_fc_parameters=MinimalFCParameters(),
                               distributor=SomeClusterDistributor())
    pd.testing.assert_frame_equal(got, want)


@pytest.mark.gpu
def test_map_reduce_returns_the_reference_tuples(gpu):
    from tsfresh_amd import MinimalFCParameters, extract_features
    chunks = _chunks()
    params = MinimalFCParameters()
    tuples = GPUDistributor().map_reduce(None, data=iter(chunks),
                                         function_kwargs={"default_fc_parameters": params,
                                                          "kind_to_fc_parameters": {"b": {"maximum": None}}})
    assert all(len(t) == 3 for t in tuples)
    got = {(t[0], t[1]): t[2] for t in tuples}
    assert len(got) == 3 * (10 + 1)
    for c in chunks:
        assert got[(c.id, c.kind + "__maximum")] == np.max(c.data.to_numpy())
        if c.kind == "a":
            x = c.data.to_numpy()
            assert abs(got[(c.id, "a__sum_values")] - np.sum(x)) <= 1e-13 * np.sum(np.abs(x))   # k_stream: tree sum
    # same numbers as the DataFrame route
    df = pd.concat([pd.DataFrame({"id": c.id, "kind": c.kind, "t": np.arange(len(c.data)), "v": c.data.to_numpy()})
                    for c in chunks])
    feats = extract_features(df, column_id="id", column_sort="t", column_kind="kind", column_value="v",
                             default_fc_parameters=params, kind_to_fc_parameters={"b": {"maximum": None}},
                             distributor=GPUDistributor(device=0))
    for (sid, name), v in got.items():
        assert feats.loc[sid, name] == v


def _load_real_tsfresh():
    """Stub-load /root/reference as tests/golden/gen_golden_main.py does (pywt / statsmodels / stumpy are missing in the
    main interpreter); None when the reference tree is absent (the GPU box)."""
    import os
    import sys
    import types
    if not os.path.isdir("/root/reference/tsfresh"):
        return None

    class _Raiser(types.ModuleType):
        def __getattr__(self, item):
            if item.startswith("__"):
                raise AttributeError(item)

            def _fail(*a, **k):
                raise RuntimeError("stubbed third-party module %s.%s was called" % (self.__name__, item))
            return _fail
    for mod in ("pywt", "stumpy", "statsmodels", "statsmodels.tools", "statsmodels.tools.sm_exceptions", "statsmodels.tsa",
                "statsmodels.tsa.ar_model", "statsmodels.tsa.stattools", "statsmodels.stats", "statsmodels.stats.multitest"):
        sys.modules.setdefault(mod, _Raiser(mod))
    sys.modules["statsmodels.tools.sm_exceptions"].MissingDataError = type("MissingDataError", (Exception,), {})
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
    import tsfresh
    return tsfresh


def test_the_real_reference_pipeline_drives_the_distributor():
    """tsfresh.extract_features(distributor=...) -- the REFERENCE's own to_tsdata / _do_extraction / pivot
    (extraction.py:193-305, data.py:86-121) -- with a GPUDistributor whose matrix comes from the g++ build of the kernel
    sources (no GPU in this container; the class under test is the product's, only `_extract_matrix` is swapped).  The
    frame must carry the reference's columns and index, and the oracle's numbers, by column name
    (tests/units/feature_extraction/test_extraction.py:292-318 style).  Build container only."""
    tsfresh = _load_real_tsfresh()
    if tsfresh is None:
        pytest.skip("/root/reference is not present on this box")
    import importlib
    import warnings

    import tsfresh_amd.utilities.distribution as dmod
    importlib.reload(dmod)  # pick up tsfresh's DistributorBaseClass now that tsfresh is importable
    from tsfresh.feature_extraction import extract_features as ref_extract_features
    from tsfresh.feature_extraction import settings as ref_settings
    from tsfresh.utilities.distribution import DistributorBaseClass

    from emul_lib import emul_extract
    from engines import oracle_engine
    from parity import compare

    class EmulDistributor(dmod.GPUDistributor):
        def _extract_matrix(self, fc_parameters, values, offsets, times, has_dt):
            names, matrix = emul_extract(fc_parameters, values, offsets, kind="k", times=times)
            return [n[len("k__"):] for n in names], matrix

    assert issubclass(dmod.GPUDistributor, DistributorBaseClass)
    rng = np.random.default_rng(12)
    lens = [60, 75, 64, 90, 128]
    ids = [11, 3, 7, 5, 9]
    frames = []
    for sid, n in zip(ids, lens):
        frames.append(pd.DataFrame({"id": sid, "time": np.arange(n), "a": rng.standard_normal(n),
                                    "b": np.cumsum(rng.standard_normal(n))}))
    df = pd.concat(frames, ignore_index=True)
    params = ref_settings.ComprehensiveFCParameters()
    for k in ("cwt_coefficients", "agg_autocorrelation", "partial_autocorrelation", "augmented_dickey_fuller", "ar_coefficient"):
        del params[k]  # their third-party modules are stubs here; the oracle comparison below does not need them
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = ref_extract_features(df, column_id="id", column_sort="time", default_fc_parameters=params,
                                   distributor=EmulDistributor(), disable_progressbar=True)
        want_ref = ref_extract_features(df, column_id="id", column_sort="time", default_fc_parameters=params, n_jobs=0,
                                        disable_progressbar=True)
    assert list(got.index) == sorted(ids) and got.index.dtype == want_ref.index.dtype
    assert set(got.columns) == set(want_ref.columns) and got.shape == want_ref.shape
    # the numbers: against the reference's own serial run, column by column
    for kind in ("a", "b"):
        cols = [c for c in want_ref.columns if c.startswith(kind + "__")]
        series = [df.loc[df.id == sid, kind].to_numpy() for sid in sorted(ids)]
        names = ["value__" + c[len(kind) + 2:] for c in cols]
        bad = compare(names, got[cols].to_numpy(), want_ref[cols].to_numpy(), series, simd_golden=True)
        assert not bad, (kind, bad[:6])


@pytest.mark.gpu
def test_partition_binding_returns_the_reference_long_rows(gpu):
    """tsfresh_amd.convenience.bindings.feature_extraction_on_partition: one pandas partition in, the long
    (id, variable, value) frame of the reference's dask helper out (bindings.py:9-60), equal to the pivoted call."""
    from tsfresh_amd import MinimalFCParameters, extract_features
    from tsfresh_amd.convenience.bindings import dask_feature_extraction_on_chunk, feature_extraction_on_partition
    chunks = _chunks()
    df = pd.concat([pd.DataFrame({"id": c.id, "kind": c.kind, "t": np.arange(len(c.data)), "v": c.data.to_numpy()})
                    for c in chunks], ignore_index=True)
    long = feature_extraction_on_partition(df, column_id="id", column_kind="kind", column_value="v", column_sort="t",
                                           default_fc_parameters=MinimalFCParameters())
    assert list(long.columns) == ["id", "variable", "value"] and long["value"].dtype == np.float64
    wide = long.pivot_table(index="id", columns="variable", values="value", aggfunc="mean")
    want = extract_features(df, column_id="id", column_sort="t", column_kind="kind", column_value="v",
                            default_fc_parameters=MinimalFCParameters())
    assert sorted(wide.columns) == sorted(want.columns)
    assert np.array_equal(wide[want.columns].to_numpy(), want.to_numpy())
    assert len(feature_extraction_on_partition(df.iloc[:0], "id", "kind", "v", "t", MinimalFCParameters())) == 0
    try:
        import dask  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="needs dask"):
            dask_feature_extraction_on_chunk(df, "id", "kind", "v", "t", MinimalFCParameters())


@pytest.mark.gpu
def test_map_reduce_over_comprehensive_parameters_matches_the_oracle(gpu):
    """N4 on the device: the reference's chunk tuples through GPUDistributor.map_reduce with
    ComprehensiveFCParameters, every (id, column) value against the oracle by column NAME."""
    from engines import oracle_engine
    from parity import compare
    from tsfresh_amd import ComprehensiveFCParameters
    rng = np.random.default_rng(17)
    Chunk = type(_chunks()[0])
    chunks, series = [], {}
    for sid in range(5):
        for kind, n in (("a", 300), ("b", 128)):
            x = rng.standard_normal(n) if kind == "a" else np.cumsum(rng.standard_normal(n))
            chunks.append(Chunk(sid, kind, pd.Series(x)))
            series[(sid, kind)] = x
    params = ComprehensiveFCParameters()
    tuples = GPUDistributor().map_reduce(None, data=iter(chunks), function_kwargs={"default_fc_parameters": params,
                                                                                   "kind_to_fc_parameters": None})
    got = {(t[0], t[1]): float(t[2]) for t in tuples}
    for kind in ("a", "b"):
        ids = sorted({c.id for c in chunks})
        xs = [series[(i, kind)] for i in ids]
        values = np.concatenate(xs)
        offsets = np.concatenate([[0], np.cumsum([len(x) for x in xs])]).astype(np.int64)
        names, want = oracle_engine(params, values, offsets, kind=kind)
        assert all((i, n) in got for i in ids for n in names), "a column is missing from the tuples"
        mat = np.array([[got[(i, n)] for n in names] for i in ids])
        bad = compare(["value__" + n.split("__", 1)[1] for n in names], mat, want, xs)
        assert not bad, bad[:8]
    assert len(got) == 10 * 783


class _StandInDaskFrame:
    """What dask_feature_extraction_on_chunk needs of a dask DataFrame: column access for the meta frame and
    map_partitions(fn, **kwargs, meta=...) -- applied eagerly to a list of pandas partitions here."""

    def __init__(self, partitions):
        self.partitions = partitions

    def __getitem__(self, col):
        return self.partitions[0][col]

    def map_partitions(self, fn, meta=None, **kwargs):
        out = pd.concat([fn(p, **kwargs) for p in self.partitions], ignore_index=True)
        assert list(out.columns) == list(meta.columns) and all(out.dtypes == meta.dtypes)
        return out


@pytest.mark.gpu
def test_dask_binding_wiring(gpu, monkeypatch):
    """bindings.py:9-60 at partition grain.  dask is not in this image: the wrapper's wiring (meta frame, keyword
    plumbing, one extraction per partition) runs against a stand-in frame; with dask installed the second half builds and
    computes a real graph."""
    import sys
    import types
    from tsfresh_amd import MinimalFCParameters
    from tsfresh_amd.convenience.bindings import dask_feature_extraction_on_chunk, feature_extraction_on_partition
    chunks = _chunks()
    df = pd.concat([pd.DataFrame({"id": c.id, "kind": c.kind, "t": np.arange(len(c.data)), "v": c.data.to_numpy()})
                    for c in chunks], ignore_index=True)
    want = feature_extraction_on_partition(df, "id", "kind", "v", "t", MinimalFCParameters())
    ids = sorted(df["id"].unique())
    parts = [df[df["id"].isin(ids[:1])], df[df["id"].isin(ids[1:])]]
    try:
        import dask.dataframe as dd
        real = True
    except ImportError:
        real = False
        fake = types.ModuleType("dask")
        fake.dataframe = types.ModuleType("dask.dataframe")
        monkeypatch.setitem(sys.modules, "dask", fake)
        monkeypatch.setitem(sys.modules, "dask.dataframe", fake.dataframe)
    got = dask_feature_extraction_on_chunk(_StandInDaskFrame(parts), "id", "kind", "v", "t", MinimalFCParameters())
    key = ["id", "variable"]
    assert got.sort_values(key).reset_index(drop=True).equals(want.sort_values(key).reset_index(drop=True))
    if real:
        ddf = dd.from_pandas(df.sort_values("id"), npartitions=2)
        out = dask_feature_extraction_on_chunk(ddf, "id", "kind", "v", "t", MinimalFCParameters()).compute()
        assert out.sort_values(key).reset_index(drop=True).equals(want.sort_values(key).reset_index(drop=True))
