"""query_similarity_count WITH a query (fc.py:2475-2519; VERDICT r5 missing #5).

The distances are stumpy's (`stumpy.core.mass` / `mass_absolute`), a dependency of the reference that is neither vendored
under /root/reference nor installed in this image (setup.cfg:47, stumpy >= 1.11.1): oracle/third_party.py restates its
published definition, and the anchor is the reference's own unit test (tests/units/feature_extraction/
test_feature_calculations.py:2017-2037) -- its inputs are reproduced here from the same seed, its four expected counts are
the vectors.  Counts are integers: compared exactly.
"""
import numpy as np
import pytest

from engines import emul_engine, hip_engine, oracle_engine


def _reference_unit_test_case():
    np.random.seed(42)   # test_feature_calculations.py:2018-2021
    query = np.random.uniform(size=10)
    x = np.random.uniform(size=100)
    params = [{"query": query}, {"query": query, "threshold": 3.0}, {"query": query, "normalize": False},
              {"query": query, "threshold": 3.0, "normalize": False}]
    return x, params, [0.0, 6.0, 0.0, 91.0]   # :2026, :2029, :2034, :2037


def _series_set(rng):
    """planted exact and scaled copies of the query, constant stretches, a series of exactly the query's length"""
    q = np.array([0.0, 1.0, 3.0, 2.0, -1.0, 0.5])
    a = rng.normal(size=200)
    a[20:26] = q                      # exact copy: distance 0 both ways
    a[90:96] = 2.5 * q + 7.0          # affine copy: z-normalised distance 0, Euclidean distance large
    b = rng.normal(size=64)
    b[10:30] = 4.0                    # constant windows: sqrt(m) from a non-constant query
    c = q + 1e-3 * rng.normal(size=6) # n == m: one window
    d = np.cumsum(rng.normal(size=333))
    e = np.full(40, -2.0)             # a constant series
    series = [a, b, c, d, e]
    params = [{"query": q}, {"query": q, "threshold": 0.5}, {"query": q, "threshold": 2.0}, {"query": q, "normalize": False},
              {"query": q, "threshold": 1.5, "normalize": False}, {"query": q, "threshold": 40.0, "normalize": False},
              {"query": np.full(6, 3.0), "threshold": 0.0},      # a constant query: matches the constant windows only
              {"query": [1.0, 2.0], "threshold": 5.0},           # fewer than three samples: NaN (fc.py:2511)
              {"query": None, "threshold": 0.0}]                 # the settings objects' own column: NaN
    return series, params


def _pack(series):
    values = np.concatenate(series)
    offsets = np.concatenate([[0], np.cumsum([len(s) for s in series])]).astype(np.int64)
    return values, offsets


def test_oracle_gives_the_references_known_answers():
    x, params, want = _reference_unit_test_case()
    _, got = oracle_engine({"query_similarity_count": params}, x, np.array([0, len(x)]))
    assert list(got[0]) == want


def test_kernel_sources_give_the_references_known_answers_and_the_oracles_counts():
    x, params, want = _reference_unit_test_case()
    _, got = emul_engine({"query_similarity_count": params}, x, np.array([0, len(x)]))
    assert list(got[0]) == want
    series, params = _series_set(np.random.default_rng(5))
    values, offsets = _pack(series)
    names_o, o = oracle_engine({"query_similarity_count": params}, values, offsets)
    names_e, e = emul_engine({"query_similarity_count": params}, values, offsets)
    assert list(names_o) == list(names_e)
    np.testing.assert_array_equal(o, e)
    # the planted copies are found: exact copy at threshold 0 both ways, the affine copy only z-normalised
    assert o[0, 0] == 2.0 and o[0, 3] == 1.0
    assert np.isnan(o[:, 7]).all() and np.isnan(o[:, 8]).all()
    assert o[4, 6] == 35.0   # constant query on a constant series: every window


def test_a_series_shorter_than_the_query_raises_like_stumpy():
    from tsfresh_amd.feature_extraction.reference_errors import check_query_lengths
    specs = [("query_similarity_count", (0.0, 1.0, 0.0, 5.0, 1.0, 2.0, 3.0, 4.0, 5.0)), ("maximum", (0.0, 0.0, 0.0, 0.0))]
    check_query_lengths(specs, np.array([0, 7]), np.array([7, 12]))   # lengths 7, 5: fine
    with pytest.raises(ValueError, match="window size must be less than or equal to 4"):
        check_query_lengths(specs, np.array([0, 7]), np.array([7, 11]))


@pytest.mark.gpu
def test_hip_query_similarity_count():
    x, params, want = _reference_unit_test_case()
    _, got = hip_engine({"query_similarity_count": params}, x.astype(np.float64), np.array([0, len(x)]))
    assert list(got[0]) == want
    series, params = _series_set(np.random.default_rng(5))
    values, offsets = _pack(series)
    _, o = oracle_engine({"query_similarity_count": params}, values, offsets)
    _, h = hip_engine({"query_similarity_count": params}, values, offsets)
    np.testing.assert_array_equal(o, h)
    # beside the tuned families in one plan, float32 samples, many series (several workgroups per slot of k_general)
    rng = np.random.default_rng(11)
    n, L = 3000, 128
    v32 = rng.normal(size=n * L).astype(np.float32)
    off = np.arange(n + 1, dtype=np.int64) * L
    q = v32[5 * L + 17: 5 * L + 29].astype(np.float64)   # a window of series 5
    fc = {"maximum": None, "query_similarity_count": [{"query": q, "threshold": 0.0}, {"query": q, "threshold": 3.5},
                                                      {"query": q, "threshold": 4.0, "normalize": False}], "median": None}
    names_o, o = oracle_engine(fc, v32.astype(np.float64), off)
    names_h, h = hip_engine(fc, v32, off)
    assert list(names_o) == list(names_h)
    np.testing.assert_array_equal(o[:, 1:4], h[:, 1:4])
    assert h[5, 1] >= 1.0
    np.testing.assert_allclose(h[:, [0, 4]], o[:, [0, 4]], rtol=1e-12)


@pytest.mark.gpu
def test_extract_features_with_a_query():
    import pandas as pd
    from tsfresh_amd import extract_features
    rng = np.random.default_rng(3)
    df = pd.DataFrame({"id": np.repeat(np.arange(6), 50), "t": np.tile(np.arange(50), 6), "v": rng.normal(size=300)})
    q = df["v"].to_numpy()[60:70].copy()
    fc = {"query_similarity_count": [{"query": q, "threshold": 0.0}, {"query": q, "threshold": 3.0}]}
    X = extract_features(df, column_id="id", column_sort="t", default_fc_parameters=fc, disable_progressbar=True)
    assert X.shape == (6, 2)
    assert X.iloc[1, 0] == 1.0 and (X.iloc[:, 1] >= X.iloc[:, 0]).all()
    # the column names are the reference's convert_to_output_format of the parameter dict (numpy's str of the array)
    assert X.columns[0].startswith("v__query_similarity_count__query_[")
    with pytest.raises(ValueError, match="window size must be less than or equal to"):
        extract_features(df[df["t"] < 8], column_id="id", column_sort="t", default_fc_parameters=fc, disable_progressbar=True)


@pytest.mark.gpu
def test_a_query_outside_the_data_pool_is_refused():
    """tsfa_plan_create_with_data (include/tsfresh_amd.h): p[2] / p[3] must lie inside the pool; tsfa_plan_create is the
    case of an empty pool, so a spec that names a query there is refused as well."""
    import ctypes
    from tsfresh_amd import _native
    lib = _native.load()
    spec = (_native.FeatureSpec * 1)()
    spec[0].calc = _native.calc_id("query_similarity_count")
    for k, v in enumerate((0.0, 1.0, 2.0, 5.0)):   # offset 2, length 5
        spec[0].p[k] = v
    pool = np.arange(6, dtype=np.float64)           # 2 + 5 > 6
    h = ctypes.c_void_p()
    rc = lib.tsfa_plan_create_with_data(spec, 1, pool.ctypes.data, len(pool), 0, ctypes.byref(h))
    assert rc == _native.TSFA_ERR_INVALID and b"outside the data pool" in lib.tsfa_last_error()
    rc = lib.tsfa_plan_create(spec, 1, 0, ctypes.byref(h))
    assert rc == _native.TSFA_ERR_INVALID
    pool = np.arange(7, dtype=np.float64)
    rc = lib.tsfa_plan_create_with_data(spec, 1, pool.ctypes.data, len(pool), 0, ctypes.byref(h))
    assert rc == 0
    lib.tsfa_plan_destroy(h)
