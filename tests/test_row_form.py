"""Row form of the BASIC / TREND families (round 6, tsfa_common.h BlkRow): series of at most 256 samples are evaluated four to a
wavefront, one per 16-lane DPP row (k_basic_rows / k_trend_rows).  Checked against the wavefront form (plan option "row_form"
0), against the oracle, and for the property the form was designed around: WHICH form evaluates a series depends on its
length alone, so a series gives the same bits whatever batch it travels in."""
import numpy as np
import pytest

from engines import hip_engine, oracle_engine
from parity import compare, is_integer_feature
from tsfresh_amd.feature_extraction import settings

pytestmark = pytest.mark.gpu


def _series(rng, n, dtype, kind):
    if kind == 0:
        x = rng.standard_normal(n)
    elif kind == 1:
        x = np.cumsum(rng.standard_normal(n))
    elif kind == 2:
        x = np.round(rng.standard_normal(n), 1)          # ties
    elif kind == 3:
        x = np.full(n, 0.1)                               # constant: the numpy-order mean / variance decide the counts
    elif kind == 4:
        x = rng.integers(-3, 4, n).astype(float)
    else:
        x = np.sin(np.arange(n) * 0.3) + 5.0
    return x.astype(dtype)


def _batch(dtype, lens, seed):
    rng = np.random.default_rng(seed)
    chunks = [_series(rng, int(n), dtype, i % 6) for i, n in enumerate(lens)]
    values = np.concatenate(chunks)
    offsets = np.concatenate([[0], np.cumsum([len(c) for c in chunks])]).astype(np.int64)
    return chunks, values, offsets


LENS = list(range(1, 40)) + [63, 64, 65, 100, 127, 128, 129, 135, 136, 137, 200, 248, 249, 250, 251, 252, 253, 254, 255, 256]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_row_form_equals_the_wavefront_form_and_the_oracle(gpu, dtype):
    chunks, values, offsets = _batch(dtype, LENS + [257, 300, 1024, 5, 256, 77], 11)
    params = settings.ComprehensiveFCParameters()
    names, rows = hip_engine(params, values, offsets)
    names2, waves = hip_engine(params, values, offsets, options={"row_form": 0})
    assert names == names2
    series = [c.astype(np.float64) for c in chunks]
    assert np.array_equal(np.isnan(rows), np.isnan(waves))
    ints = [j for j, n in enumerate(names) if is_integer_feature(n)]
    assert np.array_equal(rows[:, ints], waves[:, ints], equal_nan=True)
    bad = compare(names, rows, waves, series, check_excluded=True)     # two summation orders of the same kernels' arithmetic
    assert not bad, bad[:10]
    onames, want = oracle_engine(params, values.astype(np.float64), offsets)
    assert onames == names
    bad = compare(names, rows, want, series)
    assert not bad, bad[:10]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_a_series_gives_the_same_bits_in_every_batch(gpu, dtype):
    """Alone, among short series, among long ones (one launch group with a longer LDS carve, the family kernel skipping the
    short series), in a large batch that is launched class by class: the same row of numbers."""
    chunks, values, offsets = _batch(dtype, [256, 200, 17, 129, 64, 255, 3, 1], 5)
    params = settings.EfficientFCParameters()
    names, base = hip_engine(params, values, offsets)
    for i, c in enumerate(chunks):
        _, alone = hip_engine(params, c, np.array([0, len(c)], dtype=np.int64))
        assert np.array_equal(alone[0], base[i], equal_nan=True), (i, len(c))
    rng = np.random.default_rng(8)
    long_ones = [rng.standard_normal(n).astype(dtype) for n in (1000, 4096, 300, 2000)]
    mixed = long_ones[:2] + chunks + long_ones[2:]
    mv = np.concatenate(mixed)
    mo = np.concatenate([[0], np.cumsum([len(c) for c in mixed])]).astype(np.int64)
    # (the columns of the two families: the others choose their workgroup size from the longest series of the launch group,
    #  and their float sums follow it in the last bit -- as before this round)
    small = {"mean": None, "variance": None, "skewness": None, "kurtosis": None, "autocorrelation": [{"lag": 2}, {"lag": 9}], "c3": [{"lag": 1}],
             "time_reversal_asymmetry_statistic": [{"lag": 2}], "linear_trend": [{"attr": a} for a in ("slope", "stderr", "pvalue")],
             "agg_linear_trend": [{"attr": "slope", "chunk_len": 5, "f_agg": "var"}, {"attr": "stderr", "chunk_len": 10, "f_agg": "max"}],
             "index_mass_quantile": [{"q": 0.4}], "number_peaks": [{"n": 3}, {"n": 10}], "binned_entropy": [{"max_bins": 10}],
             "benford_correlation": None, "longest_strike_above_mean": None, "ratio_beyond_r_sigma": [{"r": 1.0}],
             "number_crossing_m": [{"m": 0}], "cid_ce": [{"normalize": True}], "mean_abs_change": None,
             "energy_ratio_by_chunks": [{"num_segments": 10, "segment_focus": 3}], "large_standard_deviation": [{"r": 0.25}],
             "first_location_of_maximum": None, "has_duplicate_min": None, "abs_energy": None}
    _, base_small = hip_engine(small, values, offsets)
    _, got = hip_engine(small, mv, mo)
    assert np.array_equal(got[2:2 + len(chunks)], base_small, equal_nan=True)
    # > 2048 series with lengths over more than a factor of two: length-class launch groups through index lists
    many = [rng.standard_normal(int(n)).astype(dtype) for n in rng.integers(8, 700, size=2300)]
    big = many[:1000] + chunks + many[1000:]
    bv = np.concatenate(big)
    bo = np.concatenate([[0], np.cumsum([len(c) for c in big])]).astype(np.int64)
    _, got_big = hip_engine(small, bv, bo)
    assert np.array_equal(got_big[1000:1000 + len(chunks)], base_small, equal_nan=True)


def test_row_form_with_a_datetime_index_and_windows(gpu):
    """linear_trend_timewise reads the per-sample times of ITS row; rolled windows are (start, end) views."""
    from tsfresh_amd import _native
    from tsfresh_amd.feature_extraction.plan import compile_fc_parameters
    rng = np.random.default_rng(2)
    lens = [50, 256, 7, 130]
    chunks = [rng.standard_normal(n) for n in lens]
    times = [np.cumsum(rng.uniform(0.5, 2.0, n)) for n in lens]
    times = [t - t[0] for t in times]
    values, tt = np.concatenate(chunks), np.concatenate(times)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    params = {"linear_trend_timewise": [{"attr": a} for a in ("pvalue", "rvalue", "intercept", "slope", "stderr")],
              "linear_trend": [{"attr": "slope"}], "mean": None}
    names, rows = hip_engine(params, values, offsets, times=tt)
    _, waves = hip_engine(params, values, offsets, times=tt, options={"row_form": 0})
    assert np.allclose(rows, waves, rtol=1e-11, atol=1e-13, equal_nan=True)
    onames, want = oracle_engine(params, values, offsets, times=tt)
    assert not compare(onames, rows[:, [names.index(n) for n in onames]], want, chunks)
    fplan = compile_fc_parameters({"mean": None, "autocorrelation": [{"lag": 3}], "number_peaks": [{"n": 1}]})
    plan = _native.Plan(fplan.native_specs(_native.calc_id), device=0)
    x = rng.standard_normal(400)
    starts = np.array([0, 10, 100, 399, 144], dtype=np.int64)
    ends = np.array([256, 20, 356, 400, 400], dtype=np.int64)
    got = plan.extract_windows_host(x, starts, ends)
    for r, (s0, e0) in enumerate(zip(starts, ends)):
        alone = plan.extract_host(x[s0:e0].copy(), np.array([0, e0 - s0], dtype=np.int64))
        assert np.array_equal(got[r], alone[0], equal_nan=True)
    plan.close()
