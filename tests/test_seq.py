"""lempel_ziv_complexity (fc.py:1825-1862) against the oracle on the inputs that stress the parse: every length class,
long runs (deep phrases: the hashed part of the trie), constant / periodic / tie-heavy series, samples ON the bin edges,
spreads of a few ulps, alphabets up to 255 symbols and more chains than one launch group holds.  The count is an
integer: every cell is compared exactly.  (Round 4 also measured a wavefront-per-series form of the kernel with the
symbols in registers -- slower, profiles/r04_d_seq_wave_ab.txt; the inputs below were its parity set.)"""
import numpy as np
import pytest

from engines import emul_engine, oracle_engine

BINS = {"lempel_ziv_complexity": [{"bins": b} for b in (2, 3, 5, 10, 100)]}                      # ComprehensiveFCParameters
WIDE = {"lempel_ziv_complexity": [{"bins": b} for b in (100, 128, 129, 255, 64, 2, 200, 17, 1, 31, 250)]}   # > 32 packed bits: several groups


def lz_series():
    rng = np.random.default_rng(77)
    out = []
    for n in (1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 63, 64, 65, 100, 255, 256, 257, 511, 512, 513, 1000, 1023, 1024, 1025, 1500, 2047, 2048):
        out.append(rng.standard_normal(n))
    out.append(np.cumsum(rng.standard_normal(1024)))                      # long runs in the coarse alphabets: deep phrases
    out.append(np.cumsum(rng.standard_normal(300)) * 0.01)
    out.append(np.full(700, 2.5))                                         # constant: every edge equals the minimum
    out.append(np.zeros(64))
    out.append(np.arange(512, dtype=np.float64))                          # ramp: samples ON the edges
    out.append(np.tile([0.0, 1.0, 2.0, 3.0], 200))                        # periodic
    out.append(rng.integers(0, 5, size=900).astype(np.float64))           # few distinct values, ties with edges
    out.append(rng.integers(0, 100, size=1024).astype(np.float64))
    out.append(1e9 + np.arange(400) % 3)                                  # spread of a few ulps: the last edge may break the order
    out.append(1e16 + 2.0 * (np.arange(300) % 5))
    out.append(np.float32(1e7) + rng.integers(0, 3, size=500).astype(np.float64))
    out.append(rng.standard_normal(1024).astype(np.float32).astype(np.float64))
    out.append(np.concatenate([np.full(500, -1.0), np.full(524, 1.0)]))   # two runs: the deepest phrases an alphabet can get
    return out


def _pack_twice(series):
    twice = [s for s in series for _ in (0, 1)]
    return twice, np.concatenate(twice), np.concatenate([[0], np.cumsum([len(s) for s in twice])]).astype(np.int64)


@pytest.mark.parametrize("params", [BINS, WIDE], ids=["comprehensive_bins", "wide_groups"])
def test_lempel_ziv_forms_equal_the_oracle_emulated(params):
    series, values, offsets = _pack_twice(lz_series())
    names, got = emul_engine(params, values, offsets)
    onames, want = oracle_engine(params, values, offsets)
    assert list(names) == list(onames)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(got[0::2], got[1::2])                   # (the emulation alternates its launch-group shapes)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("params", [BINS, WIDE], ids=["comprehensive_bins", "wide_groups"])
def test_lempel_ziv_forms_equal_the_oracle_on_the_device(gpu, params, dtype, monkeypatch):
    from engines import hip_engine
    series = [s.astype(dtype).astype(np.float64) for s in lz_series()]
    values = np.concatenate(series).astype(dtype)
    offsets = np.concatenate([[0], np.cumsum([len(s) for s in series])]).astype(np.int64)
    onames, want = oracle_engine(params, values.astype(np.float64), offsets)
    names, got = hip_engine(params, values, offsets)                      # ragged batch: one launch per length class
    assert list(names) == list(onames)
    np.testing.assert_array_equal(got, want)
    # every third series alone (its own launch shape)
    for s in series[::3]:
        _, one = hip_engine(params, s.astype(dtype), np.array([0, len(s)], dtype=np.int64))
        _, ow = oracle_engine(params, s, np.array([0, len(s)], dtype=np.int64))
        np.testing.assert_array_equal(one, ow)


@pytest.mark.gpu
@pytest.mark.parametrize("params", [BINS, WIDE], ids=["comprehensive_bins", "wide_groups"])
def test_symbol_rows_in_hbm_and_in_lds_give_the_same_counts(gpu, params):
    """Round 6: k_seq<T, true> keeps the symbol rows in HBM (a byte per symbol, read three 16-byte blocks ahead of the chain)
    so that LDS holds the tables alone; the library takes that form where it puts more series on a CU (long series).  Both
    forms on every stress input, and on 4096 .. 8192-sample series where the HBM form is the default."""
    from engines import hip_engine
    rng = np.random.default_rng(5)
    series = lz_series() + [rng.standard_normal(n) for n in (4096, 5000, 8191, 8192)] + [np.cumsum(rng.standard_normal(6000))]
    values = np.concatenate(series)
    offsets = np.concatenate([[0], np.cumsum([len(s) for s in series])]).astype(np.int64)
    _, want = oracle_engine(params, values, offsets)
    for rows in (0, 1, -1):
        _, got = hip_engine(params, values, offsets, options={"seq_rows": rows})
        np.testing.assert_array_equal(got, want, err_msg="seq_rows=%d" % rows)

