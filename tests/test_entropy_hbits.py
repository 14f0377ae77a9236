"""sample_entropy / approximate_entropy beyond 4096 samples by the bit-matrix sweep of fam_entropy_hbits.h (round 6: table of one
column part in LDS, per-sample arrays in HBM scratch, tasks in register batches; 4097 .. 17 408 samples) against the float64
pair sweep it replaces (plan option "entropy_route" 1) and against the oracle.  The counts are integers: the columns agree to
the rounding of the closing logarithms."""
import numpy as np
import pytest

from engines import hip_engine, oracle_engine
from parity import compare

pytestmark = pytest.mark.gpu
PARAMS = {"sample_entropy": None, "approximate_entropy": [{"m": 2, "r": r} for r in (0.1, 0.3, 0.5, 0.7, 0.9)]}


def _series(rng, n, kind, dtype):
    if kind == 0:
        x = rng.standard_normal(n)
    elif kind == 1:
        x = np.cumsum(rng.standard_normal(n))
    elif kind == 2:
        x = np.round(rng.standard_normal(n), 1)       # ties
    elif kind == 3:
        x = np.sin(np.arange(n) * 0.01) + 0.1 * rng.standard_normal(n)
    else:
        x = np.full(n, 2.5)
        x[: n // 7] = rng.standard_normal(n // 7)      # a constant stretch: ranges that cover thousands of equal samples
    return x.astype(dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_bit_matrix_sweep_beyond_4096_samples_equals_the_pair_sweep(gpu, dtype):
    rng = np.random.default_rng(31)
    lens = [4097, 4100, 5000, 6001, 8192, 8193, 12000, 13441, 16384, 17408, 300, 3, 31, 32, 33, 64, 1024, 2]
    chunks = [_series(rng, n, i % 5, dtype) for i, n in enumerate(lens)]
    values = np.concatenate(chunks)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    names, bits = hip_engine(PARAMS, values, offsets)
    names2, pairs = hip_engine(PARAMS, values, offsets, options={"entropy_route": 1})
    assert names == names2
    assert np.array_equal(np.isnan(bits), np.isnan(pairs)) and np.array_equal(np.isinf(bits), np.isinf(pairs))
    ok = np.isfinite(bits)
    assert np.allclose(bits[ok], pairs[ok], rtol=1e-11, atol=1e-12), np.abs(bits[ok] - pairs[ok]).max()
    # the oracle on the two shortest long series (its approximate_entropy holds an n x n float64 array per tolerance)
    sub = [0, 1]
    sv = np.concatenate([chunks[i] for i in sub]).astype(np.float64)
    so = np.concatenate([[0], np.cumsum([lens[i] for i in sub])]).astype(np.int64)
    onames, want = oracle_engine(PARAMS, sv, so)
    assert onames == names
    bad = compare(names, bits[sub], want, [chunks[i].astype(np.float64) for i in sub])
    assert not bad, bad


def test_comprehensive_on_series_of_16384_samples(gpu):
    """The whole settings object on the shape round 4 measured at 0.45 ms per series: every column written, the entropy columns
    equal to the pair sweep's."""
    from tsfresh_amd.feature_extraction import settings
    rng = np.random.default_rng(4)
    x = rng.standard_normal((3, 16384)).astype(np.float32)
    values, offsets = x.reshape(-1), np.arange(4, dtype=np.int64) * 16384
    params = settings.ComprehensiveFCParameters()
    names, got = hip_engine(params, values, offsets)
    _, ref = hip_engine(params, values, offsets, options={"entropy_route": 1})
    ent = [j for j, nm in enumerate(names) if nm.split("__")[1] in ("sample_entropy", "approximate_entropy")]
    assert len(ent) == 6 and np.all(np.isfinite(got[:, ent]))
    assert np.allclose(got[:, ent], ref[:, ent], rtol=1e-11, atol=1e-12)
    rest = [j for j in range(len(names)) if j not in ent]
    assert np.array_equal(got[:, rest], ref[:, rest], equal_nan=True)
