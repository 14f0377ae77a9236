"""permutation_entropy (fc.py:1866-1916): all dimensions of one stride from ONE sweep of the windows (fam_perm.h, k_perm:
prefix inversion-table codes, every histogram side by side in one LDS table) against the one-dimension-at-a-time path
in k_sort (fam_sort.h) and the oracle.  The emulation takes k_perm's code on the even series of a batch and k_sort's on
the odd ones, so every input appears twice in a row."""
import numpy as np
import pytest

from engines import emul_engine, oracle_engine
from parity import compare

ALL5 = {"permutation_entropy": [{"tau": 1, "dimension": d} for d in (3, 4, 5, 6, 7)]}        # ComprehensiveFCParameters
SETS = {
    "comprehensive": ALL5,                                                                                   # fused
    "comprehensive_stride3": {"permutation_entropy": [{"tau": 3, "dimension": d} for d in (7, 3, 5, 4, 6)]},  # fused, any order
    "subset": {"permutation_entropy": [{"tau": 2, "dimension": d} for d in (7, 3, 5)]},                       # other sets: not fused
    "low": {"permutation_entropy": [{"tau": 3, "dimension": d} for d in (2, 3, 4)]},
    "two_dims": {"permutation_entropy": [{"tau": 1, "dimension": 6}, {"tau": 1, "dimension": 2}]},
    "mixed_strides": {"permutation_entropy": [{"tau": 1, "dimension": 3}, {"tau": 2, "dimension": 4}, {"tau": 1, "dimension": 5}]},  # not fused
    "single": {"permutation_entropy": [{"tau": 1, "dimension": 7}]},                                                                  # not fused
}


def pe_series():
    rng = np.random.default_rng(91)
    out = [rng.standard_normal(n) for n in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 20, 50, 129, 300, 1000, 1024, 1500, 2049)]
    out.append(np.round(rng.standard_normal(400), 1))                    # ties inside the windows: stable ranks
    out.append(rng.integers(0, 3, size=700).astype(np.float64))          # three distinct values
    out.append(np.full(90, 2.5))                                         # one pattern
    out.append(np.arange(300, dtype=np.float64))                         # one pattern
    out.append(np.tile([1.0, 3.0, 2.0], 200))                            # three patterns
    out.append(np.cumsum(rng.standard_normal(1024)))
    out.append(rng.standard_normal(1024).astype(np.float32).astype(np.float64))
    return out


@pytest.mark.parametrize("name", sorted(SETS))
def test_fused_dimensions_equal_the_oracle_emulated(name):
    params = SETS[name]
    twice = [s for s in pe_series() for _ in (0, 1)]
    values = np.concatenate(twice)
    offsets = np.concatenate([[0], np.cumsum([len(s) for s in twice])]).astype(np.int64)
    names, got = emul_engine(params, values, offsets)
    onames, want = oracle_engine(params, values, offsets)
    assert list(names) == list(onames)
    bad = compare(names, got, want, twice)          # (stable ranks: the oracle's, parity.py R1 only applies to SIMD fixtures)
    assert not bad, bad[:6]
    # fused sweep vs one dimension at a time: the same counts, logarithms taken as log c - log num instead of log(c / num)
    np.testing.assert_allclose(got[0::2], got[1::2], rtol=1e-13, atol=1e-14, equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_fused_dimensions_equal_the_oracle_on_the_device(gpu, dtype):
    from engines import hip_engine
    rng = np.random.default_rng(92)
    # (beyond 8 x 1024 windows the codes are recomputed per pass instead of held in registers)
    series = [s.astype(dtype).astype(np.float64) for s in pe_series() + [rng.standard_normal(3000), rng.standard_normal(9000)]]
    values = np.concatenate(series).astype(dtype)
    offsets = np.concatenate([[0], np.cumsum([len(s) for s in series])]).astype(np.int64)
    for name in sorted(SETS):
        params = SETS[name]
        names, got = hip_engine(params, values, offsets)
        onames, want = oracle_engine(params, values.astype(np.float64), offsets)
        assert list(names) == list(onames)
        bad = compare(names, got, want, series)
        assert not bad, (name, bad[:6])


@pytest.mark.gpu
def test_the_kernel_of_its_own_and_the_columns_in_k_sort_agree(gpu, monkeypatch):
    """TSFA_NO_PE_FUSED (read when the plan is built) leaves the columns to k_sort, one dimension at a time."""
    from engines import hip_engine
    series = [s.astype(np.float32) for s in pe_series()]
    values = np.concatenate(series)
    offsets = np.concatenate([[0], np.cumsum([len(s) for s in series])]).astype(np.int64)
    _, own = hip_engine(ALL5, values, offsets)
    monkeypatch.setenv("TSFA_NO_PE_FUSED", "1")
    _, in_sort = hip_engine(ALL5, values, offsets)
    np.testing.assert_allclose(own, in_sort, rtol=1e-13, atol=1e-14, equal_nan=True)
