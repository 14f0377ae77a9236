"""permutation_entropy (fc.py:1866-1916): all dimensions of one stride from ONE sweep of the windows (fam_perm.h, k_perm:
prefix inversion-table codes, every histogram side by side in one LDS table) against the one-dimension-at-a-time path
in k_sort (fam_sort.h) and the oracle.  The emulation takes k_perm's code on the even series of a batch and k_sort's on
the odd ones, so every input appears twice in a row."""
import json
import os
import re
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from engines import emul_engine, oracle_engine
from parity import compare
from perm_cases import ALL5, SETS, pe_series

REF = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_perm.json")))


def reference_matrix(name, names):
    """The real reference's values (tests/golden/gen_golden_perm.py) in the engine's column order."""
    cols = {(c["dimension"], c["tau"]): c["values"] for c in REF["sets"][name]}
    out = []
    for nm in names:
        m = re.search(r"permutation_entropy__dimension_(\d+)__tau_(\d+)$", nm)
        out.append([np.nan if v is None else v for v in cols[(int(m.group(1)), int(m.group(2)))]])
    return np.array(out, dtype=np.float64).T


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
    ref = reference_matrix(name, names)             # the real reference under numpy's scalar (stable) sort
    bad = compare(names, want[0::2], ref, pe_series()) + compare(names, got[0::2], ref, pe_series()) + compare(names, got[1::2], ref, pe_series())
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
        if dtype == np.float64:   # (the fixture's series are float64; the last two of this batch are not in it)
            k = REF["n_series"]
            bad = compare(names, got[:k], reference_matrix(name, names), series[:k])
            assert not bad, (name, bad[:6])


@pytest.mark.gpu
def test_the_kernel_of_its_own_and_the_columns_in_k_sort_agree(gpu, monkeypatch):
    """Option "perm_fused" 0 leaves the columns to k_sort, one dimension at a time."""
    from engines import hip_engine
    series = [s.astype(np.float32) for s in pe_series()]
    values = np.concatenate(series)
    offsets = np.concatenate([[0], np.cumsum([len(s) for s in series])]).astype(np.int64)
    _, own = hip_engine(ALL5, values, offsets)
    _, in_sort = hip_engine(ALL5, values, offsets, options={"perm_fused": 0})
    np.testing.assert_allclose(own, in_sort, rtol=1e-13, atol=1e-14, equal_nan=True)
