"""Pin the oracle: it must reproduce what the REAL reference returned (tests/golden/ref_*.npz, generated in the
build container by tests/golden/gen_golden_main.py / gen_golden_conda.py)."""
import os

import numpy as np
import pytest

from engines import emul_engine, oracle_engine
from parity import compare
from tsfresh_amd.feature_extraction.settings import ComprehensiveFCParameters

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _golden():
    g1 = np.load(os.path.join(G, "ref_main.npz"))
    g2 = np.load(os.path.join(G, "ref_conda.npz"))
    assert np.array_equal(g1["values"], g2["values"]) and np.array_equal(g1["offsets"], g2["offsets"])
    names = list(g1["names"]) + list(g2["names"])
    matrix = np.concatenate([g1["matrix"], g2["matrix"]], axis=1)
    values, offsets = g1["values"], g1["offsets"]
    series = [values[offsets[i]:offsets[i + 1]] for i in range(len(offsets) - 1)]
    return names, matrix, values, offsets, series


def _align(names_want, names_got, got):
    assert set(names_want) == set(names_got), (set(names_want) ^ set(names_got))
    idx = [names_got.index(n) for n in names_want]
    return got[:, idx]


def test_golden_has_all_783_columns():
    names, matrix, *_ = _golden()
    assert len(names) == 783 and matrix.shape[1] == 783  # 788 minus 5 linear_trend_timewise


@pytest.mark.parametrize("engine", [oracle_engine, emul_engine], ids=["oracle", "emul"])
def test_engine_matches_reference_golden(engine):
    names, want, values, offsets, series = _golden()
    got_names, got = engine(ComprehensiveFCParameters(), values, offsets)
    got = _align(names, got_names, got)
    bad = compare(names, got, want, series)
    assert not bad, "%d mismatches, first: %s" % (len(bad), bad[:10])


def test_emul_column_order_is_the_reference_order():
    g1 = np.load(os.path.join(G, "ref_main.npz"))
    ref_order = list(g1["names"])
    values, offsets = g1["values"][:60], np.array([0, 60])
    names, _ = emul_engine(ComprehensiveFCParameters(), values, offsets)
    ours = [n for n in names if n in set(ref_order)]
    assert ours == ref_order


def test_emul_number_cwt_peaks_long_windows_match_oracle():
    """Series long enough (n > ~1400) that the SNR filter's noise percentile takes the argsort-scan path."""
    rng = np.random.default_rng(123)
    lens = [1500, 3000, 2049]
    chunks = [rng.standard_normal(lens[0]), np.cumsum(rng.standard_normal(lens[1])), rng.standard_normal(lens[2])]
    values = np.concatenate(chunks)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    params = {"number_cwt_peaks": [{"n": 1}, {"n": 5}]}
    names, got = emul_engine(params, values, offsets)
    onames, want = oracle_engine(params, values, offsets)
    assert names == onames
    assert np.array_equal(got, want), (got, want)
