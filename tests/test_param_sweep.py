"""Parameters away from the grids of ComprehensiveFCParameters (tests/golden/param_cases.py: 229 columns of 37
calculators -- what a from_columns / hand-written settings dict sends down the same kernels) against outputs of the REAL
reference (ref_main_sweep.npz / ref_conda_sweep.npz, `gen_golden_main.py --params sweep`, `gen_golden_conda.py --params
sweep`)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import goldens
from engines import emul_engine, oracle_engine
from param_cases import sweep_parameters


def test_fixture_has_every_column_of_the_sweep():
    import warnings
    from tsfresh_amd.feature_extraction.plan import compile_fc_parameters
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fplan = compile_fc_parameters(sweep_parameters())
    g = goldens.load("sweep")
    assert sorted(g["names"]) == sorted("value__" + n for n in fplan.names)
    assert len(g["names"]) == 233    # (229 until round 6: + binned_entropy 257 / 1000 bins, fourier_entropy 129 / 300 bins)


# skipped cells (tests/parity.py R1-R11) per set: the sweep is MADE of the calculators those exclusions are about, and the
# degenerate set of the series they are about
SETS = {"sweep": 0.015, "degenerate_sweep": 0.045, "offset_sweep": 0.005}
SETS_HIP = dict(SETS, long_sweep=0.01)


@pytest.mark.parametrize("pair", sorted(SETS))
@pytest.mark.parametrize("engine", [oracle_engine, emul_engine], ids=["oracle", "emul"])
def test_engine_matches_the_reference_on_other_parameters(engine, pair):
    bad, skipped, cells = goldens.check_engine(engine, pair, sweep_parameters())
    assert not bad, "%d mismatches, first: %s" % (len(bad), bad[:10])
    assert len(skipped) <= SETS[pair] * cells, (len(skipped), cells)


def test_emulation_matches_the_reference_on_other_parameters_beyond_1024_samples():
    """The 1025 .. 8192-sample series (lags of 500, peak supports of 60, Welch / FFT coefficients that only exist there).
    The kernels' sources only: the oracle's approximate_entropy for m = 1 and 3 holds several n x n float64 matrices per
    series and does not finish this set in an hour on the 8 build cores."""
    bad, skipped, cells = goldens.check_engine(emul_engine, "long_sweep", sweep_parameters())
    assert not bad, "%d mismatches, first: %s" % (len(bad), bad[:10])
    assert len(skipped) <= 0.01 * cells, (len(skipped), cells)


@pytest.mark.gpu
def test_hip_matches_the_reference_on_other_parameters(gpu):
    from engines import hip_engine
    bad, skipped, cells = goldens.check_engine(hip_engine, "sweep", sweep_parameters())
    assert not bad, "%d mismatches, first: %s" % (len(bad), bad[:10])
    assert len(skipped) <= SETS["sweep"] * cells, (len(skipped), cells)


@pytest.mark.gpu
@pytest.mark.parametrize("pair", ["degenerate_sweep", "offset_sweep", "long_sweep"])
def test_hip_second_passes_match_the_reference_on_other_parameters(gpu, pair):
    """k_ar_degenerate with AR orders 3 / 5 / 12, k_langevin_dd with five (m, r) fits in one plan; the long series."""
    from engines import hip_engine
    bad, skipped, cells = goldens.check_engine(hip_engine, pair, sweep_parameters())
    assert not bad, "%d mismatches, first: %s" % (len(bad), bad[:10])
    assert len(skipped) <= SETS_HIP[pair] * cells, (len(skipped), cells)


@pytest.mark.parametrize("params, needle", [
    ({"permutation_entropy": [{"tau": 1, "dimension": 11}]}, "dimension must be in [2, 10]"),
])
def test_parameters_beyond_the_tables_are_refused_by_name(params, needle):
    """tsfa_validate_spec (tsfa_host_tables.h) is what tsfa_plan_create runs on every spec: a value a kernel cannot serve
    fails the plan with the calculator and the bound, it is never computed as something else.  (Round 6: the other seven
    bounds this test used to list are gone -- tests/test_param_beyond.py compares those plans with the reference.)"""
    import numpy as np
    with pytest.raises(RuntimeError) as e:
        emul_engine(params, np.arange(50.0), np.array([0, 50], dtype=np.int64))
    assert needle in str(e.value) and list(params)[0] in str(e.value)


LIFTED = {"binned_entropy": [{"max_bins": b} for b in (256, 257, 1000, 5000)],
          "fourier_entropy": [{"bins": b} for b in (128, 129, 300, 1000)],
          "permutation_entropy": [{"tau": t, "dimension": d} for t, d in ((1, 8), (2, 8), (1, 9), (1, 10), (3, 10), (1, 7))]}


def _lifted_batch(dtype):
    import numpy as np
    rng = np.random.default_rng(66)
    lens = [5, 9, 10, 11, 30, 100, 257, 1024, 1500, 3000]
    series = [(np.cumsum(rng.standard_normal(n)) if i % 2 else rng.standard_normal(n)).astype(dtype) for i, n in enumerate(lens)]
    values = np.concatenate(series)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    return series, values, offsets


def test_bounds_lifted_in_round_6_emulation():
    """Round-5 VERDICT missing #2: binned_entropy beyond 256 bins and fourier_entropy beyond 128 (the LDS counters are swept in
    rounds), permutation_entropy of dimension 8 .. 10 (sort-and-count of the pattern codes instead of a histogram over up to
    3 628 800 patterns) used to be refused; the reference takes any value.  Tie-free float64 series: the oracle's
    argsort(argsort) ranks are then independent of numpy's sort kind."""
    import numpy as np
    from parity import compare
    series, values, offsets = _lifted_batch(np.float64)
    names, got = emul_engine(LIFTED, values, offsets)
    onames, want = oracle_engine(LIFTED, values, offsets)
    assert names == onames
    bad = compare(names, got, want, series)
    assert not bad, bad[:8]


def test_permutation_entropy_of_high_dimension_ranks_ties_by_position():
    """Tied windows: the kernels rank ties stably (numpy's scalar argsort); a direct restatement with Python's stable sort."""
    import math
    import numpy as np
    rng = np.random.default_rng(8)
    x = np.round(rng.standard_normal(400), 0)
    for tau, D in ((1, 8), (2, 9), (1, 10)):
        counts = {}
        for t in range((len(x) - D) // tau + 1):           # fc.py _into_subchunks: D consecutive samples, a window every tau
            w = x[t * tau:t * tau + D]
            ranks = tuple(np.argsort(np.argsort(w, kind="stable"), kind="stable"))
            counts[ranks] = counts.get(ranks, 0) + 1
        tot = sum(counts.values())
        want = -sum(c / tot * math.log(c / tot) for c in counts.values())
        names, got = emul_engine({"permutation_entropy": [{"tau": tau, "dimension": D}]}, x, np.array([0, len(x)], dtype=np.int64))
        assert abs(got[0, 0] - want) <= 1e-9 * abs(want), (tau, D, got[0, 0], want)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_hip_bounds_lifted_in_round_6(gpu, dtype):
    import numpy as np
    from engines import hip_engine
    from parity import compare
    series, values, offsets = _lifted_batch(np.dtype(dtype).type)
    names, got = hip_engine(LIFTED, values, offsets)
    onames, want = oracle_engine(LIFTED, values.astype(np.float64), offsets)
    assert names == onames
    bad = compare(names, got, want, [s.astype(np.float64) for s in series])
    assert not bad, bad[:8]


def test_an_ar_column_beyond_the_order_does_not_starve_the_others():
    """ar_coefficient coeff > k is NaN without a fit (fc.py:1500).  The columns behind the first one read the fit the first one
    cached: a plan whose FIRST column was such a NaN left the cache empty (a random-parameter fuzz find; the device always
    takes that path, the emulation on its even series)."""
    import numpy as np
    from parity import compare
    rng = np.random.default_rng(5)
    series = [rng.standard_normal(n) for n in (68, 25, 58, 55, 66, 31, 97, 77)]
    values = np.concatenate(series)
    offsets = np.concatenate([[0], np.cumsum([len(s) for s in series])]).astype(np.int64)
    params = {"ar_coefficient": [{"coeff": 16, "k": 15}, {"coeff": 9, "k": 15}, {"coeff": 0, "k": 15}]}
    names, got = emul_engine(params, values, offsets)
    onames, want = oracle_engine(params, values, offsets)
    assert names == onames
    assert np.isnan(got[:, 0]).all() and np.isfinite(got[np.array([len(s) >= 32 for s in series]), 1]).all()
    bad = compare(names, got, want, series)
    assert not bad, bad[:6]


def test_from_columns_round_trips_every_name_of_the_sweep(monkeypatch):
    """settings.from_columns (the reference's settings.py:24-84: what feature selection hands back) on the 229 names of the
    sweep, the permutation_entropy sets and the ADF lag selections: tuples, negative and exponent-form numbers, None."""
    import warnings
    from perm_cases import SETS as PERM_SETS
    from param_cases import adf_autolag_parameters
    from tsfresh_amd.feature_extraction import settings
    from tsfresh_amd.feature_extraction.plan import compile_fc_parameters
    sets = [sweep_parameters()] + list(PERM_SETS.values())
    sets += [{"augmented_dickey_fuller": [p for p in adf_autolag_parameters()["augmented_dickey_fuller"] if p["autolag"] == al]}
             for al in ("BIC", "t-stat", None)]
    for params in sets:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            names = ["value__" + n for n in compile_fc_parameters(params).names]
            back = settings.from_columns(names)["value"]
            again = ["value__" + n for n in compile_fc_parameters(back).names]
        assert sorted(again) == sorted(names)
