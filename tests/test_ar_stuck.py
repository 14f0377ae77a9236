"""ar_coefficient on designs that are singular to working precision WITHOUT an exact dependency among their leading columns:
k + 1 noisy samples followed by a stuck sensor, AR orders 20 .. 31 (tests/golden/ar_stuck_cases.npz, found by the
random-parameter fuzz).  The natural-order Cholesky of the normal equations shows no small pivot there -- every exact pivot
is above 6e-4 of its diagonal while the smallest singular value is 5e-24 s_max -- so the float64 pass has to estimate the
condition (fam_ar.h, orders above 12) and the double-double pass has to order the columns by pivoting (fam_ar_dd.h).
Expected values: statsmodels' own (tests/golden/gen_golden_ar_stuck.py, second interpreter)."""
import os

import numpy as np
import pytest

from engines import emul_engine, oracle_engine

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cases():
    c = np.load(os.path.join(G, "ar_stuck_cases.npz"))
    r = np.load(os.path.join(G, "ref_conda_ar_stuck.npz"))
    assert np.array_equal(c["ks"], r["ks"])
    return c["ks"], c["offsets"], c["values"], r["coefficients"], r["singular_values"]


def _check(engine):
    ks, o, v, want, sv = _cases()
    for i, k in enumerate(ks):
        k = int(k)
        x = v[o[i]:o[i + 1]]
        ratios = sv[i, :k + 1] / sv[i, 0]
        # (statsmodels' pinv cuts at 1e-15 s_max; these designs' smallest singular value is far below it and the next far above:
        #  the minimum-norm answer is well defined)
        assert ratios[-1] < 1e-16 and not np.any((ratios >= 1e-16) & (ratios <= 1e-12)), ratios[-3:]
        params = {"ar_coefficient": [{"coeff": c, "k": k} for c in range(k + 1)]}
        # twice in a row: the emulation takes the device's column-loop / epilogue split on even series only
        names, got = engine(params, np.concatenate([x, x]), np.array([0, len(x), 2 * len(x)], dtype=np.int64))
        for row in got:
            np.testing.assert_allclose(row, want[i, :k + 1], rtol=1e-6, atol=1e-9, err_msg="case %d, AR(%d)" % (i, k))


@pytest.mark.parametrize("engine", [oracle_engine, emul_engine], ids=["oracle", "emul"])
def test_stuck_sensor_designs_match_statsmodels(engine):
    _check(engine)


def _low_order_batches():
    """k + 1 noisy samples followed by a stuck sensor at the AR orders of the settings objects and around the old cutoff of
    the condition estimate (12): the natural-order Cholesky sees no small pivot on some of them (round-4 ADVICE; 1 of 60 such
    series at AR(12) was wrong in the first digit until the estimate ran for every order)."""
    rng = np.random.default_rng(12)
    for k in (2, 4, 8, 10, 12, 16):
        series = []
        for _ in range(60):
            n = int(rng.integers(2 * k + 6, 400))
            head = rng.standard_normal(k + 1) * rng.choice([1.0, 1e-3, 50.0])
            lvl = rng.choice([head[-1], 0.0, 3.25, -1e4])
            series.append(np.concatenate([head, np.full(n - k - 1, lvl)]))
        yield k, series


def _check_low_orders(engine):
    from parity import compare
    for k, series in _low_order_batches():
        values = np.concatenate(series)
        offsets = np.concatenate([[0], np.cumsum([len(x) for x in series])]).astype(np.int64)
        params = {"ar_coefficient": [{"coeff": c, "k": k} for c in range(k + 1)]}
        names, got = engine(params, values, offsets)
        onames, want = oracle_engine(params, values, offsets)
        assert names == onames
        bad = compare(names, got, want, series)
        assert not bad, "AR(%d): %d mismatches, first: %s" % (k, len(bad), bad[:4])


def test_stuck_sensor_designs_at_low_orders():
    _check_low_orders(emul_engine)


@pytest.mark.gpu
def test_hip_stuck_sensor_designs_at_low_orders(gpu):
    from engines import hip_engine
    _check_low_orders(hip_engine)


@pytest.mark.gpu
def test_hip_stuck_sensor_designs_match_statsmodels(gpu):
    from engines import hip_engine
    _check(hip_engine)
