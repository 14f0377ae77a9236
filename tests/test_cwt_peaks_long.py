"""number_cwt_peaks on series whose noise window is LONG (beyond ~1400 samples the 10th percentile of the window is no longer
among its eight smallest values).  Round 6 replaced "argsort row 0 + walk the order" by a decision from window COUNTS against
thresholds at +-|signal| (fam_cwt.h: cwt_filter_counts); what the counts leave open -- a threshold between the two order
statistics, values EQUAL to +-|signal| -- goes through a second list that reads the two order statistics next to the
threshold.  The inputs below are chosen for that second list: periodic and quantized series (every window holds copies of
the maximum), plateaus, sign-symmetric series (values equal to -|signal|), a window that is all ties.  Counts are integers:
compared exactly with the oracle (fc.py:1320: scipy's own find_peaks_cwt) -- except where tests/parity.py R8 applies: a
transform row with neighbours equal up to round-off has no defined set of strict maxima (the square wave's plateaus of
1.104084545991277 come out as exact ties from scipy's convolution and as 1-ulp steps from a different summation order)."""
import numpy as np
import pytest

from engines import emul_engine, oracle_engine
from parity import compare

PARAMS = {"number_cwt_peaks": [{"n": n} for n in (1, 5, 3)]}


def long_series():
    rng = np.random.default_rng(2026)
    t = np.arange(6000)
    out = [
        rng.standard_normal(1500),
        rng.standard_normal(4096),
        np.cumsum(rng.standard_normal(3000)),
        np.sin(t[:2400] * 2 * np.pi / 37.0),                                  # periodic: every maximum has exact copies
        np.tile(rng.standard_normal(50), 60),                                  # period 50, 3000 samples: ties everywhere
        np.tile([0.0, 1.0, 0.0, -1.0], 500),                                   # values equal to +-|signal|
        rng.integers(-3, 4, size=2500).astype(np.float64),                     # quantized
        rng.integers(0, 2, size=1800).astype(np.float64),
        np.round(np.cumsum(rng.standard_normal(5000)) * 0.5),                  # quantized walk: plateaus
        np.concatenate([np.zeros(800), rng.standard_normal(700), np.zeros(900)]),   # windows that are all zeros
        np.sin(t * 0.01) + 1e-3 * rng.standard_normal(6000),
        (t[:2048] % 64 < 32).astype(np.float64),                               # square wave
        1e6 + rng.standard_normal(2000),
        rng.standard_normal(1430), rng.standard_normal(1421), rng.standard_normal(1420),   # either side of the switch
    ]
    return out


def test_counting_filter_equals_the_oracle_on_the_emulated_kernel_sources():
    series = long_series()
    values = np.concatenate(series)
    offsets = np.concatenate([[0], np.cumsum([len(s) for s in series])]).astype(np.int64)
    names, got = emul_engine(PARAMS, values, offsets)
    onames, want = oracle_engine(PARAMS, values, offsets)
    assert list(names) == list(onames)
    skipped = []
    bad = compare(list(names), got, want, series, skipped=skipped)
    assert not bad, bad[:10]
    assert len(skipped) <= 15, skipped   # (R8: the five series whose width-1 row holds exact or 1-ulp ties between neighbours)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_counting_filter_equals_the_oracle_on_the_device(gpu, dtype):
    from engines import hip_engine
    rng = np.random.default_rng(9)
    series = [s.astype(dtype) for s in long_series()] + [rng.standard_normal(n).astype(dtype) for n in (8192, 7000, 12000)]
    values = np.concatenate(series)
    offsets = np.concatenate([[0], np.cumsum([len(s) for s in series])]).astype(np.int64)
    onames, want = oracle_engine(PARAMS, values.astype(np.float64), offsets)
    names, got = hip_engine(PARAMS, values, offsets)
    assert list(names) == list(onames)
    skipped = []
    bad = compare(list(names), got, want, [s.astype(np.float64) for s in series], skipped=skipped)
    assert not bad, bad[:10]
    assert len(skipped) <= 15, skipped
