"""Quantized inputs (rounded sensor values, counters, +-1 walks): the two DOCUMENTED divergences from the reference
(README "Where the reference disagrees with itself", profiles/r05_divergence_frequency.md) stay inside their predicates.

  * permutation_entropy on tied data: the kernels equal the reference under numpy's SCALAR (stable) argsort -- what the
    oracle restates and the *_nosimd fixtures pin -- in every cell; the reference under an AVX-512 numpy differs from
    THAT by up to 2 % in ~99 % of such series (measured), a property of the host's sort, not of the data.
  * number_cwt_peaks on integer-valued data: compared exactly wherever no CWT row holds a round-off tie (R8); on +-1 walks
    R8 fires for most series -- there the reference's own count changes with the numpy build (39 % of such series).
"""
import numpy as np
import pytest

from engines import emul_engine, hip_engine, oracle_engine_parallel
from parity import compare, excluded

PARAMS = {"permutation_entropy": [{"tau": 1, "dimension": d} for d in (3, 4, 5, 6, 7)],
          "number_cwt_peaks": [{"n": 1}, {"n": 5}]}


def _families(n_series=24, length=400):
    rng = np.random.default_rng(2025)
    return {"rounded": [np.round(rng.standard_normal(length), 1) for _ in range(n_series)],
            "poisson": [rng.poisson(3.0, length).astype(np.float64) for _ in range(n_series)],
            "walk": [np.cumsum(rng.choice([-1.0, 1.0], length)) for _ in range(n_series)]}


def _check(engine):
    for name, series in _families().items():
        values = np.concatenate(series)
        offsets = (np.arange(len(series) + 1) * len(series[0])).astype(np.int64)
        names, got = engine(PARAMS, values, offsets)
        onames, want = oracle_engine_parallel(PARAMS, values, offsets)
        assert names == onames
        skipped = []
        bad = compare(names, got, want, series, simd_golden=False, skipped=skipped)
        assert not bad, (name, bad[:5])
        # permutation_entropy is never excluded against the stable ranking; number_cwt_peaks only by R8
        assert all("number_cwt_peaks" in col for _, col in skipped), skipped[:5]
        if name != "walk":
            assert len(skipped) <= 0.05 * got.size, (name, len(skipped))
        # where R8 does not fire the counts are equal (checked by compare); where it fires the difference stays small
        cw = [j for j, c in enumerate(names) if "number_cwt_peaks" in c]
        assert np.max(np.abs(got[:, cw] - want[:, cw])) <= 6, name


def test_kernel_sources_on_quantized_inputs():
    _check(emul_engine)


@pytest.mark.gpu
def test_hip_on_quantized_inputs(gpu):
    _check(hip_engine)


def test_r8_fires_on_walks_and_not_on_rounded_noise():
    fam = _families(12, 300)
    fired = {k: sum(excluded("value__number_cwt_peaks__n_5", x) for x in v) for k, v in fam.items()}
    assert fired["rounded"] <= 1 and fired["walk"] >= 6, fired
