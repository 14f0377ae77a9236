"""number_cwt_peaks (fc.py:1320): the Ricker convolutions on the float64 matrix cores (fam_cwt.h: cwt_rows_mfma, the opt-in
instantiation of the plan option "cwt_mfma" -- measured slower than the float64 FMA tiles on gfx950) against the
register-tiled default and against the oracle (scipy.signal.find_peaks_cwt)."""
import numpy as np
import pytest

from engines import hip_engine, oracle_engine_parallel
from parity import compare

pytestmark = pytest.mark.gpu

# around the admission bound n >= 20 W + 66 of each width set, the 256-output tiles and the length classes
LENS = [51, 64, 100, 165, 166, 167, 200, 255, 256, 257, 300, 385, 386, 387, 511, 512, 513, 600, 767, 768, 769, 1000, 1023, 1024,
        1025, 1279, 1280, 1281, 1500, 2047, 2048, 2049, 3000, 4095, 4096]
PARAMS = {"number_cwt_peaks": [{"n": n} for n in (1, 2, 3, 5, 8, 12, 16)]}


def _batch(dtype, seed):
    rng = np.random.default_rng(seed)
    chunks = []
    for i, n in enumerate(LENS):
        kind = i % 4
        x = (rng.standard_normal(n) if kind == 0 else np.cumsum(rng.standard_normal(n)) if kind == 1 else
             rng.standard_normal(n) * np.linspace(0.1, 3.0, n) if kind == 2 else
             np.sin(np.arange(n) * 0.07) + 0.3 * rng.standard_normal(n))
        chunks.append(x.astype(dtype))
    values = np.concatenate(chunks)
    offsets = np.concatenate([[0], np.cumsum(LENS)]).astype(np.int64)
    return chunks, values, offsets


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_mfma_phase_a_equals_the_tiles_and_the_oracle(gpu, dtype, monkeypatch):
    chunks, values, offsets = _batch(dtype, 5)
    names2, tiles = hip_engine(PARAMS, values, offsets)
    names, got = hip_engine(PARAMS, values, offsets, options={"cwt_mfma": 1})
    assert names == names2
    assert np.array_equal(got, tiles), np.argwhere(got != tiles)[:10]
    onames, want = oracle_engine_parallel(PARAMS, values.astype(np.float64), offsets)
    assert onames == names
    skipped = []
    bad = compare(names, got, want, [c.astype(np.float64) for c in chunks], skipped=skipped)
    assert not bad, bad[:10]
    assert len(skipped) <= 0.02 * got.size, skipped[:10]


def test_mfma_phase_a_uniform_batch_and_nonfinite_samples(gpu, monkeypatch):
    """1024-sample series in one launch (the headline shape: one 256-output tile per wavefront and width); a series with an
    infinite sample keeps the VALU tiles (inf x 0 would poison the 15 other outputs of its MFMA row): same result as the
    default form."""
    rng = np.random.default_rng(9)
    x = rng.standard_normal((24, 1024)).astype(np.float32)
    x[3, 500] = np.inf
    values = x.reshape(-1)
    offsets = np.arange(25, dtype=np.int64) * 1024
    params = {"number_cwt_peaks": [{"n": 1}, {"n": 5}]}
    _, tiles = hip_engine(params, values, offsets)
    names, got = hip_engine(params, values, offsets, options={"cwt_mfma": 1})
    assert np.array_equal(got, tiles, equal_nan=True), np.argwhere(got != tiles)[:10]
    onames, want = oracle_engine_parallel(params, values.astype(np.float64), offsets)
    assert names == onames
    ok = np.ones(24, dtype=bool)
    ok[3] = False
    assert np.array_equal(got[ok], want[ok]), np.argwhere(got != want)[:10]
