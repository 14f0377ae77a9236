"""The chirp-z (Bluestein) transform of non-power-of-two lengths (fam_spectral.h: blk_rfft_bluestein), kernel sources on the
CPU: every tile shape -- even lengths as n / 2 complex points (M / T = 2, 4), odd lengths (M / T = 4, 8) -- from 17 to
32 767 samples against numpy's rfft and the oracle (fc.py:1067 fft_coefficient, fc.py:1123 fft_aggregated).  The device
twin: tests/test_gpu_parity.py::test_chirp_z_transform_on_every_tile_shape_and_in_several_launches."""
import numpy as np

from engines import emul_engine, oracle_engine
from parity import compare


def test_chirp_z_transform_matches_numpy_on_every_tile_shape(monkeypatch):
    monkeypatch.setenv("TSFA_EMUL_BLUESTEIN_MIN", "16")
    rng = np.random.default_rng(5)
    lens = [17, 18, 20, 22, 30, 31, 34, 62, 66, 126, 130, 254, 258, 510, 514, 1022, 1026, 2046, 2050, 4094, 4098, 5001, 6000,
            8190, 8191, 8194, 12345, 16382, 16386, 20001, 32766, 32767]
    series = []
    for n in lens:   # the emulation sends EVEN series indices down the chirp-z route, odd ones down the Goertzel sweep
        series.append(rng.standard_normal(n) + (3.0 if n % 3 == 0 else 0.0))
        series.append(np.cumsum(rng.standard_normal(n)))
    values = np.concatenate(series)
    offsets = np.concatenate([[0], np.cumsum([len(x) for x in series])]).astype(np.int64)
    params = {"fft_coefficient": [{"attr": a, "coeff": k} for a in ("real", "imag", "abs", "angle") for k in (0, 1, 2, 3, 7, 50, 99)],
              "fft_aggregated": [{"aggtype": t} for t in ("centroid", "variance", "skew", "kurtosis")]}
    names, got = emul_engine(params, values, offsets)
    worst = 0.0
    for i in range(0, len(series), 2):
        x = series[i]
        X = np.fft.rfft(x)
        scale = float(np.abs(x).sum())
        for j, nm in enumerate(names):
            if "fft_coefficient" not in nm or "angle" in nm:
                continue
            k = int(nm.split("coeff_")[1])
            if k >= len(X):
                assert np.isnan(got[i, j])
                continue
            a = nm.split('attr_"')[1].split('"')[0]
            want = {"real": X[k].real, "imag": X[k].imag, "abs": abs(X[k])}[a]
            worst = max(worst, abs(got[i, j] - want) / scale)
    assert worst < 5e-15, worst      # measured 5.9e-16 (the Goertzel sweep on the same series: 1.9e-9)
    onames, want = oracle_engine(params, values, offsets)
    assert names == onames
    bad = compare(names, got, want, series)
    assert not bad, bad[:8]
