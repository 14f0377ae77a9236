"""Pin the oracle: it must reproduce what the REAL reference returned (tests/golden/ref_*.npz, generated in the
build container by tests/golden/gen_golden_main.py / gen_golden_conda.py)."""
import os

import numpy as np
import pytest

import goldens
from engines import emul_engine, oracle_engine, oracle_engine_parallel
from tsfresh_amd.feature_extraction.settings import ComprehensiveFCParameters

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_golden_has_all_783_columns():
    g = goldens.load("main")
    assert len(g["names"]) == 783 and g["matrix"].shape[1] == 783  # 788 minus 5 linear_trend_timewise


@pytest.mark.parametrize("pair", sorted(goldens.PAIRS))
@pytest.mark.parametrize("engine", [oracle_engine, emul_engine], ids=["oracle", "emul"])
def test_engine_matches_reference_golden(engine, pair):
    if engine is oracle_engine and pair.startswith("long"):
        engine = oracle_engine_parallel   # 4096 .. 8192-sample series: the same oracle, one process per series
    bad, skipped, cells = goldens.check_engine(engine, pair, ComprehensiveFCParameters())
    assert not bad, "%d mismatches, first: %s" % (len(bad), bad[:10])
    # the exclusions of tests/parity.py stay marginal: < 0.5 % of the ordinary series, < 2 % of the set that was
    # built out of degenerate series
    assert len(skipped) <= (0.02 if pair.startswith("degenerate") else 0.005) * cells, (len(skipped), cells)


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


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_emul_matches_oracle_on_config3_rows(dtype):
    """CPU twin of tests/test_gpu_parity.py::test_config3_comprehensive_1024: the 16 structured series of exactly 1024
    samples through the g++ build of the kernel sources."""
    import cases
    from parity import compare
    rows = [r.astype(np.float64) for r in cases.config3_rows(dtype)]
    values = np.concatenate(rows)
    offsets = np.arange(len(rows) + 1, dtype=np.int64) * 1024
    params = ComprehensiveFCParameters()
    names, got = emul_engine(params, values, offsets)
    onames, want = oracle_engine(params, values, offsets)
    skipped = []
    bad = compare(onames, got[:, [names.index(n) for n in onames]], want, rows, skipped=skipped)
    assert not bad, "%d mismatches, first: %s" % (len(bad), bad[:10])
    assert len(skipped) <= 0.01 * want.size, len(skipped)


def test_emul_agg_linear_trend_on_series_no_longer_than_the_chunk():
    """chunk_len >= len(x): every aggregate of that chunk length is NaN (fc.py:2171) -- the keyed evaluation handles up to
    four aggregates of one chunk length per sweep and has to clear all of them (a fuzz find: 'min' kept a stale value
    next to a NaN 'max'); lengths around the chunk lengths of ComprehensiveFCParameters."""
    from parity import compare
    rng = np.random.default_rng(77)
    lens = [1, 2, 3, 5, 5, 6, 10, 11, 49, 50, 51, 120]
    rows = [np.full(5, -1062.46435546875) if i == 4 else rng.standard_normal(n) for i, n in enumerate(lens)]
    values = np.concatenate(rows)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    params = {"agg_linear_trend": ComprehensiveFCParameters()["agg_linear_trend"],
              "index_mass_quantile": ComprehensiveFCParameters()["index_mass_quantile"],
              "linear_trend": ComprehensiveFCParameters()["linear_trend"]}
    names, got = emul_engine(params, values, offsets)
    onames, want = oracle_engine(params, values, offsets)
    assert names == onames
    assert np.array_equal(np.isnan(got), np.isnan(want))
    bad = compare(names, got, want, rows)
    assert not bad, bad[:10]


def test_emul_long_entropy_sweep_in_a_ragged_launch():
    """Series of 1025 .. 4096 samples share a launch whose work region is sized for the LONGEST of them (fewer
    tolerances per round); a shorter series must not take more tolerances per round than that region holds
    (a GPU fuzz find: 2192 samples next to 2500, six tolerances -> NaN for the last one).  The emulation sizes the region
    the device's way and checks canary words behind it."""
    from parity import compare
    rng = np.random.default_rng(41)
    lens = [1065, 2192, 2500]
    rows = [rng.standard_normal(n) for n in lens]
    values = np.concatenate(rows)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    params = {"approximate_entropy": ComprehensiveFCParameters()["approximate_entropy"], "sample_entropy": None}
    names, got = emul_engine(params, values, offsets)
    onames, want = oracle_engine(params, values, offsets)
    assert names == onames
    bad = compare(names, got, want, rows)
    assert not bad, bad[:10]


def test_emul_outputs_do_not_depend_on_what_the_scratch_held(monkeypatch):
    """The device never clears LDS between the series a CU works through: whatever a kernel reads from its scratch
    before writing it is the PREVIOUS series' data.  The emulation's stand-in buffers start as zeros, NaN, a huge and a
    small negative number in turn (TSFA_EMUL_POISON); every one of the 783 columns must come out bit-identical."""
    rng = np.random.default_rng(5)
    lens = [1, 2, 3, 4, 5, 7, 9, 12, 17, 30, 33, 64, 100, 255, 256, 257, 300, 777, 1024, 1500]
    rows = []
    for i, n in enumerate(lens):
        k = i % 5
        rows.append(rng.standard_normal(n) if k == 0 else np.round(rng.standard_normal(n) * 2) if k == 1
                    else np.cumsum(rng.standard_normal(n)) if k == 2 else np.full(n, 0.5) if k == 3
                    else 1e6 + rng.standard_normal(n))
    values = np.concatenate(rows)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    outs = []
    for mode in ("0", "1", "2", "3"):
        monkeypatch.setenv("TSFA_EMUL_POISON", mode)
        names, got = emul_engine(ComprehensiveFCParameters(), values, offsets)
        outs.append(got)
    for mode, got in enumerate(outs[1:], 1):
        same = (got == outs[0]) | (np.isnan(got) & np.isnan(outs[0]))
        bad = np.argwhere(~same)
        assert len(bad) == 0, [(lens[r], names[c], outs[0][r, c], got[r, c]) for r, c in bad[:8]]


def test_fuzz_rounds_on_the_emulated_kernel_sources():
    """Six rounds of profiles/fuzz_parity.py (random calculator / parameter subsets on ragged batches of structured,
    offset and rescaled series) through the g++ build of the kernel sources, its scratch poisoned: what the GPU box
    runs with the HIP library and what found this round's defects, kept alive in the CPU suite."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TSFA_FUZZ_ENGINE="emul", TSFA_EMUL_POISON="2")
    out = subprocess.run([sys.executable, os.path.join(root, "profiles", "fuzz_parity.py"), "6", "2026"], env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "TOTAL mismatches 0" in out.stdout, out.stdout[-3000:]


def test_emul_welch_segment_batches_on_very_long_series():
    """The 256-sample Welch segments are transformed in batches (fam_spectral.h: blk_fft_pow2_batch); the batch size is
    capped by the reduction scratch that carries the segment means -- a 20 000-sample series (156 segments) used to
    overflow it on the device (found by test_series_longer_than_lds_match_oracle)."""
    from parity import compare
    rng = np.random.default_rng(8)
    lens = [20000, 300, 9001, 1024, 640]
    series = [rng.standard_normal(n) if i % 2 == 0 else np.cumsum(rng.standard_normal(n)) for i, n in enumerate(lens)]
    values = np.concatenate(series)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    params = {"spkt_welch_density": [{"coeff": c} for c in (2, 5, 8)], "fourier_entropy": [{"bins": b} for b in (2, 3, 5, 10, 100)],
              "fft_aggregated": [{"aggtype": a} for a in ("centroid", "variance", "skew", "kurtosis")]}
    names, got = emul_engine(params, values, offsets)
    onames, want = oracle_engine(params, values, offsets)
    assert list(names) == list(onames)
    bad = compare(names, got, want, series)
    assert not bad, bad[:6]
