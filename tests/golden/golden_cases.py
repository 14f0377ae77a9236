"""Seeded input series shared by the golden generators and the tests (pure numpy, no reference needed)."""
import numpy as np


def golden_series():
    """-> list[(label, float64 ndarray)].  float32-representable values where the label says f32."""
    rng = np.random.default_rng(20240921)
    cases = []
    for i in range(3):
        cases.append(("randn_f32_300_%d" % i, rng.standard_normal(300, dtype=np.float32).astype(np.float64)))
    cases.append(("walk_f32_256", np.cumsum(rng.standard_normal(256, dtype=np.float32)).astype(np.float32).astype(np.float64)))
    cases.append(("randn_f32_1024", rng.standard_normal(1024, dtype=np.float32).astype(np.float64)))
    cases.append(("walk_f64_1000", np.cumsum(rng.standard_normal(1000))))
    cases.append(("ints_dup_100", rng.integers(-3, 4, size=100).astype(np.float64)))
    cases.append(("const_0p1_50", np.full(50, 0.1)))
    cases.append(("decimals_200", np.round(rng.standard_normal(200), 1)))
    cases.append(("randn_f64_37", rng.standard_normal(37)))
    cases.append(("sine_noise_512", np.sin(np.arange(512) * 0.1) + 0.1 * rng.standard_normal(512)))
    for n in (1, 2, 3, 4, 5, 10, 21, 22, 23, 45):
        cases.append(("short_%d" % n, rng.standard_normal(n)))
    cases.append(("zeros_30", np.zeros(30)))
    cases.append(("ramp_64", np.arange(64, dtype=np.float64)))
    cases.append(("big_scale_128", 1e6 + 1e3 * rng.standard_normal(128)))
    return cases


def degenerate_series():
    """Series on which the regressions of the path are rank-deficient or nearly so (stuck sensors, ramps, periodic
    signals, noiseless float32 sines): the reference answers with the pseudo-inverse's minimum-norm solution
    (statsmodels OLS pinv, np.polyfit's lstsq), and these vectors pin that behaviour."""
    rng = np.random.default_rng(20260923)
    t = np.arange(1024, dtype=np.float64)
    cases = [
        ("const_3_100", np.full(100, 3.0)),
        ("const_f32_64", np.full(64, np.float32(-1.7)).astype(np.float64)),
        ("const_big_1024", np.full(1024, 1.0e6 + 0.25)),
        ("zeros_100", np.zeros(100)),
        ("ramp_neg_100", 5.0 - 0.5 * t[:100]),
        ("ramp_f32_1000", (t[:1000] * 0.25 + 3.0).astype(np.float32).astype(np.float64)),
        ("alt_pm1_80", np.where(np.arange(80) % 2 == 0, 1.0, -1.0)),
        ("period4_120", np.tile([0.0, 1.0, 2.0, 1.0], 30)),
        ("period3_99", np.tile([1.0, 2.0, 4.0], 33)),
        ("step_60", np.concatenate([np.zeros(30), np.ones(30)])),
        ("sat_ramp_90", np.minimum(t[:90], 50.0)),
        ("glitch_ramp_70", np.concatenate([[5.0], t[1:70]])),
        ("quad_200", t[:200] ** 2),
        ("geom_40", 2.0 ** t[:40]),
        ("const_then_noise_100", np.concatenate([np.full(50, 2.0), rng.standard_normal(50)])),
        ("noise_then_const_100", np.concatenate([rng.standard_normal(50), np.full(50, 2.0)])),
        ("sine_f32_512", np.sin(t[:512] * 0.1).astype(np.float32).astype(np.float64)),
        ("sine_2dec_400", np.round(np.sin(t[:400] * 0.05), 2)),
        ("two_valued_1024", rng.integers(0, 2, size=1024).astype(np.float64)),
        ("const_1024", np.full(1024, 0.1)),
        ("tiny_noise_ramp_300", t[:300] + 1e-9 * rng.standard_normal(300)),
    ]
    return cases


def offset_series():
    """Series whose mean is far from zero relative to their spread (grid frequency 50 Hz +- 5 mHz, pressure
    101 325 +- 5 Pa, float64 counters / epoch seconds): np.polyfit's rank cut (friedrich_coefficients,
    max_langevin_fixed_point; fc.py:131-173) and statsmodels' pinv cut (ar_coefficient, augmented_dickey_fuller;
    fc.py:1459-1508, :499-545) decide what the reference returns there."""
    rng = np.random.default_rng(20260924)
    cases = []
    for q in (1e2, 1e3, 1e4, 3e4, 1e5, 1e6):
        for sigma in (1.0, 10.0):
            for kind in ("iid", "walk"):
                for dt in (np.float32, np.float64):
                    e = rng.standard_normal(300)
                    if kind == "walk":
                        e = np.cumsum(e) * 0.1
                    x = (q * sigma + sigma * e).astype(dt).astype(np.float64)
                    cases.append(("off_%g_s%g_%s_%s" % (q, sigma, kind, dt.__name__), x))
    t = np.arange(400, dtype=np.float64)
    cases.append(("neg_off_1e5_iid", -1e5 + rng.standard_normal(300)))
    cases.append(("off_1e7_iid", 1e7 + rng.standard_normal(300)))
    cases.append(("off_1e8_iid", 1e8 + rng.standard_normal(300)))
    cases.append(("off_1e9_iid", 1e9 + rng.standard_normal(300)))
    cases.append(("off_1e8_walk", 1e8 + np.cumsum(rng.standard_normal(400))))
    cases.append(("epoch_seconds_400", 1.7e9 + t))
    cases.append(("epoch_jitter_400", 1.7e9 + t + 0.01 * rng.standard_normal(400)))
    cases.append(("grid_hz_1024", 50.0 + 0.005 * rng.standard_normal(1024)))
    cases.append(("pressure_pa_512", 101325.0 + 5.0 * rng.standard_normal(512)))
    cases.append(("scaled_small_300", 1e-6 * (3.0 + rng.standard_normal(300))))
    cases.append(("scaled_big_300", 1e6 * (0.5 + rng.standard_normal(300))))
    return cases


def long_series():
    """Series beyond 1024 samples (BASELINE configs[4] is 4096 .. 8192): the reference changes algorithm there --
    agg_autocorrelation switches to an FFT autocorrelation past 1250 samples (fc.py:421-429), ADF's maxlag grows with n
    (fc.py:499-545: 12 (n / 100)^(1/4)), the CWT-peak noise window grows with n (fc.py:1320-1340), the entropy sweeps
    leave the kernels' 1024- and 4096-sample classes, the FFTs leave the power-of-two path.  Lengths just past every
    switch; iid / walk / tie-heavy / constant / periodic; float32 and float64; and three windows of ONE walk, the rolled
    layout of configs[4] (dataframe_functions.py:340-372)."""
    rng = np.random.default_rng(20260925)
    f32 = lambda x: np.asarray(x, dtype=np.float32).astype(np.float64)   # noqa: E731
    cases = [
        ("randn_f32_1025", f32(rng.standard_normal(1025))),
        ("randn_f32_1251", f32(rng.standard_normal(1251))),
        ("walk_f64_1251", np.cumsum(rng.standard_normal(1251))),
        ("const_1300", np.full(1300, 2.5)),
        ("randn_f32_2048", f32(rng.standard_normal(2048))),
        ("walk_f32_2048", f32(np.cumsum(rng.standard_normal(2048)))),
        ("ties_1dec_3000", np.round(rng.standard_normal(3000), 1)),
        ("sine_noise_3000", np.sin(np.arange(3000) * 0.02) + 0.05 * rng.standard_normal(3000)),
        ("randn_f64_4096", rng.standard_normal(4096)),
        ("walk_f32_4096", f32(np.cumsum(rng.standard_normal(4096)) * 0.1)),
        ("randn_f32_4097", f32(rng.standard_normal(4097))),
        ("ints_dup_5000", rng.integers(-3, 4, size=5000).astype(np.float64)),
        ("randn_f32_8192", f32(rng.standard_normal(8192))),
        ("walk_f64_8192", np.cumsum(rng.standard_normal(8192))),
        ("ar1_f32_6000", None),
    ]
    e = rng.standard_normal(6000)
    for t in range(1, 6000):
        e[t] += 0.9 * e[t - 1]
    cases[-1] = ("ar1_f32_6000", f32(e))
    walk = f32(np.cumsum(rng.standard_normal(5200)) * 0.05 + 20.0)
    for lo, hi in ((0, 4096), (1000, 5096), (1104, 5200)):   # windows of one series, as roll_time_series cuts them
        cases.append(("rolled_walk_%d_%d" % (lo, hi), walk[lo:hi].copy()))
    return cases


CASE_SETS = {"main": golden_series, "degenerate": degenerate_series, "offset": offset_series, "long": long_series}


def pack(cases):
    values = np.concatenate([c[1] for c in cases])
    offsets = np.zeros(len(cases) + 1, dtype=np.int64)
    np.cumsum([len(c[1]) for c in cases], out=offsets[1:])
    return values, offsets
