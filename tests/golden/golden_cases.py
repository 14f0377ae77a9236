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


def pack(cases):
    values = np.concatenate([c[1] for c in cases])
    offsets = np.zeros(len(cases) + 1, dtype=np.int64)
    np.cumsum([len(c[1]) for c in cases], out=offsets[1:])
    return values, offsets
