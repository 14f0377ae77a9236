"""Seeded input rows shared by the CPU (emulation) and GPU parity tests."""
import numpy as np


def config3_rows(dtype, L=1024):
    """The oracle-checked rows of BASELINE configs[2] (100k x 1024 Comprehensive): 16 series of exactly L samples that
    exercise what iid noise does not -- ties, integers, a constant and a two-valued series (every entropy counter at
    its maximum: the staged sweep's packed 10-bit counters at L = TSFA_ENT_STAGED_MAXN), heavy-tailed and smooth data."""
    rng = np.random.default_rng(20260923)
    t = np.arange(L, dtype=np.float64)
    rows = [
        rng.standard_normal(L),                                   # iid
        rng.standard_normal(L),
        np.cumsum(rng.standard_normal(L)),                        # tests/benchmark.py's randn().cumsum()
        np.cumsum(rng.standard_normal(L)) * 0.05,
        np.round(rng.standard_normal(L), 1),                      # ties: one decimal
        np.round(np.cumsum(rng.standard_normal(L)), 0),           # ties: integer-valued walk
        rng.integers(-3, 4, L).astype(np.float64),                # seven distinct values
        rng.integers(0, 2, L).astype(np.float64),                 # two-valued
        np.full(L, 0.1),                                          # constant (stuck sensor)
        np.full(L, -7.0),
        np.sin(0.05 * t) + 0.1 * rng.standard_normal(L),          # smooth + noise
        rng.standard_t(2, L),                                     # heavy tails
        np.abs(rng.standard_normal(L)) * 1e3 + 1e5,               # offset
        np.where(t < L // 2, 0.0, 1.0) + 0.01 * rng.standard_normal(L),   # step
        np.exp(0.004 * t) * (1 + 0.01 * rng.standard_normal(L)),  # trend with multiplicative noise
        np.round(np.sin(0.3 * t) * 4, 0),                         # quantised periodic-ish
    ]
    return [np.asarray(r, dtype=dtype) for r in rows]
