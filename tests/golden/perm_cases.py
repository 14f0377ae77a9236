"""Inputs of tests/test_perm_entropy.py and of gen_golden_perm.py (which runs the real reference on them)."""
import numpy as np

ALL5 = {"permutation_entropy": [{"tau": 1, "dimension": d} for d in (3, 4, 5, 6, 7)]}        # ComprehensiveFCParameters
SETS = {
    "comprehensive": ALL5,                                                                                   # k_perm
    "comprehensive_stride3": {"permutation_entropy": [{"tau": 3, "dimension": d} for d in (7, 3, 5, 4, 6)]},  # k_perm, any order
    "subset": {"permutation_entropy": [{"tau": 2, "dimension": d} for d in (7, 3, 5)]},                       # other sets: k_sort, one by one
    "low": {"permutation_entropy": [{"tau": 3, "dimension": d} for d in (2, 3, 4)]},
    "two_dims": {"permutation_entropy": [{"tau": 1, "dimension": 6}, {"tau": 1, "dimension": 2}]},
    "mixed_strides": {"permutation_entropy": [{"tau": 1, "dimension": 3}, {"tau": 2, "dimension": 4}, {"tau": 1, "dimension": 5}]},
    "single": {"permutation_entropy": [{"tau": 1, "dimension": 7}]},
    # round 6: dimensions beyond the 5 040 patterns an LDS histogram sweeps (sort-and-count of the codes in k_sort)
    "high": {"permutation_entropy": [{"tau": 1, "dimension": 8}, {"tau": 1, "dimension": 9}, {"tau": 1, "dimension": 10},
                                     {"tau": 2, "dimension": 8}, {"tau": 5, "dimension": 10}]},
    "high_beside_the_kernel_of_its_own": {"permutation_entropy": [{"tau": 1, "dimension": d} for d in (3, 4, 5, 6, 7, 8, 10)]},
}


def pe_series():
    rng = np.random.default_rng(91)
    out = [rng.standard_normal(n) for n in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 20, 50, 129, 300, 1000, 1024, 1500, 2049)]
    out.append(np.round(rng.standard_normal(400), 1))                    # ties inside the windows: stable ranks
    out.append(rng.integers(0, 3, size=700).astype(np.float64))          # three distinct values
    out.append(np.full(90, 2.5))                                         # one pattern
    out.append(np.arange(300, dtype=np.float64))                         # one pattern
    out.append(np.tile([1.0, 3.0, 2.0], 200))                            # three patterns
    out.append(np.cumsum(rng.standard_normal(1024)))
    out.append(rng.standard_normal(1024).astype(np.float32).astype(np.float64))
    return out
