"""Series holding infinite samples and the settings they are run under (gen_golden_nonfinite.py, tests/test_nonfinite.py)."""
import numpy as np

BINNED = {"binned_entropy": [{"max_bins": 10}]}
AR = {"ar_coefficient": [{"coeff": c, "k": 10} for c in range(3)] + [{"coeff": 1, "k": 3}]}


def cases():
    rng = np.random.default_rng(20260930)
    out = []

    def add(name, n, plant):
        x = rng.standard_normal(n)
        for i, v in plant:
            x[i] = v
        out.append((name, x))

    add("finite", 64, [])
    add("posinf_mid", 64, [(10, np.inf)])
    add("neginf_mid", 64, [(20, -np.inf)])
    add("both", 64, [(3, np.inf), (40, -np.inf)])
    add("posinf_first", 64, [(0, np.inf)])
    add("posinf_last", 64, [(63, np.inf)])           # only the regressand of AutoReg holds it: no exception there
    add("posinf_second_last", 64, [(62, np.inf)])
    add("short_posinf", 15, [(2, np.inf)])           # n < 2 k + 1 for k = 10: AutoReg refuses first (NaN), k = 3 raises
    add("very_short_posinf", 6, [(1, np.inf)])       # n < 2 k + 1 for both orders
    add("len21_posinf", 21, [(5, np.inf)])           # the shortest series AutoReg(lags=10) sets up
    add("len20_posinf", 20, [(5, np.inf)])
    out.append(("all_posinf", np.full(24, np.inf)))
    out.append(("subnormal", np.array([1e-310, 2e-310, 3e-310, 0.0, 1e-310])))
    out.append(("subnormal_tiny", np.array([1e-320, 5e-324, 0.0, 3e-322, 2e-321, 7e-322])))
    out.append(("huge", np.array([1e300, -1e300, 5e299, 0.0, 2.5e299, -7e299])))
    # a range of a few ulps: numpy >= 2.0 refuses to cut it into 10 distinct edges ("Too many bins for data range")
    out.append(("ulp_range", 2.0 ** 53 + np.array([0.0, 2.0, 4.0, 8.0, 6.0, 2.0, 0.0, 16.0, 10.0, 12.0, 4.0, 14.0] * 3)))
    return out
