#!/usr/bin/env python
"""Ad-hoc parity fuzz on the GPU box: random subsets of ComprehensiveFCParameters (random parameter sub-lists, random
order of appearance is the reference's dict order) on random ragged batches of mixed structured / random series, HIP
path against the oracle.    python profiles/fuzz_parity.py [rounds] [seed]
TSFA_FUZZ_ENGINE=emul runs the g++ build of the kernel sources instead (no GPU needed).
TSFA_FUZZ_PARAMS=random draws every parameter at random inside the range the kernels serve (since round 6 also beyond the tuned
kernels' tables: k_general, the double-double AR pass) instead of
from the Comprehensive grids: lags, chunk lengths, bin counts and coefficient indices beyond the series length included."""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from engines import emul_engine, hip_engine, oracle_engine  # noqa: E402
from parity import compare  # noqa: E402
from tsfresh_amd.feature_extraction import settings  # noqa: E402


def make_series(rng, n, dtype=np.float64):
    """A structured or random series; a third of them are then moved off zero and rescaled (offset 1e2 .. 1e9 times
    the spread, scale 1e-6 .. 1e6): the rank cuts of np.polyfit and statsmodels' pinv live there (round 2 never
    generated a non-zero mean, which is how the Langevin fit's missing truncation got through 80 clean rounds)."""
    x = make_base_series(rng, n)
    if os.environ.get("TSFA_FUZZ_EXTREME"):
        # round 6: the magnitudes the ordinary families never leave [1e-3, 1e9] for -- 1e+-300, subnormal
        # ranges, signed zeros, integers at 2^53 (TSFA_FUZZ_EXTREME=1: half of the series; float32 batches clip at 1e+-38)
        e = rng.integers(0, 8)
        f32 = (dtype == np.float32)
        hi, lo, sub, mant = ((30, 36), (-36, -30), (-44, -38), 24) if f32 else ((290, 300), (-300, -290), (-322, -308), 53)
        if e == 0:
            return x * 10.0 ** rng.uniform(*hi)
        if e == 1:
            return x * 10.0 ** rng.uniform(*lo)
        if e == 2:
            return x * 10.0 ** rng.uniform(*sub)              # subnormal in the batch's dtype
        if e == 3:
            z = np.where(rng.random(n) < 0.5, 0.0, -0.0)         # signed zeros among a few values
            m = rng.random(n) < 0.3
            z[m] = np.round(x[m], 1)
            return z
        if e == 4:
            return np.float64(2.0 ** mant) + rng.integers(-4, 5, n).astype(np.float64) * 2.0   # exactly representable neighbours
        if e == 5:
            return np.round(x) * 2.0 ** (mant - 1)
    u = rng.random()
    if u < 0.33:
        spread = float(np.std(x)) or 1.0
        x = x + spread * 10.0 ** rng.uniform(2, 9) * rng.choice([-1.0, 1.0])
    if u < 0.15 or u > 0.85:
        x = x * 10.0 ** rng.uniform(-6, 6)
    return x


def make_base_series(rng, n):
    k = rng.integers(0, 12)
    if k == 7:  # ramp (rank-deficient regressions)
        return rng.uniform(-5, 5) + rng.uniform(-2, 2) * np.arange(n)
    if k == 8:  # exactly periodic
        return np.resize(rng.integers(-2, 3, int(rng.integers(2, 6))).astype(float), n)
    if k == 9:  # noiseless sine, float32-rounded (ill-conditioned but resolvable)
        return np.sin(np.arange(n) * rng.uniform(0.02, 0.5)).astype(np.float32).astype(float)
    if k == 10:  # stuck sensor after a noisy start
        x = rng.standard_normal(n)
        x[n // 3:] = x[n // 3] if n >= 3 else x[-1]
        return x
    if k == 11:  # ramp + small noise
        return np.arange(n) * 0.5 + 10.0 ** float(rng.integers(-7, -1)) * rng.standard_normal(n)
    if k == 0:
        return np.full(n, float(rng.integers(-3, 4)) * 0.25)
    if k == 1:
        return rng.integers(-3, 4, n).astype(float)
    if k == 2:
        return np.cumsum(rng.standard_normal(n))
    if k == 3:
        return np.round(rng.standard_normal(n), 1)
    if k == 4:
        x = rng.standard_normal(n)
        x[rng.integers(0, n, max(1, n // 50))] *= 30
        return x
    if k == 5:
        return np.sin(np.arange(n) * rng.uniform(0.01, 1.0)) + 0.05 * rng.standard_normal(n)
    return rng.standard_normal(n) * 10.0 ** float(rng.integers(-3, 4))


def random_params(rng):
    """One random parameter list per parameterised calculator (a random 45 % of them per round)."""
    ri = lambda lo, hi: int(rng.integers(lo, hi + 1))   # noqa: E731
    rf = lambda lo, hi: float(np.round(rng.uniform(lo, hi), 3))   # noqa: E731
    several = lambda f: [f() for _ in range(ri(1, 3))]   # noqa: E731
    gen = {
        "ratio_beyond_r_sigma": lambda: {"r": rf(0.1, 12)},
        "large_standard_deviation": lambda: {"r": rf(0.0, 1.0)},
        "symmetry_looking": lambda: {"r": rf(0.0, 1.0)},
        "cid_ce": lambda: {"normalize": bool(rng.integers(0, 2))},
        "fft_coefficient": lambda: {"coeff": ri(0, 600), "attr": str(rng.choice(["real", "imag", "abs", "angle"]))},
        "fft_aggregated": lambda: {"aggtype": str(rng.choice(["centroid", "variance", "skew", "kurtosis"]))},
        "number_peaks": lambda: {"n": ri(1, 70)},
        "index_mass_quantile": lambda: {"q": rf(0.0, 1.0)},
        "number_cwt_peaks": lambda: {"n": ri(1, 16) if rng.random() < 0.7 else ri(17, 24)},   # (beyond 16: k_general, round 6)
        "linear_trend": lambda: {"attr": str(rng.choice(["pvalue", "rvalue", "intercept", "slope", "stderr"]))},
        "spkt_welch_density": lambda: {"coeff": ri(0, 200)},
        "change_quantiles": lambda: (lambda a, b: {"ql": min(a, b), "qh": max(a, b), "isabs": bool(rng.integers(0, 2)),
                                                   "f_agg": str(rng.choice(["mean", "var"]))})(rf(0, 1), rf(0, 1)),
        "time_reversal_asymmetry_statistic": lambda: {"lag": ri(0, 60)},
        "c3": lambda: {"lag": ri(0, 60)},
        "mean_n_absolute_max": lambda: {"number_of_maxima": ri(1, 400)},
        "binned_entropy": lambda: {"max_bins": ri(1, 256) if rng.random() < 0.7 else ri(257, 3000)},
        "approximate_entropy": lambda: {"m": ri(1, 3), "r": rf(0.0, 1.5)},
        "fourier_entropy": lambda: {"bins": ri(1, 128) if rng.random() < 0.7 else ri(129, 600)},
        "lempel_ziv_complexity": lambda: {"bins": ri(1, 255) if rng.random() < 0.7 else int(rng.choice([256, 300, 1000, 5000, 70000]))},
        "permutation_entropy": lambda: {"tau": ri(1, 4), "dimension": ri(2, 10)},
        "autocorrelation": lambda: {"lag": ri(0, 400)},
        "quantile": lambda: {"q": rf(0.0, 1.0)},
        "number_crossing_m": lambda: {"m": rf(-3, 3)},
        "value_count": lambda: {"value": float(rng.integers(-3, 4)) * 0.25},
        "range_count": lambda: {"min": rf(-2, 1), "max": rf(-1, 2)},
        "friedrich_coefficients": lambda: (lambda m: {"coeff": ri(0, m), "m": m, "r": ri(2, 64) if rng.random() < 0.7 else ri(65, 150)})(
            ri(1, 3) if rng.random() < 0.7 else ri(4, 6)),
        "max_langevin_fixed_point": lambda: {"m": ri(1, 3) if rng.random() < 0.7 else ri(4, 6), "r": ri(2, 64) if rng.random() < 0.7 else ri(65, 150)},
        "agg_linear_trend": lambda: {"attr": str(rng.choice(["rvalue", "intercept", "slope", "stderr"])), "chunk_len": ri(1, 120),
                                     "f_agg": str(rng.choice(["max", "min", "mean", "var"]))},
        "energy_ratio_by_chunks": lambda: (lambda k: {"num_segments": k, "segment_focus": ri(0, k - 1)})(ri(1, 20)),
        "count_above": lambda: {"t": rf(-2, 2)},
        "count_below": lambda: {"t": rf(-2, 2)},
        "agg_autocorrelation": lambda: {"f_agg": str(rng.choice(["mean", "median", "var"])), "maxlag": ri(1, 60) if rng.random() < 0.7 else ri(61, 400)},
        "partial_autocorrelation": lambda: {"lag": ri(0, 40) if rng.random() < 0.7 else ri(41, 200)},
        "ar_coefficient": lambda: (lambda k: {"coeff": ri(0, k + 1), "k": k})(ri(1, 31) if rng.random() < 0.8 else ri(32, 60)),
        "cwt_coefficients": lambda: (lambda ws: {"widths": ws, "coeff": ri(0, 40), "w": int(rng.choice(ws))})(
            tuple(sorted(set(int(v) for v in rng.integers(1, 24, size=ri(1, 4)))))),
    }
    al = [None, "AIC", "BIC", "t-stat"][ri(0, 3)]   # one lag selection per plan (tsfa_validate_plan)
    gen["augmented_dickey_fuller"] = lambda: {"attr": str(rng.choice(["teststat", "pvalue", "usedlag"])), "autolag": al}
    params = {}
    for nm, g in gen.items():
        if rng.random() < 0.45:
            seen, lst = set(), []
            for p in several(g):
                key = repr(sorted(p.items()))
                if key not in seen:
                    seen.add(key)
                    lst.append(p)
            params[nm] = lst
    return params or {"quantile": [{"q": 0.5}]}


SENTINEL = 123456.789


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    # every plan of the fuzz pre-fills the matrix with a sentinel (read when the plan is built): a cell no kernel wrote
    # shows up as SENTINEL instead of a stale value of an earlier round (round-4 ADVICE)
    if os.environ.get("TSFA_FUZZ_ENGINE") != "emul":
        os.environ.setdefault("TSFA_DEBUG_FILL", repr(SENTINEL))
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    full = settings.ComprehensiveFCParameters()
    names_all = list(full.keys())
    total_bad = 0
    for r in range(rounds):
        pick = [nm for nm in names_all if rng.random() < 0.45]
        params = {}
        if os.environ.get("TSFA_FUZZ_PARAMS") == "random":
            params = random_params(rng)
            pick = list(params)
        for nm in ([] if os.environ.get("TSFA_FUZZ_PARAMS") == "random" else pick):
            pl = full[nm]
            if pl is None:
                params[nm] = None
            else:
                sub = [p for p in pl if rng.random() < 0.6] or [pl[0]]
                params[nm] = sub
        # TSFA_FUZZ_MAXLENS="40,300,1024,1024,2500,4500": other length mixes (the default keeps a round within seconds)
        maxlen = int(rng.choice([int(v) for v in os.environ.get("TSFA_FUZZ_MAXLENS", "40,300,1024,1024,2500").split(",")]))
        lens = rng.integers(1, maxlen + 1, size=int(rng.integers(3, 14)))
        dtype = np.float32 if rng.random() < 0.6 else np.float64
        series = [make_series(rng, int(n), dtype).astype(dtype) for n in lens]
        values = np.concatenate(series)
        offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            engine = emul_engine if os.environ.get("TSFA_FUZZ_ENGINE") == "emul" else hip_engine
            names, got = engine(params, values, offsets)
            try:
                names_o, want = oracle_engine(params, values.astype(np.float64), offsets)
            except (ValueError, TypeError) as e:   # (TypeError: np.polyfit on the empty frame friedrich_coefficients is left with when every bin mean overflowed)   # the REFERENCE raises here (np.histogram: "Too many bins for data range" ...): no value to compare
                print("round", r, "skipped: the oracle raises as the reference does:", str(e)[:80])
                continue
        assert names == names_o, (names[:3], names_o[:3])
        unwritten = np.argwhere(got == SENTINEL)
        if len(unwritten):
            total_bad += len(unwritten)
            print("round", r, "UNWRITTEN CELLS", [(int(i), names[int(j)]) for i, j in unwritten[:6]])
        bad = compare(names, got, want, [values[offsets[i]:offsets[i + 1]].astype(np.float64) for i in range(len(series))])
        total_bad += len(bad)
        dump = os.environ.get("TSFA_FUZZ_DUMP")   # directory: the offending series as .npy, for adjudication
        for bmsg in bad[:3]:
            si = int(bmsg.split()[1])
            print("   offending series", si, repr(values[offsets[si]:offsets[si + 1]].astype(np.float64).tolist()[:60]))
            if dump:
                os.makedirs(dump, exist_ok=True)
                np.save(os.path.join(dump, "round%d_series%d.npy" % (r, si)), values[offsets[si]:offsets[si + 1]])
        print("round", r, "calcs", len(pick), "cols", len(names), "series", len(lens), "maxlen", maxlen, dtype.__name__,
              "mismatches", len(bad), bad[:4])
    print("TOTAL mismatches", total_bad)


if __name__ == "__main__":
    main()
