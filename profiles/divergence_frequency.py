#!/usr/bin/env python
"""How often the two DOCUMENTED divergences from the reference occur on quantized inputs (VERDICT r4 "Next" #2d).

  1. permutation_entropy with ties inside a window: the reference ranks windows with np.argsort's default kind
     (fc.py:1866-1916), an UNSTABLE vectorised sort on AVX-512 / AVX2 hosts; the kernels (and numpy's scalar path) rank
     ties stably.  Measured: the REAL reference under this host's SIMD numpy vs the real reference under
     NPY_DISABLE_CPU_FEATURES (scalar sort) vs the kernel sources.
  2. number_cwt_peaks on integer-valued data: CWT rows hold exact ties on the humps, and which of two equal neighbours
     is the strict maximum is decided by the summation order of scipy's convolution.  Measured: the real
     scipy.signal.find_peaks_cwt (what fc.py:1320 calls) vs the kernel sources.

Kernel sources = tests/emul (the g++ build of tsfresh_amd/csrc/fam_*.h; on the device the same cells are compared by
tests/test_quantized_divergence.py).  Needs /root/reference.  ~4 minutes on 8 cores.
    python profiles/divergence_frequency.py [n_series] > profiles/r05_divergence_frequency.md"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
SIMD_FEATURES = "AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX2 FMA3"
DIMS = (3, 4, 5, 6, 7)
WIDTHS = (1, 5)
LENGTH = 600


def families(n_series, seed=2025):
    rng = np.random.default_rng(seed)
    return {
        "np.round(N(0,1), 1)": [np.round(rng.standard_normal(LENGTH), 1) for _ in range(n_series)],
        "Poisson(3)": [rng.poisson(3.0, LENGTH).astype(np.float64) for _ in range(n_series)],
        "+-1 random walk": [np.cumsum(rng.choice([-1.0, 1.0], LENGTH)) for _ in range(n_series)],
        "iid N(0,1) (control)": [rng.standard_normal(LENGTH) for _ in range(max(n_series // 10, 10))],
    }


def reference_worker(n_series):
    """(run as a subprocess, possibly under NPY_DISABLE_CPU_FEATURES) the real reference's calculators -> JSON on stdout"""
    import types
    import warnings

    class _Raiser(types.ModuleType):
        def __getattr__(self, item):
            if item.startswith("__"):
                raise AttributeError(item)
            raise RuntimeError("stubbed module %s.%s" % (self.__name__, item))
    for mod in ("pywt", "stumpy", "statsmodels", "statsmodels.tools", "statsmodels.tools.sm_exceptions", "statsmodels.tsa",
                "statsmodels.tsa.ar_model", "statsmodels.tsa.stattools", "statsmodels.stats", "statsmodels.stats.multitest"):
        sys.modules[mod] = types.ModuleType(mod)
    sys.modules["pywt"].cwt = None
    sys.modules["stumpy"].core = None
    sys.modules["statsmodels.tools.sm_exceptions"].MissingDataError = type("MissingDataError", (Exception,), {})
    sys.modules["statsmodels.tsa.ar_model"].AutoReg = None
    for nm in ("acf", "adfuller", "pacf"):
        setattr(sys.modules["statsmodels.tsa.stattools"], nm, None)
    sys.modules["statsmodels.stats.multitest"].multipletests = None
    sys.path.insert(0, "/root/reference")
    from tsfresh.feature_extraction import feature_calculators as fc
    from multiprocessing import Pool
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, series in families(n_series).items():
            pe = [[float(fc.permutation_entropy(x, tau=1, dimension=d)) for d in DIMS] for x in series]
            with Pool(min(os.cpu_count() or 1, 8)) as pool:
                cw = pool.starmap(_cwt_one, [(x,) for x in series], chunksize=8)
            out[name] = {"pe": pe, "cwt": cw}
    json.dump(out, sys.stdout)


def _cwt_one(x):
    sys.path.insert(0, "/root/reference")
    from tsfresh.feature_extraction import feature_calculators as fc
    return [float(fc.number_cwt_peaks(x, n)) for n in WIDTHS]


def run_reference(n_series, nosimd):
    env = dict(os.environ)
    if nosimd:
        env["NPY_DISABLE_CPU_FEATURES"] = SIMD_FEATURES
    txt = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", str(n_series)], env=env, check=True,
                         capture_output=True, text=True).stdout
    return json.loads(txt)


def kernels(series):
    from engines import emul_engine
    lens = [len(x) for x in series]
    values = np.concatenate(series)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    params = {"permutation_entropy": [{"tau": 1, "dimension": d} for d in DIMS], "number_cwt_peaks": [{"n": n} for n in WIDTHS]}
    names, got = emul_engine(params, values, offsets)
    pe = got[:, [names.index("value__permutation_entropy__dimension_%d__tau_1" % d) for d in DIMS]]
    cw = got[:, [names.index("value__number_cwt_peaks__n_%d" % n) for n in WIDTHS]]
    return pe, cw


def main():
    n_series = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    simd = run_reference(n_series, nosimd=False)
    scalar = run_reference(n_series, nosimd=True)
    fam = families(n_series)
    print("# Documented divergences on quantized inputs: measured frequency (profiles/divergence_frequency.py, %d series x %d samples per family)\n" % (n_series, LENGTH))
    print("numpy %s, this host's SIMD dispatch vs NPY_DISABLE_CPU_FEATURES=\"%s\"\n" % (np.__version__, SIMD_FEATURES))
    print("## permutation_entropy (tau = 1): cells that differ by more than 1e-6 relative\n")
    print("| input | dimension | reference(SIMD sort) vs reference(scalar sort) | kernels vs reference(scalar sort) | kernels vs reference(SIMD sort) | largest relative difference (SIMD vs scalar) |")
    print("|---|---|---|---|---|---|")
    rows_cwt = []
    for name, series in fam.items():
        pe_k, cw_k = kernels(series)
        a, s = np.array(simd[name]["pe"]), np.array(scalar[name]["pe"])
        for j, d in enumerate(DIMS):
            rel = np.abs(a[:, j] - s[:, j]) / np.maximum(np.abs(s[:, j]), 1e-300)
            ks = np.abs(pe_k[:, j] - s[:, j]) > 1e-6 * np.abs(s[:, j])
            ka = np.abs(pe_k[:, j] - a[:, j]) > 1e-6 * np.abs(a[:, j])
            print("| %s | %d | %d / %d (%.1f %%) | %d | %d (%.1f %%) | %.2e |" % (name, d, int((rel > 1e-6).sum()), len(series), 100.0 * (rel > 1e-6).mean(),
                                                                     int(ks.sum()), int(ka.sum()), 100.0 * ka.mean(), float(rel.max())))
        ca, cs = np.array(simd[name]["cwt"]), np.array(scalar[name]["cwt"])
        for j, n in enumerate(WIDTHS):
            diff = cw_k[:, j] - ca[:, j]
            rows_cwt.append("| %s | %d | %d / %d (%.1f %%) | %d | %d | %.2f | %d | %.1f |" % (
                name, n, int((diff != 0).sum()), len(series), 100.0 * (diff != 0).mean(), int((cw_k[:, j] != cs[:, j]).sum()),
                int((ca[:, j] != cs[:, j]).sum()),
                float(np.abs(diff).mean()), int(np.abs(diff).max()), float(ca[:, j].mean())))
    print("\n## number_cwt_peaks: series whose count differs from scipy.signal.find_peaks_cwt\n")
    print("| input | n | kernels vs reference (this host, SIMD numpy) | kernels vs reference (scalar numpy) | reference(SIMD) vs reference(scalar) | mean abs difference | max abs difference | mean count (reference) |")
    print("|---|---|---|---|---|---|---|---|")
    print("\n".join(rows_cwt))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--worker":
        reference_worker(int(sys.argv[2]))
    else:
        main()
