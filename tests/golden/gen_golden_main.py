"""Generate tests/golden/ref_main.npz by running the REAL reference (read from /root/reference) in the main
interpreter of the build container.

`import tsfresh` fails here because pywt / statsmodels / stumpy are not installed, so those modules
are stubbed in sys.modules with objects that RAISE if ever used; the five calculators that need them
(cwt_coefficients, agg_autocorrelation, partial_autocorrelation, augmented_dickey_fuller, ar_coefficient) are
removed from the FCParameters and covered by gen_golden_conda.py instead.  Everything else -- 70 calculators,
701 columns -- is the unmodified reference code path `extraction._do_extraction_on_chunk`.

    python tests/golden/gen_golden_main.py        # needs /root/reference; writes tests/golden/ref_main.npz
    python tests/golden/gen_golden_main.py --set degenerate             # ref_main_degenerate.npz
    python tests/golden/gen_golden_main.py [--set S] --nosimd           # ref_main[_S]_nosimd.npz
    python tests/golden/gen_golden_main.py --params sweep               # ref_main_sweep.npz (param_cases.py)
    python tests/golden/gen_golden_main.py --params beyond              # ref_main_beyond.npz (param_cases.beyond_parameters)

--nosimd re-executes the interpreter with NPY_DISABLE_CPU_FEATURES set, so numpy's runtime dispatch falls back to its
scalar loops.  It exists for ONE calculator: permutation_entropy ranks every window with np.argsort's default kind
(fc.py:1866-1916), which on AVX-512 / AVX2 machines is an unstable vectorised sort, so windows holding tied values
get CPU-dependent ranks; numpy's scalar path sorts such short rows by insertion (stable), the ranking the kernels
and the oracle use.  The two fixtures differ in the permutation_entropy cells of tie-holding series and otherwise
by <= 5e-14 relative (libm-vs-SIMD rounding).
"""
import os
import sys

NOSIMD = "--nosimd" in sys.argv
_SIMD_FEATURES = "AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX2 FMA3"
if NOSIMD and os.environ.get("NPY_DISABLE_CPU_FEATURES") != _SIMD_FEATURES:
    os.environ["NPY_DISABLE_CPU_FEATURES"] = _SIMD_FEATURES
    os.execv(sys.executable, [sys.executable] + sys.argv)
import types
import warnings

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


class _Raiser(types.ModuleType):
    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)

        def _fail(*a, **k):
            raise RuntimeError("stubbed third-party module %s.%s was called" % (self.__name__, item))
        return _fail


for mod in ("pywt", "stumpy", "statsmodels", "statsmodels.tools", "statsmodels.tools.sm_exceptions", "statsmodels.tsa",
            "statsmodels.tsa.ar_model", "statsmodels.tsa.stattools", "statsmodels.stats", "statsmodels.stats.multitest"):
    sys.modules[mod] = _Raiser(mod)
sys.modules["statsmodels.tools.sm_exceptions"].MissingDataError = type("MissingDataError", (Exception,), {})
sys.path.insert(0, "/root/reference")

from tsfresh.feature_extraction import settings as ref_settings  # noqa: E402
from tsfresh.feature_extraction.extraction import _do_extraction_on_chunk  # noqa: E402

from golden_cases import CASE_SETS, pack  # noqa: E402

NEED_THIRD_PARTY = ("cwt_coefficients", "agg_autocorrelation", "partial_autocorrelation", "augmented_dickey_fuller",
                    "ar_coefficient")


def main():
    case_set = sys.argv[sys.argv.index("--set") + 1] if "--set" in sys.argv else "main"
    cases = CASE_SETS[case_set]()
    params = ref_settings.ComprehensiveFCParameters()
    sweep = "--params" in sys.argv and sys.argv[sys.argv.index("--params") + 1] == "sweep"
    if sweep:   # parameters away from the Comprehensive grids (param_cases.py) -> ref_main_sweep.npz
        from param_cases import sweep_parameters
        params = sweep_parameters()
    beyond = "--params" in sys.argv and sys.argv[sys.argv.index("--params") + 1] == "beyond"
    if beyond:  # values beyond the tuned kernels' tables (param_cases.beyond_parameters) -> ref_main_beyond.npz
        from param_cases import beyond_parameters
        params = beyond_parameters()
    full_names = {}
    for cls in ("ComprehensiveFCParameters", "EfficientFCParameters", "MinimalFCParameters"):
        p = getattr(ref_settings, cls)()
        res = _do_extraction_on_chunk((0, "value", pd.Series(np.arange(60.0) % 7)),
                                      {k: v for k, v in p.items() if k not in NEED_THIRD_PARTY}, None, False)
        full_names[cls] = [r[1] for r in res]
        full_names[cls + "_keys"] = list(p.keys())
    for k in NEED_THIRD_PARTY:
        params.pop(k, None)
    if sweep or beyond:
        full_names = {}
    names, rows = None, []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for label, x in cases:
            res = _do_extraction_on_chunk((label, "value", pd.Series(x)), params, None, False)
            cols = [r[1] for r in res]
            if names is None:
                names = cols
            assert cols == names
            rows.append([float(r[2]) for r in res])
    values, offsets = pack(cases)
    suffix = ("" if case_set == "main" else "_" + case_set) + ("_sweep" if sweep else "_beyond" if beyond else "") + ("_nosimd" if NOSIMD else "")
    out = os.path.join(HERE, "ref_main%s.npz" % suffix)
    np.savez_compressed(
        out, values=values, offsets=offsets, labels=np.array([c[0] for c in cases]), names=np.array(names),
        matrix=np.asarray(rows, dtype=np.float64),
        **{"names_" + k: np.array(v) for k, v in full_names.items()},
        versions=np.array(["numpy " + np.__version__, "pandas " + pd.__version__,
                           "scipy " + __import__("scipy").__version__, "python " + sys.version.split()[0]]))
    print("wrote", out, len(names), "columns x", len(rows), "series")


if __name__ == "__main__":
    main()
