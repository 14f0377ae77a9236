"""Generate tests/golden/ref_perm.json: the REAL reference's permutation_entropy (read from /root/reference,
feature_calculators.py:1866-1916, unmodified) on tests/golden/perm_cases.py -- strides and dimension sets beyond the
tau = 1, D = 3 .. 7 of ComprehensiveFCParameters, series of 1 .. 2049 samples, ties, constant stretches.

    python tests/golden/gen_golden_perm.py        # needs /root/reference

Run under numpy's scalar loops (NPY_DISABLE_CPU_FEATURES, as gen_golden_main.py --nosimd): np.argsort's default kind is
an unstable vectorised sort on AVX-512 / AVX2 machines, so windows holding tied values would get CPU-dependent ranks; the
scalar path sorts such short rows by insertion (stable) -- the ranking the kernels and the oracle use."""
import json
import os
import sys

_SIMD_FEATURES = "AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX2 FMA3"
if os.environ.get("NPY_DISABLE_CPU_FEATURES") != _SIMD_FEATURES:
    os.environ["NPY_DISABLE_CPU_FEATURES"] = _SIMD_FEATURES
    os.execv(sys.executable, [sys.executable] + sys.argv)
import types

import numpy as np

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

from tsfresh.feature_extraction import feature_calculators as fc  # noqa: E402

from perm_cases import SETS, pe_series  # noqa: E402


def main():
    series = pe_series()
    doc = {"generator": "tests/golden/gen_golden_perm.py", "numpy": np.__version__, "n_series": len(series), "sets": {}}
    for name, params in sorted(SETS.items()):
        cols = []
        for p in params["permutation_entropy"]:
            col = []
            for x in series:
                v = fc.permutation_entropy(x, tau=p["tau"], dimension=p["dimension"])
                col.append(None if v != v else float(v))
            cols.append({"tau": p["tau"], "dimension": p["dimension"], "values": col})
        doc["sets"][name] = cols
    with open(os.path.join(HERE, "ref_perm.json"), "w") as f:
        json.dump(doc, f)
    print("wrote ref_perm.json:", {k: len(v) for k, v in doc["sets"].items()})


if __name__ == "__main__":
    main()
