"""Generate tests/golden/ref_conda_ar_stuck.npz: the REAL reference's ar_coefficient (statsmodels AutoReg, second
interpreter, as gen_golden_conda.py) on the series of tests/golden/ar_stuck_cases.npz -- k + 1 noisy samples followed by a
stuck sensor, AR orders 20 .. 31: designs that are singular to 1e-17 .. 1e-24 of s_max WITHOUT an exact dependency among
their leading columns (found by the random-parameter fuzz, profiles/r04_fuzz_params_emul.log).

    /opt/conda/bin/python3.9 tests/golden/gen_golden_ar_stuck.py
"""
import os
import sys
import types
import warnings

warnings.filterwarnings("ignore")
import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402


class _MachAr:
    def __init__(self, *a, **k):
        fi = np.finfo(float)
        self.eps, self.tiny, self.huge, self.epsneg, self.xmin, self.xmax = fi.eps, fi.tiny, fi.max, fi.epsneg, fi.tiny, fi.max


if not hasattr(np, "MachAr"):
    np.MachAr = _MachAr
for _n in ("Int64Index", "Float64Index", "UInt64Index"):
    if not hasattr(pd, _n):
        setattr(pd, _n, pd.Index)
_st = types.ModuleType("stumpy")
_st.core = types.SimpleNamespace()
sys.modules["stumpy"] = _st
sys.modules["dask"] = None
sys.modules["distributed"] = None
sys.path.insert(0, "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))

from tsfresh.feature_extraction import feature_calculators as fc  # noqa: E402


def main():
    g = np.load(os.path.join(HERE, "ar_stuck_cases.npz"))
    ks, o, v = g["ks"], g["offsets"], g["values"]
    out = np.full((len(ks), int(ks.max()) + 1), np.nan)
    sv = np.full((len(ks), int(ks.max()) + 1), np.nan)
    for i, k in enumerate(ks):
        x = v[o[i]:o[i + 1]]
        res = dict(fc.ar_coefficient(x, [{"coeff": c, "k": int(k)} for c in range(int(k) + 1)]))
        for c in range(int(k) + 1):
            out[i, c] = res["coeff_%d__k_%d" % (c, k)]
        n = len(x)
        tt = np.arange(k, n)
        X = np.column_stack([np.ones(n - k)] + [x[tt - j] for j in range(1, k + 1)])
        sv[i, : k + 1] = np.linalg.svd(X, compute_uv=False)   # (as this interpreter's LAPACK returns them: goldens.py, R4)
    np.savez_compressed(os.path.join(HERE, "ref_conda_ar_stuck.npz"), ks=ks, coefficients=out, singular_values=sv)
    print("wrote ref_conda_ar_stuck.npz", out.shape)


if __name__ == "__main__":
    main()
