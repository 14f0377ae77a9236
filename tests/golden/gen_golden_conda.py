"""Generate tests/golden/ref_conda.npz: the five calculators whose arithmetic lives in statsmodels / PyWavelets,
run through the REAL reference code (feature_calculators.py read from /root/reference) with the REAL libraries of
the second interpreter of the build container:

    /opt/conda/bin/python3.9 tests/golden/gen_golden_conda.py [--set S] [--params sweep | beyond | adf]
        numpy 1.26.4, scipy 1.7.1, pandas 2.3.3, pywt 1.1.1, statsmodels 0.12.2

statsmodels 0.12.2 does not import against that numpy/pandas as shipped (np.MachAr and pd.Int64Index are gone);
two attribute shims below make it import -- no statsmodels code is modified.  0.12.2 is below the reference's
floor (statsmodels>=0.13, setup.cfg:42): these vectors pin the algorithms (acf / pacf-ld / adfuller / AutoReg),
which did not change across that boundary, but they are the weakest pin of the suite and DESIGN.md says so.
`stumpy` and `dask` are absent/broken there and are stubbed/blocked; none of the five calculators touches them.
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
sys.path.insert(0, HERE)

import pywt  # noqa: E402
import statsmodels  # noqa: E402
from tsfresh.feature_extraction import settings as ref_settings  # noqa: E402
from tsfresh.feature_extraction.extraction import _do_extraction_on_chunk  # noqa: E402

from golden_cases import CASE_SETS, pack  # noqa: E402

CALCS = ("cwt_coefficients", "agg_autocorrelation", "partial_autocorrelation", "augmented_dickey_fuller", "ar_coefficient")


def main():
    case_set = sys.argv[sys.argv.index("--set") + 1] if "--set" in sys.argv else "main"
    cases = CASE_SETS[case_set]()
    full = ref_settings.ComprehensiveFCParameters()
    sweep = "--params" in sys.argv and sys.argv[sys.argv.index("--params") + 1] == "sweep"
    if sweep:   # parameters away from the Comprehensive grids (param_cases.py) -> ref_conda_sweep.npz
        from param_cases import sweep_parameters
        full = sweep_parameters()
    beyond = "--params" in sys.argv and sys.argv[sys.argv.index("--params") + 1] == "beyond"
    if beyond:  # values beyond the tuned kernels' tables -> ref_conda_beyond.npz
        from param_cases import beyond_parameters
        full = beyond_parameters()
    adf = "--params" in sys.argv and sys.argv[sys.argv.index("--params") + 1] == "adf"
    if adf:     # the other lag selections of adfuller -> ref_conda_*_adf.npz (one fixture per autolag value: a plan holds one)
        from param_cases import adf_autolag_parameters
        full = adf_autolag_parameters()
    params = {k: full[k] for k in CALCS if k in full}
    names, rows = None, []
    for label, x in cases:
        res = _do_extraction_on_chunk((label, "value", pd.Series(x)), params, None, False)
        cols = [r[1] for r in res]
        if names is None:
            names = cols
        assert cols == names
        rows.append([float(r[2]) for r in res])
    values, offsets = pack(cases)
    extra = {}
    if sweep or beyond:
        # Singular values of every AR(k) design [1, x[t-1] .. x[t-k]] AS THIS INTERPRETER'S LAPACK RETURNS THEM: the
        # reference's pinv cuts at 1e-15 s_max, and for an exactly rank-deficient design (a constant series) whether a
        # direction that does not exist comes back above that cut is round-off of the LAPACK build -- tests/parity.py R4
        # asks that question of the test interpreter's LAPACK, these rows let it ask the reference's own
        for k in sorted({p["k"] for p in params["ar_coefficient"]}):
            sv = np.full((len(cases), k + 1), np.nan)
            for i, (_, x) in enumerate(cases):
                x = np.asarray(x, dtype=np.float64)
                n = len(x)
                if n < 2 * k + 2 or not np.all(np.isfinite(x)):
                    continue
                tt = np.arange(k, n)
                X = np.column_stack([np.ones(n - k)] + [x[tt - j] for j in range(1, k + 1)])
                sv[i] = np.linalg.svd(X, compute_uv=False)
            extra["ar_sv_k%d" % k] = sv
    out = os.path.join(HERE, ("ref_conda.npz" if case_set == "main" else "ref_conda_%s.npz" % case_set).replace(".npz", "_sweep.npz" if sweep else "_beyond.npz" if beyond else "_adf.npz" if adf else ".npz"))
    np.savez_compressed(out, values=values, offsets=offsets, labels=np.array([c[0] for c in cases]),
                        names=np.array(names), matrix=np.asarray(rows, dtype=np.float64), **extra,
                        versions=np.array(["numpy " + np.__version__, "pandas " + pd.__version__,
                                           "pywt " + pywt.__version__, "statsmodels " + statsmodels.__version__,
                                           "python " + sys.version.split()[0]]))
    print("wrote", out, len(names), "columns x", len(rows), "series")


if __name__ == "__main__":
    main()
