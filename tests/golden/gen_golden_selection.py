"""Generate tests/golden/ref_selection.json: relevance tables of the REAL reference
(tsfresh/feature_selection/relevance.py:calculate_relevance_table, read from /root/reference) with the real
scipy / statsmodels of the second interpreter of the build container:

    /opt/conda/bin/python3.9 tests/golden/gen_golden_selection.py
        numpy 1.26.4, scipy 1.7.1, pandas 2.3.3, statsmodels 0.12.2 (two attribute shims make it import)

Cases: binary and multiclass targets, real / binary / constant features, heavy ties, small classes (exact
Mann-Whitney), both FDR procedures.  The inputs are regenerated from the seeds by tests (numpy Generator streams are
stable across these numpy versions); the frames themselves are stored too, so the test does not depend on that.
"""
import json
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
for _m in ("stumpy", "pywt"):
    sys.modules[_m] = types.ModuleType(_m)
sys.modules["stumpy"].core = types.SimpleNamespace()
sys.modules["dask"] = None
sys.modules["distributed"] = None
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from selection_cases import CASES, make_case  # noqa: E402
from tsfresh.feature_selection.relevance import calculate_relevance_table  # noqa: E402

out = []
for name in CASES:
    X, y, kw = make_case(name)
    tab = calculate_relevance_table(X, y, n_jobs=0, **kw)
    rec = {"name": name, "kwargs": kw, "X": {c: [float(v) for v in X[c]] for c in X.columns}, "index": [int(i) for i in X.index],
           "y": [(v if isinstance(v, str) else (float(v) if y.dtype.kind == "f" else int(v))) for v in y], "y_index": [int(i) for i in y.index],
           "table_index": [str(i) for i in tab.index], "columns": list(tab.columns), "table": {}}
    for c in tab.columns:
        col = tab[c]
        if col.dtype == bool:
            rec["table"][c] = [bool(v) for v in col]
        elif col.dtype.kind in "fi":
            rec["table"][c] = [None if (isinstance(v, float) and np.isnan(v)) else float(v) for v in col]
        else:
            rec["table"][c] = [str(v) for v in col]
    out.append(rec)
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_selection.json"), "w") as f:
    json.dump(out, f)
print("wrote", len(out), "cases")
