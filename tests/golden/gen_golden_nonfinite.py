"""Generate tests/golden/ref_nonfinite.json: what the REAL reference does with series that hold +-inf (and, for
binned_entropy, series of subnormal / 1e300 magnitude): the exception it raises (type name + message) or its values.

    python tests/golden/gen_golden_nonfinite.py                           # binned_entropy (main interpreter)
    /opt/conda/bin/python3.9 tests/golden/gen_golden_nonfinite.py         # ar_coefficient (statsmodels 0.12.2)

Each interpreter merges its half into the same file.  The module stubs / attribute shims are those of
gen_golden_main.py / gen_golden_conda.py.
"""
import json
import os
import sys
import types
import warnings

warnings.filterwarnings("ignore")
import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
CONDA = sys.version_info[:2] == (3, 9)

if CONDA:
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
else:
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

from tsfresh.feature_extraction.extraction import _do_extraction_on_chunk  # noqa: E402

from nonfinite_cases import AR, BINNED, cases  # noqa: E402


def main():
    key, params = ("ar_coefficient", AR) if CONDA else ("binned_entropy", BINNED)
    block = {}
    for name, x in cases():
        try:
            res = _do_extraction_on_chunk((name, "value", pd.Series(x)), params, None, False)
            block[name] = {"names": [r[1] for r in res], "values": [repr(float(r[2])) for r in res]}
        except Exception as e:   # noqa: BLE001 - the exception IS the datum
            block[name] = {"raises": type(e).__name__, "message": str(e)}
    path = os.path.join(HERE, "ref_nonfinite.json")
    doc = json.load(open(path)) if os.path.exists(path) else {}
    doc[key] = block
    doc.setdefault("versions", {})[key] = {"python": sys.version.split()[0], "numpy": np.__version__}
    json.dump(doc, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(block, indent=1))


if __name__ == "__main__":
    main()
