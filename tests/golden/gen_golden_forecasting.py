"""Generate tests/golden/ref_forecasting.json: `make_forecasting_frame` of the REAL reference
(tsfresh/utilities/dataframe_functions.py:606), third-party stubs as in gen_golden_main.py.

    python tests/golden/gen_golden_forecasting.py
"""
import json
import os
import sys
import warnings

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden_main  # noqa: E402,F401
from tsfresh.utilities.dataframe_functions import make_forecasting_frame  # noqa: E402

from forecasting_cases import forecasting_cases  # noqa: E402


def dump(df, y):
    d = df.copy()
    d["id"] = [list(map(lambda v: v.item() if hasattr(v, "item") else str(v), t)) for t in d["id"]]
    d["time"] = [str(v) for v in d["time"]]
    return {"columns": list(map(str, d.columns)), "rows": json.loads(d.to_json(orient="values")),
            "y_index": [[str(a) for a in t] for t in y.index], "y": [float(v) for v in y]}


out = {}
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for name, x, kw in forecasting_cases():
        df, y = make_forecasting_frame(x, **kw)
        out[name] = dump(df, y)
json.dump(out, open(os.path.join(HERE, "ref_forecasting.json"), "w"))
print("wrote", {k: len(v["rows"]) for k, v in out.items()})
