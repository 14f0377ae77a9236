"""Generate tests/golden/ref_roll.json: `roll_time_series` of the REAL reference
(tsfresh/utilities/dataframe_functions.py:377) on a few small frames, third-party stubs as in gen_golden_main.py.

    python tests/golden/gen_golden_roll.py     # needs /root/reference; writes tests/golden/ref_roll.json
"""
import json
import os
import sys
import warnings

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden_main  # noqa: E402,F401  (stubs + /root/reference on sys.path)
from tsfresh.utilities.dataframe_functions import roll_time_series  # noqa: E402

from gen_golden_roll_cases import roll_cases  # noqa: E402


def frame_to_json(df):
    d = df.copy()
    d["id"] = [list(map(lambda v: v.item() if hasattr(v, "item") else v, t)) for t in d["id"]]
    return {"index": [int(i) for i in d.index], "columns": list(map(str, d.columns)),
            "rows": json.loads(d.to_json(orient="values"))}


def main():
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, df, kw in roll_cases():
            res = roll_time_series(df.copy(), n_jobs=0, disable_progressbar=True, **kw)
            out[name] = frame_to_json(res)
    path = os.path.join(HERE, "ref_roll.json")
    json.dump(out, open(path, "w"))
    print("wrote", path, {k: len(v["rows"]) for k, v in out.items()})


if __name__ == "__main__":
    main()
