"""Generate tests/golden/ref_impute.json: the REAL reference's imputation helpers
(tsfresh/utilities/dataframe_functions.py:49-214) on a small frame with NaN / +-inf / an all-non-finite column.

    python tests/golden/gen_golden_impute.py     # needs /root/reference; writes tests/golden/ref_impute.json
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
from tsfresh.utilities import dataframe_functions as ref  # noqa: E402


def impute_frame():
    rng = np.random.default_rng(11)
    x = rng.standard_normal((12, 6)).round(3)
    x[1, 0] = np.nan; x[5, 0] = np.inf; x[7, 0] = -np.inf
    x[:, 1] = np.nan
    x[2, 2] = np.inf; x[3, 2] = np.inf
    x[0, 3] = -np.inf; x[4, 3] = np.nan; x[9, 3] = np.nan
    x[:, 4] = [np.inf, -np.inf, np.nan] * 4
    return pd.DataFrame(x, columns=list("abcdef"), index=np.arange(100, 112))


def main():
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        df = impute_frame()
        mx, mn, med = ref.get_range_values_per_column(df)
        out["range"] = {"max": {k: float(v) for k, v in mx.items()}, "min": {k: float(v) for k, v in mn.items()},
                        "median": {k: float(v) for k, v in med.items()}}
        out["impute"] = ref.impute(impute_frame()).values.tolist()
        out["impute_zero"] = ref.impute_dataframe_zero(impute_frame()).values.tolist()
    path = os.path.join(HERE, "ref_impute.json")
    json.dump(out, open(path, "w"))
    print("wrote", path)


if __name__ == "__main__":
    main()
