#!/usr/bin/env python
"""extract -> impute -> select (tsfresh/convenience/relevant_extraction.py:18) on one MI355X: the default chain (the feature
matrix crosses PCIe after every step) against device_resident=True (it stays in HBM; only the selected columns return).
    python profiles/chain_timing.py [n_ids] [length]"""
import json
import os
import sys
import time

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tsfresh_amd import ComprehensiveFCParameters, extract_relevant_features  # noqa: E402


def main():
    n_ids = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    rng = np.random.default_rng(0)
    label = rng.integers(0, 2, n_ids)
    x = rng.standard_normal((n_ids, L)).astype(np.float32) + 0.3 * label[:, None].astype(np.float32)
    df = pd.DataFrame({"id": np.repeat(np.arange(n_ids), L), "t": np.tile(np.arange(L), n_ids), "x": x.ravel()})
    y = pd.Series(label, index=np.arange(n_ids))
    kw = dict(column_id="id", column_sort="t", default_fc_parameters=ComprehensiveFCParameters())
    out = {"n_ids": n_ids, "length": L}
    for name, flag in (("host_chain", False), ("device_resident", True)):
        extract_relevant_features(df, y, device_resident=flag, **kw)  # warm: plans, pinned pools
        t = []
        for _ in range(3):
            t0 = time.perf_counter()
            res = extract_relevant_features(df, y, device_resident=flag, **kw)
            t.append(time.perf_counter() - t0)
        out[name] = {"seconds": sorted(t)[1], "selected_columns": int(res.shape[1])}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
