#!/usr/bin/env python
"""Boundary timings beside bench.py's HBM-resident number:
  device   : tsfa_extract on device pointers (what bench.py times)
  host     : tsfa_extract on host buffers  (H2D of the values + kernels + D2H of the matrix: the PCIe-inclusive rate)
  frame    : extract_features(DataFrame) -> DataFrame (packer + host path + DataFrame construction)
    python profiles/e2e_timing.py [--n-series 20000] [--length 1024]
"""
import argparse
import json
import os
import sys
import time
import warnings

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-series", type=int, default=20000)
    ap.add_argument("--length", type=int, default=1024)
    args = ap.parse_args()
    import torch
    from tsfresh_amd import ComprehensiveFCParameters, _native, extract_features
    from tsfresh_amd.feature_extraction.plan import compile_fc_parameters
    n, L = args.n_series, args.length
    rng = np.random.default_rng(42)
    x = rng.standard_normal((n, L), dtype=np.float32)
    offsets = np.arange(n + 1, dtype=np.int64) * L
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fplan = compile_fc_parameters(ComprehensiveFCParameters())
    plan = _native.Plan(fplan.native_specs(_native.calc_id), device=0)
    res = {"n_series": n, "length": L, "n_cols": len(fplan)}
    # device-resident
    dev = torch.device("cuda", 0)
    dv = torch.from_numpy(x.reshape(-1)).to(dev)
    do = torch.from_numpy(offsets).to(dev)
    out = torch.empty((n, len(fplan)), device=dev, dtype=torch.float64)
    st = torch.cuda.current_stream(dev).cuda_stream
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plan.extract_device(dv.data_ptr(), _native.TSFA_F32, do.data_ptr(), n, out.data_ptr(), len(fplan), st)
        torch.cuda.synchronize()
        res["device_s"] = time.perf_counter() - t0
    for rep in range(2):
        t0 = time.perf_counter()
        m = plan.extract_host(x.reshape(-1), offsets)
        res["host_s"] = time.perf_counter() - t0
    assert np.array_equal(np.nan_to_num(m), np.nan_to_num(out.cpu().numpy()))
    df = pd.DataFrame({"id": np.repeat(np.arange(n), L), "time": np.tile(np.arange(L), n), "value": x.reshape(-1)})
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for rep in range(2):
            t0 = time.perf_counter()
            f = extract_features(df, column_id="id", column_sort="time", default_fc_parameters=ComprehensiveFCParameters())
            res["frame_s"] = time.perf_counter() - t0
    assert f.shape == (n, len(fplan))
    for k in ("device_s", "host_s", "frame_s"):
        res[k.replace("_s", "_series_per_s")] = n / res[k]
    res["bytes_in"] = int(x.nbytes)
    res["bytes_out"] = int(m.nbytes)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
