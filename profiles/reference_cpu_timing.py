#!/usr/bin/env python
"""CPU baseline of the REAL reference (build container only: /root/reference does not exist on the GPU box).

Times `tsfresh.extract_features(..., distributor=MultiprocessingDistributor(n_workers))` -- the path north_star names
(tsfresh/utilities/distribution.py:438, extraction.py:262-275) -- on synthetic float32 series x 1024 with
ComprehensiveFCParameters minus the five calculators whose third-party modules are missing in the main interpreter
(pywt / statsmodels: cwt_coefficients, agg_autocorrelation, partial_autocorrelation, augmented_dickey_fuller,
ar_coefficient -> 70 of 75 calculators, 696 of 783 columns; the number is therefore OPTIMISTIC FOR THE CPU), and the
oracle ("port", all 75 calculators) with the same protocol on the same cores, so that bench.py's cpu_baseline (the port,
timed on the GPU box's host) can be read against the reference:

    >= 32 series per worker, pool started outside the clock, 3 repeats, median;  OMP/MKL/OPENBLAS threads = 1
    (docs/text/tsfresh_on_a_cluster.rst:216-231);  n_jobs = all cores and the reference default cores // 2 (defaults.py:7).

    python profiles/reference_cpu_timing.py > profiles/r02_reference_cpu.json
"""
import json
import os
import statistics
import sys
import time
import types
import warnings

for v in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
    os.environ[v] = "1"
import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
L = 1024
PER_WORKER = 32
NEED_THIRD_PARTY = ("cwt_coefficients", "agg_autocorrelation", "partial_autocorrelation", "augmented_dickey_fuller",
                    "ar_coefficient")


def load_reference():
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
    import tsfresh  # noqa: F401
    from tsfresh.feature_extraction import extract_features, settings
    from tsfresh.utilities.distribution import MultiprocessingDistributor
    return extract_features, settings, MultiprocessingDistributor


def frame(n, seed):
    x = np.random.default_rng(seed).standard_normal((n, L), dtype=np.float32)
    return x, pd.DataFrame({"id": np.repeat(np.arange(n), L), "time": np.tile(np.arange(L), n), "value": x.reshape(-1)})


def time_reference(n_jobs, repeats=3):
    extract_features, settings, MultiprocessingDistributor = load_reference()
    params = settings.ComprehensiveFCParameters()
    for k in NEED_THIRD_PARTY:
        del params[k]
    n = n_jobs * PER_WORKER
    walls = []
    for r in range(repeats):
        _, df = frame(n, 42 + r)
        dist = MultiprocessingDistributor(n_workers=n_jobs, disable_progressbar=True, progressbar_title="x",
                                          show_warnings=False)
        dist.pool.map(abs, range(4 * n_jobs))  # workers are up before the clock starts
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            t0 = time.perf_counter()
            out = extract_features(df, column_id="id", column_sort="time", default_fc_parameters=params,
                                   distributor=dist, disable_progressbar=True)
            walls.append(time.perf_counter() - t0)
        assert out.shape[0] == n, out.shape
    w = statistics.median(walls)
    return {"n_jobs": n_jobs, "series": n, "n_cols": int(out.shape[1]), "walls_s": walls, "median_s": w,
            "series_per_sec": n / w, "series_per_sec_per_core": n / w / n_jobs}


def _port_worker(args):
    values, offsets = args
    from oracle.extract import oracle_matrix
    from tsfresh_amd.feature_extraction.settings import ComprehensiveFCParameters
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return oracle_matrix(values, offsets, ComprehensiveFCParameters())[1].shape


def _warm(_):
    import oracle.extract  # noqa: F401
    import tsfresh_amd.feature_extraction.settings  # noqa: F401
    return 0


def time_port(n_jobs, repeats=3):
    import multiprocessing as mp
    walls = []
    with mp.get_context("spawn").Pool(n_jobs) as pool:
        pool.map(_warm, range(4 * n_jobs))
        for r in range(repeats):
            jobs = []
            for w in range(n_jobs):
                x, _ = frame(PER_WORKER, 1000 * r + w)
                jobs.append((x.astype(np.float64).reshape(-1), np.arange(PER_WORKER + 1, dtype=np.int64) * L))
            t0 = time.perf_counter()
            pool.map(_port_worker, jobs)
            walls.append(time.perf_counter() - t0)
    n = n_jobs * PER_WORKER
    w = statistics.median(walls)
    return {"n_jobs": n_jobs, "series": n, "n_cols": 783, "walls_s": walls, "median_s": w, "series_per_sec": n / w,
            "series_per_sec_per_core": n / w / n_jobs}


def main():
    cores = os.cpu_count() or 1
    doc = {"host": {"cpu_count": cores, "model": open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t")},
           "workload": "float32 i.i.d. N(0,1) series x %d, %d per worker" % (L, PER_WORKER),
           "reference": {"what": "tsfresh.extract_features + MultiprocessingDistributor, 70 of 75 calculators (696 columns)",
                         "all_cores": time_reference(cores), "default_half_cores": time_reference(max(1, cores // 2))},
           "port": {"what": "oracle/ (numpy restatement), all 75 calculators (783 columns), same protocol",
                    "all_cores": time_port(cores)}}
    doc["port_over_reference_per_core"] = (doc["port"]["all_cores"]["series_per_sec_per_core"] /
                                           doc["reference"]["all_cores"]["series_per_sec_per_core"])
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
