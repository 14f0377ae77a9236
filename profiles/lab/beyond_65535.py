#!/usr/bin/env python
"""Which calculator families survive series beyond the 65 535-sample cap if the check is simply lifted
import os, sys, time, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
def main():
    from engines import hip_engine, oracle_engine_parallel
    from parity import compare
    from tsfresh_amd.feature_extraction.settings import EfficientFCParameters
    from tsfresh_amd.feature_extraction.registry import CALCULATORS
    full = EfficientFCParameters()
    rng = np.random.default_rng(1)
    lens = [int(v) for v in os.environ.get("B65_LENS", "70000,100001").split(",")]
    series = [rng.standard_normal(lens[0]).astype(np.float32), np.cumsum(rng.standard_normal(lens[1])).astype(np.float32)]
    values = np.concatenate(series); offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    groups = {
     "spectral": ["fft_coefficient", "fft_aggregated", "spkt_welch_density", "fourier_entropy"],
     "ar": ["agg_autocorrelation", "partial_autocorrelation", "ar_coefficient", "augmented_dickey_fuller"],
     "cwt_peaks": ["number_cwt_peaks"], "cwt_gemm": ["cwt_coefficients"], "seq": ["lempel_ziv_complexity"],
     "sort": ["median", "quantile", "change_quantiles", "mean_n_absolute_max", "percentage_of_reoccurring_values_to_all_values", "sum_of_reoccurring_values", "friedrich_coefficients", "max_langevin_fixed_point", "symmetry_looking", "ratio_value_number_to_time_series_length"],
     "perm": ["permutation_entropy"],
     "trend": ["index_mass_quantile", "linear_trend", "agg_linear_trend"],
    }
    used = set(sum(groups.values(), []))
    groups["basic"] = [k for k in full if k not in used and k not in ("sample_entropy", "approximate_entropy")]
    only = sys.argv[1:]
    for name, keys in groups.items():
        if only and name not in only:
            continue
        params = {k: full[k] for k in keys if k in full}
        t0 = time.time()
        try:
            names, got = hip_engine(params, values, offsets)
            t1 = time.time()
            onames, want = oracle_engine_parallel(params, values.astype(np.float64), offsets)
            bad = compare(names, got, want, [s.astype(np.float64) for s in series])
            cols = sorted({b.split()[2].split("__")[1] for b in bad})
            print("%-10s %4d columns  gpu %.2fs  mismatches %d %s %s" % (name, len(names), t1 - t0, len(bad), cols, bad[:2]), flush=True)
        except Exception as e:  # noqa: BLE001
            print("%-10s FAILED: %s" % (name, repr(e)[:300]), flush=True)


if __name__ == "__main__":
    main()
