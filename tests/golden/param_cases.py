"""Parameters AWAY from the grids of ComprehensiveFCParameters (settings.py:133-296), for every calculator that takes
any: what a `from_columns` / hand-written settings dict sends down the same kernels.  gen_golden_main.py /
gen_golden_conda.py `--params sweep` run the real reference on them (ref_main_sweep.npz, ref_conda_sweep.npz);
tests/test_param_sweep.py compares the oracle, the emulation and the HIP path.  permutation_entropy has its own
fixture (perm_cases.py, ref_perm.json).  Plain Python 3.9 (the second interpreter imports this file)."""


def sweep_parameters():
    p = {}
    p["ratio_beyond_r_sigma"] = [{"r": r} for r in (0.3, 4.5, 12)]
    p["large_standard_deviation"] = [{"r": r} for r in (0.01, 0.33, 0.97)]
    p["symmetry_looking"] = [{"r": r} for r in (0.02, 0.33, 1.0)]
    p["cid_ce"] = [{"normalize": True}, {"normalize": False}]
    p["fft_coefficient"] = [{"coeff": c, "attr": a} for a in ("real", "imag", "abs", "angle") for c in (0, 7, 101, 300, 600)]
    p["fft_aggregated"] = [{"aggtype": s} for s in ("centroid", "variance", "skew", "kurtosis")]
    p["number_peaks"] = [{"n": n} for n in (2, 7, 25, 60)]
    p["index_mass_quantile"] = [{"q": q} for q in (0.05, 0.33, 0.999)]
    p["number_cwt_peaks"] = [{"n": n} for n in (2, 3, 8)]
    p["linear_trend"] = [{"attr": a} for a in ("pvalue", "rvalue", "intercept", "slope", "stderr")]
    p["spkt_welch_density"] = [{"coeff": c} for c in (0, 1, 3, 17, 128, 200)]
    p["change_quantiles"] = [{"ql": ql, "qh": qh, "isabs": b, "f_agg": f}
                             for (ql, qh) in ((0.1, 0.9), (0.33, 0.34), (0.0, 0.05), (0.5, 1.0))
                             for b in (False, True) for f in ("mean", "var")]
    p["time_reversal_asymmetry_statistic"] = [{"lag": lag} for lag in (4, 10, 50)]
    p["c3"] = [{"lag": lag} for lag in (4, 9, 40)]
    p["mean_n_absolute_max"] = [{"number_of_maxima": k} for k in (1, 3, 50, 2000)]
    p["binned_entropy"] = [{"max_bins": k} for k in (2, 7, 50, 256, 257, 1000)]   # (beyond 256 bins: round 6, the LDS counters are swept in rounds)
    p["approximate_entropy"] = [{"m": 2, "r": r} for r in (0.05, 0.45, 1.3)] + [{"m": 1, "r": 0.2}, {"m": 3, "r": 0.4}]
    p["fourier_entropy"] = [{"bins": k} for k in (4, 16, 128, 129, 300)]   # (beyond 128: round 6)
    p["lempel_ziv_complexity"] = [{"bins": k} for k in (4, 7, 33, 250)]
    p["autocorrelation"] = [{"lag": lag} for lag in (0, 11, 30, 500)]
    p["quantile"] = [{"q": q} for q in (0.0, 0.05, 0.5, 0.95, 1.0)]
    p["number_crossing_m"] = [{"m": m} for m in (-2.5, 0.3, 5)]
    p["value_count"] = [{"value": v} for v in (0.5, 2, -3.0)]
    p["range_count"] = [{"min": a, "max": b} for (a, b) in ((-0.5, 0.5), (0, 1e9), (1, -1))]
    p["friedrich_coefficients"] = ([{"coeff": c, "m": 1, "r": 10} for c in range(2)] + [{"coeff": c, "m": 2, "r": 5} for c in range(3)] +
                                   [{"coeff": c, "m": 3, "r": 64} for c in range(4)])
    p["max_langevin_fixed_point"] = [{"m": 2, "r": 10}, {"m": 3, "r": 50}]
    p["agg_linear_trend"] = [{"attr": a, "chunk_len": cl, "f_agg": f} for a in ("rvalue", "intercept", "slope", "stderr")
                             for cl in (3, 7, 100) for f in ("max", "min", "mean", "var")]
    p["energy_ratio_by_chunks"] = ([{"num_segments": 3, "segment_focus": i} for i in range(3)] +
                                   [{"num_segments": 7, "segment_focus": i} for i in (0, 3, 6)] +
                                   [{"num_segments": 16, "segment_focus": i} for i in (0, 9, 15)])
    p["count_above"] = [{"t": t} for t in (-1, 0.5, 3)]
    p["count_below"] = [{"t": t} for t in (-1, 0.5, 3)]
    # --- statsmodels / PyWavelets (second interpreter) ---
    p["agg_autocorrelation"] = [{"f_agg": f, "maxlag": m} for f in ("mean", "median", "var") for m in (5, 17, 60)]
    p["partial_autocorrelation"] = [{"lag": lag} for lag in (0, 11, 25, 40)]
    p["augmented_dickey_fuller"] = [{"attr": a} for a in ("teststat", "pvalue", "usedlag")]   # (autolag defaults to "AIC")
    p["ar_coefficient"] = [{"coeff": c, "k": k} for k in (3, 5, 12) for c in (0, 1, k, k + 1)]
    p["cwt_coefficients"] = ([{"widths": (1, 3, 7), "coeff": c, "w": w} for c in (0, 4, 19) for w in (1, 3, 7)] +
                             [{"widths": (2, 5, 10, 20), "coeff": c, "w": 20} for c in (1, 14)])
    return p


def beyond_parameters():
    """Values BEYOND the tables of the tuned kernels (round 6; tsfa_host_tables.h: tsfa_spec_beyond_tables): until then the
    library refused such a plan, now the calculators holding one are served by k_general (fam_general.h) -- ALL their columns,
    the in-table ones included -- and ar_coefficient orders above 31 by the double-double second pass (fam_ar_dd.h).
    `--params beyond` of both generators -> ref_main*_beyond.npz / ref_conda*_beyond.npz."""
    p = {}
    p["lempel_ziv_complexity"] = [{"bins": k} for k in (3, 256, 1000, 70000)]
    p["friedrich_coefficients"] = ([{"coeff": c, "m": 5, "r": 30} for c in range(6)] + [{"coeff": c, "m": 3, "r": 100} for c in range(4)] +
                                   [{"coeff": c, "m": 1, "r": 65} for c in range(2)] + [{"coeff": 1, "m": 2, "r": 10}])
    p["max_langevin_fixed_point"] = [{"m": 3, "r": 100}, {"m": 5, "r": 30}, {"m": 4, "r": 20}, {"m": 3, "r": 30}]
    p["number_cwt_peaks"] = [{"n": n} for n in (2, 17, 30)]
    # --- statsmodels (second interpreter) ---
    p["agg_autocorrelation"] = [{"f_agg": f, "maxlag": m} for f in ("mean", "median", "var") for m in (5, 61, 200, 5000)]
    p["partial_autocorrelation"] = [{"lag": lag} for lag in (0, 3, 41, 100, 700)]
    p["ar_coefficient"] = [{"coeff": c, "k": k} for k in (32, 50) for c in (0, 1, k, k + 1)] + [{"coeff": 2, "k": 5}]
    return p


THIRD_PARTY = ("cwt_coefficients", "agg_autocorrelation", "partial_autocorrelation", "augmented_dickey_fuller", "ar_coefficient")


def adf_autolag_parameters():
    """augmented_dickey_fuller with the lag selections ComprehensiveFCParameters does not use (fc.py:499-545 passes
    `autolag` through to statsmodels.adfuller): gen_golden_conda.py --params adf -> ref_conda_*_adf.npz."""
    return {"augmented_dickey_fuller": [{"attr": a, "autolag": al} for al in ("BIC", "t-stat", None)
                                        for a in ("teststat", "pvalue", "usedlag")]}
