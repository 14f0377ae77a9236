"""Registry of the feature calculators that have a native (HIP) implementation.

The reference discovers calculators by introspecting its module
(tsfresh/feature_extraction/feature_calculators.py: every function carrying an ``fctype`` attribute set by
``set_property`` :222-235).  Here the same facts are a table: name, fctype ("simple" | "combiner"), the flags
``minimal`` / ``high_comp_cost`` / ``input`` / ``index_type``, and how one parameter dict of the FCParameters
mapping becomes (a) the column-name suffix the reference would generate and (b) the numeric slots p[0..3] of a
``tsfa_feature_spec`` (include/tsfresh_amd.h, tsfresh_amd/csrc/tsfa_specs.h).

The ORDER of this table is the definition order of the functions in the reference module, because
``ComprehensiveFCParameters`` lists the parameter-less calculators in that order (settings.py:157-163).
"""
from collections import OrderedDict

import numpy as np

from tsfresh_amd.utilities.string_manipulation import convert_to_output_format

ATTR_LINREG = {"pvalue": 0, "rvalue": 1, "intercept": 2, "slope": 3, "stderr": 4}
AGG = {"max": 0, "min": 1, "mean": 2, "var": 3, "median": 4}
ATTR_FFT = {"real": 0, "imag": 1, "abs": 2, "angle": 3}
AGG_FFT = {"centroid": 0, "variance": 1, "skew": 2, "kurtosis": 3}
ATTR_ADF = {"teststat": 0, "pvalue": 1, "usedlag": 2}


class UnsupportedFeature(NotImplementedError):
    """A calculator / parameter combination that has no native kernel (there is no CPU fallback)."""


def _code(table, value, what):
    try:
        return float(table[value])
    except KeyError:
        raise UnsupportedFeature("{}: unsupported value {!r}".format(what, value)) from None


class Calc:
    """One calculator: `encode(param) -> (p0, p1, p2, p3)`, `key(param) -> column-name suffix`."""

    def __init__(self, name, fctype, n_args=2, minimal=False, high_comp_cost=False, input=None, index_type=None,
                 encode=None, key=None, native=True):
        self.name = name
        self.fctype = fctype
        self.n_args = n_args  # 1: f(x) only (listed with parameters None)
        self.minimal = minimal
        self.high_comp_cost = high_comp_cost
        self.input = input
        self.index_type = index_type
        self._encode = encode
        self._key = key
        self.native = native

    def encode(self, param):
        if self._encode is None:
            return (0.0, 0.0, 0.0, 0.0)
        p = tuple(float(v) for v in self._encode(param))
        return p + (0.0,) * (4 - len(p))

    def key(self, param):
        if param is None:
            return ""
        if self._key is not None:
            return self._key(param)
        return convert_to_output_format(param)


def _simple0(name, **kw):
    return Calc(name, "simple", n_args=1, **kw)


ADF_AUTOLAG = {"aic": 0.0, "bic": 1.0, "t-stat": 2.0}   # tsfa_specs.h TSFA_AUTOLAG_*; None (the regression at maxlag): 3


def _adf_encode(p):
    """fc.py:499-545 hands `autolag` to statsmodels.adfuller (case-insensitive "AIC" / "BIC" / "t-stat", or None).
    Every settings object of the reference uses "AIC"; the other three selections are served by the same kernels
    (fam_ar.h, fam_ar_dd.h; one autolag value per plan) and pinned on statsmodels' own output on the device
    (tests/test_adf_autolag.py, first device run: profiles/r05_a_pytest_new.log)."""
    autolag = p.get("autolag", "AIC")
    if autolag is not None and not isinstance(autolag, str):
        # statsmodels.tools.validation.string_like raises TypeError, which fc.py:521-527 does not catch: it propagates
        raise TypeError("autolag must be a string or None")
    mode = 3.0 if autolag is None else ADF_AUTOLAG.get(autolag.lower())
    if mode is None:
        # any other string (the STRING "None" that from_columns makes of autolag_"None" included): statsmodels raises
        # ValueError, fc.py:523 turns that into (nan, nan, nan) -- a NaN column under the name the caller gave it
        # (attr code 3 = "always NaN": exempt from the one-autolag-value-per-plan rule, tsfa_validate_plan)
        return (3.0, 0.0)
    return (_code(ATTR_ADF, p["attr"], "augmented_dickey_fuller attr") if p["attr"] in ATTR_ADF else 3.0, mode)


def _cwt_encode(p):
    widths = tuple(p["widths"])
    if p["w"] not in widths:
        raise ValueError("{} is not in widths {}".format(p["w"], widths))  # reference: tuple.index raises
    return (p["w"], p["coeff"])


_CALCS = [
    # --- in the definition order of feature_calculators.py ---
    _simple0("variance_larger_than_standard_deviation"),
    Calc("ratio_beyond_r_sigma", "simple", encode=lambda p: (p["r"],)),
    Calc("large_standard_deviation", "simple", encode=lambda p: (p["r"],)),
    Calc("symmetry_looking", "combiner", encode=lambda p: (p["r"],), key=lambda p: "r_{}".format(p["r"])),
    _simple0("has_duplicate_max"),
    _simple0("has_duplicate_min"),
    _simple0("has_duplicate"),
    _simple0("sum_values", minimal=True),
    Calc("agg_autocorrelation", "combiner",
         encode=lambda p: (_code({"mean": 2, "median": 4, "var": 3}, p["f_agg"], "agg_autocorrelation f_agg"), p["maxlag"]),
         key=lambda p: 'f_agg_"{}"__maxlag_{}'.format(p["f_agg"], p["maxlag"])),
    Calc("partial_autocorrelation", "combiner", encode=lambda p: (p["lag"],), key=lambda p: "lag_{}".format(p["lag"])),
    Calc("augmented_dickey_fuller", "combiner", encode=_adf_encode,
         key=lambda p: 'attr_"{}"__autolag_"{}"'.format(p["attr"], p.get("autolag", "AIC"))),
    _simple0("abs_energy"),
    Calc("cid_ce", "simple", encode=lambda p: (1.0 if p["normalize"] else 0.0,)),
    _simple0("mean_abs_change"),
    _simple0("mean_change"),
    _simple0("mean_second_derivative_central"),
    _simple0("median", minimal=True),
    _simple0("mean", minimal=True),
    _simple0("length", minimal=True),
    _simple0("standard_deviation", minimal=True),
    _simple0("variation_coefficient"),
    _simple0("variance", minimal=True),
    _simple0("skewness", input="pd.Series"),
    _simple0("kurtosis", input="pd.Series"),
    _simple0("root_mean_square", minimal=True),
    _simple0("absolute_sum_of_changes"),
    _simple0("longest_strike_below_mean"),
    _simple0("longest_strike_above_mean"),
    _simple0("count_above_mean"),
    _simple0("count_below_mean"),
    _simple0("last_location_of_maximum"),
    _simple0("first_location_of_maximum"),
    _simple0("last_location_of_minimum"),
    _simple0("first_location_of_minimum"),
    _simple0("percentage_of_reoccurring_values_to_all_values"),
    _simple0("percentage_of_reoccurring_datapoints_to_all_datapoints", input="pd.Series"),
    _simple0("sum_of_reoccurring_values"),
    _simple0("sum_of_reoccurring_data_points"),
    _simple0("ratio_value_number_to_time_series_length"),
    Calc("fft_coefficient", "combiner",
         encode=lambda p: (p["coeff"], _code(ATTR_FFT, p["attr"], "fft_coefficient attr")),
         key=lambda p: 'attr_"{}"__coeff_{}'.format(p["attr"], p["coeff"])),
    Calc("fft_aggregated", "combiner", encode=lambda p: (_code(AGG_FFT, p["aggtype"], "fft_aggregated aggtype"),),
         key=lambda p: 'aggtype_"{}"'.format(p["aggtype"])),
    Calc("number_peaks", "simple", encode=lambda p: (p["n"],)),
    Calc("index_mass_quantile", "combiner", encode=lambda p: (p["q"],), key=lambda p: "q_{}".format(p["q"])),
    Calc("number_cwt_peaks", "simple", encode=lambda p: (p["n"],)),
    Calc("linear_trend", "combiner", encode=lambda p: (_code(ATTR_LINREG, p["attr"], "linear_trend attr"),),
         key=lambda p: 'attr_"{}"'.format(p["attr"])),
    Calc("cwt_coefficients", "combiner", encode=_cwt_encode,
         key=lambda p: "coeff_{}__w_{}__widths_{}".format(p["coeff"], p["w"], tuple(p["widths"]))),
    Calc("spkt_welch_density", "combiner", encode=lambda p: (p["coeff"],), key=lambda p: "coeff_{}".format(p["coeff"])),
    Calc("ar_coefficient", "combiner", encode=lambda p: (p["coeff"], p["k"]),
         key=lambda p: "coeff_{}__k_{}".format(p["coeff"], p["k"])),
    Calc("change_quantiles", "simple",
         encode=lambda p: (p["ql"], p["qh"], 1.0 if p["isabs"] else 0.0,
                           _code({"mean": 2, "var": 3}, p["f_agg"], "change_quantiles f_agg"))),
    Calc("time_reversal_asymmetry_statistic", "simple", encode=lambda p: (p["lag"],)),
    Calc("c3", "simple", encode=lambda p: (p["lag"],)),
    Calc("mean_n_absolute_max", "simple", encode=lambda p: (p["number_of_maxima"],)),
    Calc("binned_entropy", "simple", encode=lambda p: (p["max_bins"],)),
    _simple0("sample_entropy", high_comp_cost=True),
    Calc("approximate_entropy", "simple", high_comp_cost=True, encode=lambda p: (p["m"], p["r"])),
    Calc("fourier_entropy", "simple", encode=lambda p: (p["bins"],)),
    Calc("lempel_ziv_complexity", "simple", encode=lambda p: (p["bins"],)),
    Calc("permutation_entropy", "simple", encode=lambda p: (p["tau"], p["dimension"])),
    Calc("autocorrelation", "simple", encode=lambda p: (p["lag"],)),
    Calc("quantile", "simple", encode=lambda p: (p["q"],)),
    Calc("number_crossing_m", "simple", encode=lambda p: (p["m"],)),
    _simple0("maximum", minimal=True),
    _simple0("absolute_maximum", minimal=True),
    _simple0("minimum", minimal=True),
    Calc("value_count", "simple", encode=lambda p: (p["value"],)),
    Calc("range_count", "simple", encode=lambda p: (p["min"], p["max"])),
    Calc("friedrich_coefficients", "combiner", encode=lambda p: (p["coeff"], p["m"], p["r"]),
         key=lambda p: "coeff_{}__m_{}__r_{}".format(p["coeff"], p["m"], p["r"])),
    Calc("max_langevin_fixed_point", "simple", encode=lambda p: (p["m"], p["r"])),
    Calc("agg_linear_trend", "combiner",
         encode=lambda p: (_code(ATTR_LINREG, p["attr"], "agg_linear_trend attr"), p["chunk_len"],
                           _code({"max": 0, "min": 1, "mean": 2, "var": 3}, p["f_agg"], "agg_linear_trend f_agg")),
         key=lambda p: 'attr_"{}"__chunk_len_{}__f_agg_"{}"'.format(p["attr"], p["chunk_len"], p["f_agg"])),
    Calc("energy_ratio_by_chunks", "combiner", encode=lambda p: (p["num_segments"], p["segment_focus"]),
         key=lambda p: "num_segments_{}__segment_focus_{}".format(p["num_segments"], p["segment_focus"])),
    # needs a DatetimeIndex; on any other index the reference skips it with a warning (extraction.py:349-358)
    Calc("linear_trend_timewise", "combiner", input="pd.Series", index_type="DatetimeIndex",
         encode=lambda p: (_code(ATTR_LINREG, p["attr"], "linear_trend_timewise attr"),),
         key=lambda p: 'attr_"{}"'.format(p["attr"])),
    Calc("count_above", "simple", encode=lambda p: (p["t"],)),
    Calc("count_below", "simple", encode=lambda p: (p["t"],)),
    _simple0("benford_correlation"),
    # feature_calculators.py:2385 needs the optional `matrixprofile` package; without it the reference drops the
    # calculator from its settings (settings.py:282-292).  It is registered only so that the name is known.
    Calc("matrix_profile", "combiner", native=False),
    Calc("query_similarity_count", "combiner", key=convert_to_output_format),
]

CALCULATORS = OrderedDict((c.name, c) for c in _CALCS)


def _query_similarity_encode(p):
    """(threshold, normalize, 0, len(Q)) + Q: the tail is the ARRAY parameter; `_native.Plan` moves it to the plan's float64
    pool and writes its offset to p[2] (include/tsfresh_amd.h: tsfa_plan_create_with_data).  fc.py:2475-2519:
    Q = np.asarray(query).astype(float); fewer than three samples (or query=None) -> np.nan for every series."""
    q = p.get("query", None)
    thr = float(p.get("threshold", 0.0))
    norm = 1.0 if p.get("normalize", True) else 0.0
    if q is None:
        return (thr, norm, 0.0, 0.0)
    qa = np.asarray(q).astype(float).ravel()
    return (thr, norm, 0.0, float(qa.size)) + tuple(float(v) for v in qa)


CALCULATORS["query_similarity_count"]._encode = _query_similarity_encode
