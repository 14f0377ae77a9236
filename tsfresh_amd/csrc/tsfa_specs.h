// Calculator registry shared by the HIP kernels, the C-ABI and (through tsfa_calc_id) the Python host.
// One row per calculator of tsfresh/feature_extraction/feature_calculators.py that has a native kernel.
//   X(enum, "tsfresh name", family)
// Parameter slots p[0..3] of tsfa_feature_spec, per calculator (settings.py:165-280 gives the values):
//   large_standard_deviation, symmetry_looking, ratio_beyond_r_sigma : p0 = r
//   cid_ce                          : p0 = normalize (0/1)
//   count_above, count_below        : p0 = t
//   value_count                     : p0 = value
//   range_count                     : p0 = min, p1 = max
//   number_crossing_m               : p0 = m
//   number_peaks, number_cwt_peaks  : p0 = n
//   index_mass_quantile, quantile   : p0 = q
//   energy_ratio_by_chunks          : p0 = num_segments, p1 = segment_focus
//   c3, time_reversal_asymmetry_statistic, autocorrelation : p0 = lag
//   binned_entropy                  : p0 = max_bins
//   linear_trend                    : p0 = attr (TSFA_ATTR_*)
//   agg_linear_trend                : p0 = attr, p1 = chunk_len, p2 = f_agg (TSFA_AGG_*)
//   mean_n_absolute_max             : p0 = number_of_maxima
//   change_quantiles                : p0 = ql, p1 = qh, p2 = isabs (0/1), p3 = f_agg (mean/var)
//   permutation_entropy             : p0 = tau, p1 = dimension
//   friedrich_coefficients          : p0 = coeff, p1 = m, p2 = r
//   max_langevin_fixed_point        : p0 = m, p1 = r
//   fft_coefficient                 : p0 = coeff, p1 = attr (TSFA_FFT_*)
//   fft_aggregated                  : p0 = aggtype (TSFA_FFTAGG_*)
//   spkt_welch_density              : p0 = coeff
//   fourier_entropy, lempel_ziv_complexity : p0 = bins
//   agg_autocorrelation             : p0 = f_agg (mean/median/var), p1 = maxlag
//   partial_autocorrelation         : p0 = lag
//   ar_coefficient                  : p0 = coeff, p1 = k
//   augmented_dickey_fuller         : p0 = attr (TSFA_ADF_*), p1 = lag selection (TSFA_AUTOLAG_*, one value per plan)
//   approximate_entropy             : p0 = m, p1 = r
//   cwt_coefficients                : p0 = w (the width), p1 = coeff
#ifndef TSFA_SPECS_H
#define TSFA_SPECS_H

#include <stdint.h>

enum tsfa_family {
    TSFA_FAM_BASIC = 0,   // single-pass reductions / scans over the LDS-resident series
    TSFA_FAM_SORT = 1,    // in-LDS bitonic sort, then scans
    TSFA_FAM_SPECTRAL = 2,// FFT / Welch
    TSFA_FAM_AR = 3,      // autocovariance-based small dense linear algebra
    TSFA_FAM_ENTROPY = 4, // O(L^2) template-pair sweep
    TSFA_FAM_CWT = 5,     // Ricker/mexh contractions (MFMA) + ridge lines
    TSFA_FAM_SEQ = 6,     // inherently sequential parses
    TSFA_FAM_TREND = 7,   // cumulative sums / chunk aggregates / regressions over the series (float64 work array)
    TSFA_N_FAMILIES = 8
};

#define TSFA_CALC_LIST(X)                                                              \
    X(SUM_VALUES, "sum_values", TSFA_FAM_BASIC)                                         \
    X(MEAN, "mean", TSFA_FAM_BASIC)                                                     \
    X(LENGTH, "length", TSFA_FAM_BASIC)                                                 \
    X(STANDARD_DEVIATION, "standard_deviation", TSFA_FAM_BASIC)                         \
    X(VARIANCE, "variance", TSFA_FAM_BASIC)                                             \
    X(ROOT_MEAN_SQUARE, "root_mean_square", TSFA_FAM_BASIC)                             \
    X(MAXIMUM, "maximum", TSFA_FAM_BASIC)                                               \
    X(ABSOLUTE_MAXIMUM, "absolute_maximum", TSFA_FAM_BASIC)                             \
    X(MINIMUM, "minimum", TSFA_FAM_BASIC)                                               \
    X(ABS_ENERGY, "abs_energy", TSFA_FAM_BASIC)                                         \
    X(VARIATION_COEFFICIENT, "variation_coefficient", TSFA_FAM_BASIC)                   \
    X(VAR_GT_STD, "variance_larger_than_standard_deviation", TSFA_FAM_BASIC)            \
    X(LARGE_STD, "large_standard_deviation", TSFA_FAM_BASIC)                            \
    X(RATIO_BEYOND_R_SIGMA, "ratio_beyond_r_sigma", TSFA_FAM_BASIC)                     \
    X(SKEWNESS, "skewness", TSFA_FAM_BASIC)                                             \
    X(KURTOSIS, "kurtosis", TSFA_FAM_BASIC)                                             \
    X(MEAN_ABS_CHANGE, "mean_abs_change", TSFA_FAM_BASIC)                               \
    X(MEAN_CHANGE, "mean_change", TSFA_FAM_BASIC)                                       \
    X(MEAN_SECOND_DERIVATIVE_CENTRAL, "mean_second_derivative_central", TSFA_FAM_BASIC) \
    X(ABSOLUTE_SUM_OF_CHANGES, "absolute_sum_of_changes", TSFA_FAM_BASIC)               \
    X(CID_CE, "cid_ce", TSFA_FAM_BASIC)                                                 \
    X(COUNT_ABOVE_MEAN, "count_above_mean", TSFA_FAM_BASIC)                             \
    X(COUNT_BELOW_MEAN, "count_below_mean", TSFA_FAM_BASIC)                             \
    X(COUNT_ABOVE, "count_above", TSFA_FAM_BASIC)                                       \
    X(COUNT_BELOW, "count_below", TSFA_FAM_BASIC)                                       \
    X(VALUE_COUNT, "value_count", TSFA_FAM_BASIC)                                       \
    X(RANGE_COUNT, "range_count", TSFA_FAM_BASIC)                                       \
    X(NUMBER_CROSSING_M, "number_crossing_m", TSFA_FAM_BASIC)                           \
    X(FIRST_LOCATION_OF_MAXIMUM, "first_location_of_maximum", TSFA_FAM_BASIC)           \
    X(LAST_LOCATION_OF_MAXIMUM, "last_location_of_maximum", TSFA_FAM_BASIC)             \
    X(FIRST_LOCATION_OF_MINIMUM, "first_location_of_minimum", TSFA_FAM_BASIC)           \
    X(LAST_LOCATION_OF_MINIMUM, "last_location_of_minimum", TSFA_FAM_BASIC)             \
    X(HAS_DUPLICATE_MAX, "has_duplicate_max", TSFA_FAM_BASIC)                           \
    X(HAS_DUPLICATE_MIN, "has_duplicate_min", TSFA_FAM_BASIC)                           \
    X(LONGEST_STRIKE_ABOVE_MEAN, "longest_strike_above_mean", TSFA_FAM_BASIC)           \
    X(LONGEST_STRIKE_BELOW_MEAN, "longest_strike_below_mean", TSFA_FAM_BASIC)           \
    X(NUMBER_PEAKS, "number_peaks", TSFA_FAM_BASIC)                                     \
    X(INDEX_MASS_QUANTILE, "index_mass_quantile", TSFA_FAM_TREND)                       \
    X(ENERGY_RATIO_BY_CHUNKS, "energy_ratio_by_chunks", TSFA_FAM_BASIC)                 \
    X(C3, "c3", TSFA_FAM_BASIC)                                                         \
    X(TIME_REVERSAL_ASYMMETRY_STATISTIC, "time_reversal_asymmetry_statistic", TSFA_FAM_BASIC) \
    X(AUTOCORRELATION, "autocorrelation", TSFA_FAM_BASIC)                               \
    X(BINNED_ENTROPY, "binned_entropy", TSFA_FAM_BASIC)                                 \
    X(BENFORD_CORRELATION, "benford_correlation", TSFA_FAM_BASIC)                       \
    X(LINEAR_TREND, "linear_trend", TSFA_FAM_TREND)                                     \
    X(AGG_LINEAR_TREND, "agg_linear_trend", TSFA_FAM_TREND)                             \
    X(QUERY_SIMILARITY_COUNT, "query_similarity_count", TSFA_FAM_BASIC)                 \
    X(MEDIAN, "median", TSFA_FAM_SORT)                                                  \
    X(QUANTILE, "quantile", TSFA_FAM_SORT)                                              \
    X(SYMMETRY_LOOKING, "symmetry_looking", TSFA_FAM_SORT)                              \
    X(MEAN_N_ABSOLUTE_MAX, "mean_n_absolute_max", TSFA_FAM_SORT)                        \
    X(CHANGE_QUANTILES, "change_quantiles", TSFA_FAM_SORT)                              \
    X(HAS_DUPLICATE, "has_duplicate", TSFA_FAM_SORT)                                    \
    X(RATIO_VALUE_NUMBER, "ratio_value_number_to_time_series_length", TSFA_FAM_SORT)    \
    X(PCT_REOCC_VALUES, "percentage_of_reoccurring_values_to_all_values", TSFA_FAM_SORT) \
    X(PCT_REOCC_DATAPOINTS, "percentage_of_reoccurring_datapoints_to_all_datapoints", TSFA_FAM_SORT) \
    X(SUM_REOCC_VALUES, "sum_of_reoccurring_values", TSFA_FAM_SORT)                     \
    X(SUM_REOCC_DATA_POINTS, "sum_of_reoccurring_data_points", TSFA_FAM_SORT)           \
    X(PERMUTATION_ENTROPY, "permutation_entropy", TSFA_FAM_SORT)                        \
    X(FRIEDRICH_COEFFICIENTS, "friedrich_coefficients", TSFA_FAM_SORT)                  \
    X(MAX_LANGEVIN_FIXED_POINT, "max_langevin_fixed_point", TSFA_FAM_SORT)              \
    X(FFT_COEFFICIENT, "fft_coefficient", TSFA_FAM_SPECTRAL)                            \
    X(FFT_AGGREGATED, "fft_aggregated", TSFA_FAM_SPECTRAL)                              \
    X(SPKT_WELCH_DENSITY, "spkt_welch_density", TSFA_FAM_SPECTRAL)                      \
    X(FOURIER_ENTROPY, "fourier_entropy", TSFA_FAM_SPECTRAL)                            \
    X(AGG_AUTOCORRELATION, "agg_autocorrelation", TSFA_FAM_AR)                          \
    X(PARTIAL_AUTOCORRELATION, "partial_autocorrelation", TSFA_FAM_AR)                  \
    X(AR_COEFFICIENT, "ar_coefficient", TSFA_FAM_AR)                                    \
    X(AUGMENTED_DICKEY_FULLER, "augmented_dickey_fuller", TSFA_FAM_AR)                  \
    X(SAMPLE_ENTROPY, "sample_entropy", TSFA_FAM_ENTROPY)                               \
    X(APPROXIMATE_ENTROPY, "approximate_entropy", TSFA_FAM_ENTROPY)                     \
    X(CWT_COEFFICIENTS, "cwt_coefficients", TSFA_FAM_CWT)                               \
    X(NUMBER_CWT_PEAKS, "number_cwt_peaks", TSFA_FAM_CWT)                               \
    X(LEMPEL_ZIV_COMPLEXITY, "lempel_ziv_complexity", TSFA_FAM_SEQ)                     \
    X(LINEAR_TREND_TIMEWISE, "linear_trend_timewise", TSFA_FAM_TREND)

enum tsfa_calc {
#define X(id, name, fam) TSFA_C_##id,
    TSFA_CALC_LIST(X)
#undef X
    TSFA_N_CALCS
};

// string-valued parameters as codes
enum { TSFA_ATTR_PVALUE = 0, TSFA_ATTR_RVALUE = 1, TSFA_ATTR_INTERCEPT = 2, TSFA_ATTR_SLOPE = 3, TSFA_ATTR_STDERR = 4 };
enum { TSFA_AGG_MAX = 0, TSFA_AGG_MIN = 1, TSFA_AGG_MEAN = 2, TSFA_AGG_VAR = 3, TSFA_AGG_MEDIAN = 4 };
enum { TSFA_FFT_REAL = 0, TSFA_FFT_IMAG = 1, TSFA_FFT_ABS = 2, TSFA_FFT_ANGLE = 3 };
enum { TSFA_FFTAGG_CENTROID = 0, TSFA_FFTAGG_VARIANCE = 1, TSFA_FFTAGG_SKEW = 2, TSFA_FFTAGG_KURTOSIS = 3 };
enum { TSFA_ADF_TESTSTAT = 0, TSFA_ADF_PVALUE = 1, TSFA_ADF_USEDLAG = 2 };

// agg_linear_trend (fc.py:2171): the distinct (chunk_len, f_agg) regressions of a plan, sorted by chunk_len, worked
// out on the host.  All of them are computed in one go by the first agg_linear_trend column (fam_basic.h); column
// specs carry their key's index in p[3].  nkeys == 0: more than TSFA_ALT_MAXKEYS keys -> per-column evaluation.
#define TSFA_ALT_MAXKEYS 16
struct TsfaAltPlan {
    int nkeys;
    int want_p;  // some column asks for the p-value
    int cl[TSFA_ALT_MAXKEYS];
    int agg[TSFA_ALT_MAXKEYS];
    // index_mass_quantile (fc.py:1275): the distinct q of the plan; an indexed column has p[2] == 1 and its q's index
    // in p[1] (+ 128 on the column that evaluates them all).  nq == 0: more than TSFA_ALT_MAXKEYS -> per column.
    int nq;
    int small_w;  // TREND: every agg_linear_trend column is keyed and every index_mass_quantile column indexed (or absent):
                  // no calculator of the plan needs an n-double work array, the kernel's LDS holds a small scratch instead
    double q[TSFA_ALT_MAXKEYS];
};

// change_quantiles (fc.py:1511): the distinct valid corridors (ql < qh) of a plan.  All of them are evaluated by the
// first change_quantiles column, four corridors per sweep (fam_sort.h); an indexed column spec has p[1] == -2 and
// its corridor's index in p[0].  n == 0: more than TSFA_CQ_MAX corridors -> per-column evaluation.
#define TSFA_CQ_MAX 24
struct TsfaCqPlan {
    int n;
    int pad;
    double ql[TSFA_CQ_MAX];
    double qh[TSFA_CQ_MAX];
};

// what the plan's GENERAL specs (fam_general.h: parameter values beyond the tuned kernels' tables) ask for at most
// (tsfa_host_tables.h: tsfa_prepare_general); by-value kernel argument
struct TsfaGenPlan {
    int acf_maxlag;    // largest agg_autocorrelation maxlag (-1: none)
    int pacf_maxlag;   // largest partial_autocorrelation lag (-1: none)
    int fr_maxr;       // largest r of a friedrich_coefficients / max_langevin_fixed_point column (0: none)
    int fr_maxm;       // ... and the largest m
    int lz;            // lempel_ziv_complexity columns
    int cwt_maxw;      // largest n of a number_cwt_peaks column (0: none)
    int query;         // query_similarity_count columns that hold a query (their samples: the plan's float64 pool)
};

#define TSFA_AR_TABLE_K 31   // ar_coefficient orders of k_ar's float64 first pass (fam_ar.h: arres, 40 doubles); larger ones are fitted
                             // by the double-double second pass alone (fam_ar_dd.h: any order)

// device-side spec: one output column
struct TsfaSpec {
    int32_t calc;
    int32_t col;  // column in the output row
    double p[4];
};

// record of a Langevin fit that k_sort leaves to k_langevin_dd (fam_langevin_dd.h), in the plan's buffer:
// [series index, spec index, rows k, x_mean[rmax], y_mean[rmax]] as doubles
#define TSFA_PF_HDR 3
static inline int tsfa_pf_slot_doubles(int rmax) { return TSFA_PF_HDR + 2 * rmax; }

// offsets into the plan's constant tables (tsfa_host_tables.h: tsfa_build_consts)
// augmented_dickey_fuller, p[1]: the lag selection (statsmodels.adfuller autolag; one value per plan)
#define TSFA_AUTOLAG_AIC 0
#define TSFA_AUTOLAG_BIC 1
#define TSFA_AUTOLAG_TSTAT 2
#define TSFA_AUTOLAG_NONE 3   /* autolag=None: the regression at maxlag */
#define TSFA_AUTOLAG_TSTAT_STOP 1.6448536269514722   /* stats.norm.ppf(.95), stattools._autolag */
#define TSFA_CONSTS_HANN 0
// SPECTRAL: non-power-of-two lengths from here on take the chirp-z transform through HBM scratch (fam_spectral.h), shorter
// ones the O(n^2) Goertzel sweep.  Round 4's form paid from 4097 samples on (4096..8192: 16.7 -> 14.1 ms per 5 000 series);
// round 5's (fused cross pass, even lengths as n / 2 complex points) from ~1300: 10 000 series of 1025..2048 samples
// 1.94 ms with the crossover at 2049, 1.65 at 1793, 1.59 at 1281 (profiles/r05_j_spectral_nt.txt)
#define TSFA_BLUESTEIN_MIN 1281
// ... an EVEN length (transformed as n / 2 complex points: half the convolution length) from ~860: 20 000 series of 1000
// samples 1.56 -> 1.23 ms, 40 000 of 500 samples 1.15 -> 1.99 (profiles/r05_l_spectral_crossover.txt)
#define TSFA_BLUESTEIN_MIN_EVEN 897
// both crossovers travel as one kernel argument: odd | even << 16
#define TSFA_BLUESTEIN_PACK(odd, even) ((odd) | ((even) << 16))
#define TSFA_CONSTS_RICKER 256
#define TSFA_CONSTS_MAXW 16
#define TSFA_CONSTS_N (256 + 5 * TSFA_CONSTS_MAXW * (TSFA_CONSTS_MAXW + 1))

// per-series statistics record k_basic leaves for the other families of the same extraction (plan->stats_buf)
#define TSFA_STATS_MEAN 0   // np.mean(x): numpy's pairwise order
#define TSFA_STATS_VAR 1    // np.var(x)
#define TSFA_STATS_MIN 2
#define TSFA_STATS_MAX 3
#define TSFA_STATS_N 4

#endif
