/*
 * tsfresh_amd C-ABI  --  MI355X (gfx950) feature-extraction hot path.
 *
 * The reference (blue-yonder/tsfresh) is pure Python and has no FFI of its own.  Its plug points for
 * this path are:
 *   - tsfresh/feature_extraction/extraction.py:308  _do_extraction_on_chunk(chunk, fc_parameters)
 *       one (id, kind) series  x  the FCParameters dict  ->  [(id, "kind__calc__params", value), ...]
 *   - tsfresh/feature_extraction/extraction.py:193  _do_extraction  (adapter -> map_reduce -> pivot)
 *   - tsfresh/utilities/distribution.py:74          DistributorBaseClass.map_reduce
 * This header is what a ctypes binding for that path binds instead: the FCParameters dict is compiled
 * to a flat list of "feature specs" (one per output column), the list of series is handed over as one
 * ragged buffer (values + offsets), and the result is the dense [n_series x n_cols] float64 matrix
 * that tsfresh/feature_extraction/data.py:86 (pivot) would otherwise assemble from tuples.
 *
 * Plain C, plain pointers and sizes, no torch types.  All functions return 0 on success or a
 * negative tsfa_status; the message is available from tsfa_last_error() (thread-local).
 */
#ifndef TSFRESH_AMD_H
#define TSFRESH_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSFA_VERSION 100 /* 0.1.0 */

typedef enum tsfa_status {
    TSFA_OK = 0,
    TSFA_ERR_INVALID = -1,     /* bad argument */
    TSFA_ERR_UNSUPPORTED = -2, /* calculator / parameter not implemented natively */
    TSFA_ERR_NO_DEVICE = -3,   /* no HIP device: the library never falls back to the CPU */
    TSFA_ERR_HIP = -4,         /* HIP runtime error */
    TSFA_ERR_TOO_LONG = -5     /* a series of more than 33 554 431 samples (every calculator takes any length up to there; since
                                  round 6 also sample_entropy / approximate_entropy, which used to stop at 65 535) */
} tsfa_status;

/* element type of the ragged value buffer */
typedef enum tsfa_dtype { TSFA_F32 = 0, TSFA_F64 = 1, TSFA_I64 = 2, TSFA_I32 = 3 /* ids / sort keys only */ } tsfa_dtype;

/* where the caller's buffers live */
typedef enum tsfa_memspace { TSFA_HOST = 0, TSFA_DEVICE = 1 } tsfa_memspace;

/*
 * One output column: a calculator of tsfresh/feature_extraction/feature_calculators.py plus one
 * parameter combination of the FCParameters dict (tsfresh/feature_extraction/settings.py:165-280).
 * `calc` comes from tsfa_calc_id("<calculator name>"); the meaning of p[] per calculator is listed in
 * tsfresh_amd/csrc/tsfa_specs.h (string-valued parameters such as attr / f_agg are enum codes).
 */
typedef struct tsfa_feature_spec {
    int32_t calc;
    int32_t reserved;
    double p[4];
} tsfa_feature_spec;

typedef struct tsfa_plan tsfa_plan; /* opaque; owned by the library */

/* ---- library / device ---- */
int tsfa_version(void);
/* number of visible HIP devices (0 if none; never fails) */
int tsfa_device_count(void);
/* thread-local message of the last failing call on this thread ("" if none) */
const char *tsfa_last_error(void);

/* ---- calculator registry ---- */
/* id of a feature calculator by its tsfresh name (feature_calculators.py), or -1 if not native */
int tsfa_calc_id(const char *name);
/* inverse of tsfa_calc_id; NULL if out of range */
const char *tsfa_calc_name(int calc);
/* number of native calculators */
int tsfa_calc_count(void);

/* ---- plan: the compiled FCParameters dict for one kind ---- */
/*
 * specs[i] describes output column i.  `device` is the HIP device ordinal.  Replaces the Python loop
 * over fc_parameters.items() in extraction.py:339-378.  Returns TSFA_ERR_UNSUPPORTED (and names the
 * offender in tsfa_last_error) for a calculator/parameter combination that has no kernel.
 */
int tsfa_plan_create(const tsfa_feature_spec *specs, int32_t n_specs, int32_t device, tsfa_plan **out_plan);
/*
 * tsfa_plan_create for FCParameters whose parameter values include ARRAYS: `data` is a pool of n_data float64 (copied; the
 * caller keeps ownership) that such specs point into.  Today one calculator has an array parameter: query_similarity_count
 * (feature_calculators.py:2475-2519, {"query": Q, "threshold": thr, "normalize": norm}) --
 *     p[0] = thr, p[1] = norm (0 | 1), p[2] = offset of Q in the pool, p[3] = len(Q)  (0: query=None, the column is NaN).
 * TSFA_ERR_INVALID if a spec points outside the pool.  tsfa_plan_create is the case n_data = 0.
 */
int tsfa_plan_create_with_data(const tsfa_feature_spec *specs, int32_t n_specs, const double *data, int64_t n_data,
                               int32_t device, tsfa_plan **out_plan);
int32_t tsfa_plan_n_cols(const tsfa_plan *plan);
void tsfa_plan_destroy(tsfa_plan *plan);

/*
 * Extract all planned features for n_series ragged series.  Replaces distributor.map_reduce(
 * _do_extraction_on_chunk, ...) + pivot (extraction.py:294-304) for one kind.
 *
 *   values   concatenated samples, series s occupies [offsets[s], offsets[s+1]);  dtype per `dtype`
 *   offsets  n_series+1 int64, offsets[0] may be non-zero (a view into a larger buffer)
 *   out      row-major [n_series x ld_out] float64, ld_out >= n_cols; caller-allocated, caller-owned;
 *            NaN where the reference yields NaN
 *   space    TSFA_HOST: values/offsets/out are host pointers; the batch is staged through HBM in row chunks, copy-in
 *            of chunk c + 1 and copy-out of chunk c - 1 overlapping the kernels of chunk c (three streams);
 *            TSFA_DEVICE: all three are device pointers on the plan's device
 * A batch whose lengths span more than a factor of two is launched by length class (each class with the LDS carve and
 * workgroup size of ITS longest series), so one long series does not slow a batch of short ones down.
 *   stream   a hipStream_t (or NULL = the plan's own stream).  With TSFA_DEVICE and a non-NULL
 *            stream the call returns after enqueueing; otherwise it is synchronous.
 *
 * One plan may be used by one host thread at a time; plans on different devices are independent.
 */
int tsfa_extract(tsfa_plan *plan, const void *values, int32_t dtype, const int64_t *offsets,
                 int64_t n_series, double *out, int64_t ld_out, int32_t space, void *stream);

/*
 * tsfa_extract plus the per-sample abscissa that linear_trend_timewise regresses on
 * (feature_calculators.py:2274: hours since the series' first timestamp, taken from the DatetimeIndex
 * the reference hands the calculator as x.index, extraction.py:345-360).
 *
 *   times    float64, laid out exactly like `values` (same offsets, same memory space); NULL is allowed
 *            only for plans without linear_trend_timewise columns (TSFA_ERR_INVALID otherwise).
 */
int tsfa_extract_timed(tsfa_plan *plan, const void *values, int32_t dtype, const double *times,
                       const int64_t *offsets, int64_t n_series, double *out, int64_t ld_out,
                       int32_t space, void *stream);

/*
 * The window form: series s is the view values[starts[s] .. ends[s]) of one shared buffer; views may overlap.
 * This is what tsfresh/utilities/dataframe_functions.py:340-372 (_roll_out_time_series, called by
 * roll_time_series :377) produces by COPYING every window into a new DataFrame before extract_features is
 * called on it (the forecasting workflow, BASELINE configs[4]); here the windows stay views.
 * tsfa_extract / tsfa_extract_timed are the special case ends = starts + 1 (a ragged batch).
 *
 *   starts, ends   n_series int64 each (same memory space as values); 1 <= ends[s] - starts[s] <= 33554431 (TSFA_ERR_TOO_LONG)
 *   times          as in tsfa_extract_timed, indexed like values; the kernel regresses on
 *                  times[i] - times[starts[s]], i.e. hours since the window's own first stamp
 */
int tsfa_extract_windows(tsfa_plan *plan, const void *values, int32_t dtype, const double *times,
                         const int64_t *starts, const int64_t *ends, int64_t n_series, double *out,
                         int64_t ld_out, int32_t space, void *stream);

/*
 * Timing of the kernels of the last tsfa_extract on this plan, measured with HIP events on the
 * stream the kernels were launched on.  names[i] / ms[i] for i < returned count (<= cap).
 * Only recorded when tsfa_plan_set_profiling(plan, 1) was called before the extract.
 */
int tsfa_plan_set_profiling(tsfa_plan *plan, int32_t enable);
int32_t tsfa_plan_last_timings(const tsfa_plan *plan, const char **names, float *ms, int32_t cap);

/*
 * Launch options of one plan, set by the caller who owns it.  The library reads NO environment variable that can change a
 * result; what used to be debugging switches of the environment are named options here, each an alternative ROUTE to the same
 * numbers (the tests compare both ways) or a diagnostic pre-fill:
 *   "length_classes" 0|1   launch a batch whose lengths span more than 2x class by class (default 1)
 *   "stats_share"    0|1   k_basic's per-series statistics serve the other families (1)
 *   "perm_share"     0|1   the entropy kernel's sample order serves k_sort (1)
 *   "select"         0|1   median / quantile-only plans by selection instead of a sort (1)
 *   "fused_minimal"  0     MinimalFCParameters-shaped plans through the family kernels instead of k_stream
 *   "perm_fused"     0     permutation_entropy inside k_sort instead of k_perm
 *   "bluestein"      0|1   chirp-z transform for long spectra of non-power-of-two length (1); "bluestein_min" n: its crossover
 *   "gscratch_slots" n     cap of the chirp-z scratch slots (several launches per group)
 *   "cwt_mfma"       0|1   number_cwt_peaks' convolutions on the float64 matrix cores (0: measured 4 % slower)
 *   "entropy_route"  0|1|2 bit-matrix sweep | windowed pair sweep | general kernel
 *   "force_long"     0|1   the HBM-scratch build of the family kernels whatever the length
 *   "row_form"       0|1   BASIC / TREND columns of series of <= 256 samples four series to a wavefront (1)
 *   "seq_rows"       -1|0|1 lempel_ziv_complexity's symbol rows in LDS (0), in HBM (1), or in HBM where that puts more series on a CU (-1, default)
 *   "host_chunks"    n     row chunks of the TSFA_HOST pipeline (0: by batch size)
 *   "fill" v / "fill_off"  pre-fill the result matrix with v before the kernels run (audits: every cell is written anyway)
 * plan == NULL: library-wide options ("relevance_batch" n: columns per sort batch of tsfa_relevance_*).
 * Unknown names return TSFA_ERR_INVALID.  The reference has no counterpart (its knobs are n_jobs / chunksize).
 */
int tsfa_plan_set_option(tsfa_plan *plan, const char *name, double value);

/* Optional: promise that every series of the following tsfa_extract* calls on this plan has min_len <= length <=
 * max_len.  The calls then skip their length scan and the host synchronisation it needs, so device-pointer calls on one
 * stream (e.g. the row chunks of a shard whose results are exchanged chunk by chunk) are enqueued back to back.  A batch
 * outside the promised range is undefined behaviour.  (0, 0) withdraws the promise.  The reference has no counterpart:
 * its per-series dispatch (extraction.py:308) sizes nothing ahead. */
int tsfa_plan_set_length_hint(tsfa_plan *plan, int64_t min_len, int64_t max_len);

/* Host-side packer helper.  The reference groups a long DataFrame into per-(id, kind) pd.Series with a pandas groupby
 * (data.py:233-291, LongTsFrameAdapter); for the usual layout -- rows already grouped by ascending id, every group already
 * in sort order -- the ragged buffer tsfa_extract wants is the value column itself, and all that is needed is ONE
 * multi-threaded pass that (a) proves the layout, (b) finds the group boundaries, (c) performs the reference's NaN check
 * of the value column (data.py:148-167).
 *   ids / sort / values: the columns (sort and values may be NULL), element types per tsfa_dtype
 *   *flags:    TSFA_PACK_UNSORTED  the layout is something else (the caller falls back to a sorting packer)
 *              TSFA_PACK_VALUE_NAN a value is NaN (the caller raises the reference's ValueError)
 *   *n_groups: number of series; tsfa_pack_offsets then writes the n_groups + 1 row offsets (same thread). */
#define TSFA_PACK_UNSORTED 1
#define TSFA_PACK_VALUE_NAN 2
int tsfa_pack_scan(const void *ids, int32_t id_type, const void *sort, int32_t sort_type, const void *values,
                   int32_t value_type, int64_t n_rows, int32_t *flags, int64_t *n_groups);
int tsfa_pack_offsets(int64_t *offsets, int64_t n_groups, int64_t n_rows);

/* Page-locked host memory for the TSFA_HOST form.  tsfa_extract* accepts ANY host pointer; from pageable memory the HIP
 * runtime stages every transfer through its own bounce buffers, from memory obtained here the copy engines read and
 * write it directly, so the chunked copy-in / compute / copy-out pipeline of tsfa_extract runs at PCIe rate.  The
 * Python host allocates the packed sample buffer and the result matrix (the future DataFrame's block) this way.  The
 * reference has no counterpart (its matrix is assembled by data.pivot, data.py:86-121, from Python tuples). */
int tsfa_host_alloc(void **ptr, size_t bytes);
int tsfa_host_free(void *ptr);

/* Device-resident matrices: the chain extract -> impute -> select (tsfresh/convenience/relevant_extraction.py:18,
 * SURVEY.md 8f N3) hands a float64 [n_ids x n_features] matrix from step to step.  tsfa_extract, tsfa_impute and
 * tsfa_relevance_* accept TSFA_DEVICE pointers; with the four entry points below a host in any language keeps that
 * matrix in HBM for the whole chain and fetches only the selected columns (the reference's `X.loc[:, relevant]`,
 * feature_selection/selection.py:181).  tsfa_device_copy: to_device != 0 copies host -> device, else device -> host.
 * tsfa_gather_columns: out_host[r * n_sel + c] = X[r * ld + cols[c]].
 * tsfa_scatter_columns: dst[r * ld_dst + cols[c]] = src[r * n_src + c], dst and src in device memory, cols on the host: the
 * columns of ONE native plan of a settings object that needs several (augmented_dickey_fuller with several autolag values,
 * more than 128 cwt_coefficients columns) land in their places of the caller's matrix without leaving the device. */
int tsfa_device_alloc(void **ptr, size_t bytes, int32_t device);
int tsfa_device_free(void *ptr, int32_t device);
int tsfa_device_copy(void *dst, const void *src, size_t bytes, int32_t to_device, int32_t device);
int tsfa_gather_columns(const double *X, int64_t n_rows, int64_t ld, const int32_t *cols, int64_t n_sel, double *out_host,
                        int32_t device);
int tsfa_scatter_columns(double *dst, int64_t ld_dst, const int32_t *cols, const double *src, int64_t n_rows, int64_t n_src,
                         int32_t device);

/* ---- feature selection: relevance statistics of the extracted matrix ----
 * Replaces the per-feature loop of tsfresh/feature_selection/relevance.py:214-322 (calculate_relevance_table ->
 * _calculate_relevance_table_for_implicit_target) for classification targets: one call yields, for EVERY column of
 * X, what the reference's univariate tests need
 *   significance_tests.py:84  target_binary_feature_real_test  (scipy.stats.mannwhitneyu): the mid-rank sum of the
 *                             rows of each class and the tie term sum(t^3 - t) of the normal approximation,
 *   significance_tests.py:43  target_binary_feature_binary_test (scipy.stats.fisher_exact): for a column with two
 *                             distinct values, the rows of each class that hold the larger one,
 *   relevance.py:396          get_feature_type: the number of distinct values (1 constant, 2 binary, more: real).
 * X: row-major float64 [n_rows x n_cols], leading dimension ld, host or device memory (`space`), no NaNs.
 * y_codes: host int32[n_rows], class codes 0 .. n_classes-1 (<= 256).  Outputs are host arrays:
 * cols[n_cols], rank_sums[n_cols * n_classes], hi_counts[n_cols * n_classes].  Synchronous. */
typedef struct tsfa_relevance_col {
    int64_t n_unique;  /* distinct values of the column */
    double v_lo, v_hi; /* smallest / largest value */
    double tie_term;   /* sum over tie groups of t^3 - t */
} tsfa_relevance_col;
int tsfa_relevance_classes(const double *X, int64_t n_rows, int64_t n_cols, int64_t ld, int32_t space,
                           const int32_t *y_codes, int32_t n_classes, int32_t device, tsfa_relevance_col *cols,
                           double *rank_sums, int64_t *hi_counts);

/* ... plus, for the 'smir' option (significance_tests.py:121, scipy.stats.ks_2samp of the column split by the label),
 * ks_d[n_cols * n_classes]: the Kolmogorov-Smirnov distance of every column for every label against the rest. */
int tsfa_relevance_classes_ks(const double *X, int64_t n_rows, int64_t n_cols, int64_t ld, int32_t space,
                              const int32_t *y_codes, int32_t n_classes, int32_t device, tsfa_relevance_col *cols,
                              double *rank_sums, int64_t *hi_counts, double *ks_d);

/* The same for real-valued targets (relevance.py:303-316):
 *   significance_tests.py:170 target_real_feature_real_test   (scipy.stats.kendalltau, asymptotic): discordant pairs
 *                             and the tie statistics of the column and of the (column, target) pairs,
 *   significance_tests.py:135 target_real_feature_binary_test (scipy.stats.ks_2samp): for a two-valued column the
 *                             Kolmogorov-Smirnov distance of the target split by the column.
 * y_rank: host int32[n_rows], dense ranks of the target (equal targets share a rank); y_perm: host int32[n_rows], the
 * rows in ascending target order; y_end: host uint8[n_rows], 1 where position p of that order ends a tie group.
 * cols: host array [n_cols].  Synchronous. */
typedef struct tsfa_relevance_real_col {
    int64_t n_unique;   /* distinct values of the column */
    double v_lo, v_hi;  /* smallest / largest value */
    int64_t dis;        /* discordant pairs */
    int64_t xtie, ntie; /* tied pairs of the column / of (column, target) */
    double x0, x1;      /* sum t(t-1)(t-2), sum t(t-1)(2t+5) over the column's tie groups */
    int64_t n_hi;       /* two-valued column: rows holding v_hi */
    double ks_d;        /* two-valued column: sup |F_hi - F_lo| of the target */
} tsfa_relevance_real_col;
int tsfa_relevance_real(const double *X, int64_t n_rows, int64_t n_cols, int64_t ld, int32_t space, const int32_t *y_rank,
                        const int32_t *y_perm, const unsigned char *y_end, int32_t device, tsfa_relevance_real_col *cols);
/* Pr(D >= h / lcm(m, n)), two-sided two-sample Kolmogorov-Smirnov, m != n, g = gcd(m, n): the exact lattice-path tail of
 * scipy.stats.ks_2samp(method="exact") -- a scalar function of four integers (host arithmetic, no device needed). */
double tsfa_ks_outer_prob(int64_t m, int64_t n, int64_t g, int64_t h);

/* ---- impute: tsfresh/utilities/dataframe_functions.py:49-214 on the feature matrix ----
 * Per column: the largest / smallest / median FINITE value (get_range_values_per_column :142-180; a column without a
 * finite value counts as zeros); then, in place, +inf -> max, -inf -> min, NaN -> median (impute_dataframe_range
 * :96-139).  X: row-major float64 [n_rows x n_cols], leading dimension ld, host or device memory (`space`).
 * col_max / col_min / col_median / finite_count: optional host arrays [n_cols] receiving the statistics and the number
 * of finite cells per column (0: the reference warns and fills with zeros).  Synchronous. */
int tsfa_impute(double *X, int64_t n_rows, int64_t n_cols, int64_t ld, int32_t space, int32_t device, double *col_max,
                double *col_min, double *col_median, int32_t *finite_count);

#ifdef __cplusplus
}
#endif
#endif /* TSFRESH_AMD_H */
