// host-side launch descriptors shared by tsfa_kernels.hip and tsfa_api.cpp
#ifndef TSFA_LAUNCH_H
#define TSFA_LAUNCH_H

#include <stddef.h>
#include <stdint.h>

#include "tsfa_specs.h"
#include "fam_seq.h"

// Length classes of a batch: class c holds the lengths in (2^(c+5), 2^(c+6)] (class 0 from 1, the last one up to
// 65535).  A ragged batch is launched class by class (merged into at most TSFA_N_LEN_CLASSES groups), each with the
// LDS carve and workgroup size of ITS longest series, through an index list per group.
#define TSFA_N_LEN_CLASSES 11
#define TSFA_LEN_STATS (3 + 3 * TSFA_N_LEN_CLASSES)
static inline int tsfa_len_class(long long l) {
    int c = 0;
    while (c < TSFA_N_LEN_CLASSES - 1 && l > (64LL << c)) ++c;
    return c;
}
struct TsfaClassMap {
    int group_of[TSFA_N_LEN_CLASSES];  // length class -> launch group
    int base[TSFA_N_LEN_CLASSES];      // first entry of the group's list in `sel`
};

struct TsfaLaunch {
    int fam;
    int dtype;  // 0 = f32, 1 = f64
    const void *values;
    const int64_t *starts;  // series s occupies values[starts[s] .. ends[s]); for a ragged batch ends = starts + 1
    const int64_t *ends;
    int64_t n_series;       // workgroups of this launch
    const int *sel;         // device: the series of this launch (indices into starts/ends/out rows), or null = 0..n-1
    const TsfaSpec *specs;  // device
    int nspecs;
    const TsfaSpec *bspecs; // device: the BASIC family's specs next to the SORT family's in `specs` (tsfa_launch_stream only)
    int nbspecs;
    double *out;
    int64_t ld;
    int maxn;   // longest series of the batch (LDS is sized for it)
    int skip_le;  // BASIC / TREND: series of at most this many samples are left to the row form (tsfa_launch_rows); 0: none
    int nt;     // workgroup size
    void *stream;
    // family extras
    const double *dectab;   // BASIC: decimal table for benford_correlation
    TsfaCqPlan cq;          // SORT: the plan's change_quantiles corridors (tsfa_prepare_family)
    TsfaAltPlan alt;        // BASIC: the plan's agg_linear_trend regressions (tsfa_prepare_family)
    const double *times;    // BASIC: per-sample hours since the series' first timestamp (linear_trend_timewise) or null
    const double *twc, *tws;  // SPECTRAL: shared FFT twiddles
    int dft_n;              // SPECTRAL: DFT twiddle slots held in LDS
    double *gscratch;       // SPECTRAL: HBM scratch of the Bluestein FFTs, one slot of gscratch_n doubles per workgroup (or null)
    int gscratch_n;
    TsfaSeqGroup seq;       // SEQ: the (<= TSFA_LZ_MAX_GROUP) specs this launch parses side by side
    unsigned char *seq_rows;  // SEQ, seq.grows: HBM for the symbol rows, seq.stride bytes per workgroup of the launch
    int ar_P;               // AR: leading dimension of the normal matrices
    int ar_P_dd;            // ... and of the second pass's (ar_coefficient orders beyond the first pass's table live only there)
    double *dd_scratch;     // AR second pass: HBM slots for double-double matrices beyond LDS (or null), dd_slots of them
    int dd_slots;
    int ar_has_coef;        // AR: the plan holds ar_coefficient columns
    long long *deg_list;    // AR: series listed for the double-double second pass ((index << 2) | calculator bits) ...
    int *deg_count;         // ... and their number (device; zeroed before the launch)
    double *stats_out;      // BASIC (k_basic): where to leave the per-series statistics record (TSFA_STATS_N doubles per series), or null
    const double *stats_in; // ENTROPY / AR / SEQ / SORT: that record, when k_basic of the same extraction ran before (or null: compute)
    const double *consts;   // SPECTRAL / CWT peaks: the plan's constant tables (tsfa_build_consts), device memory
    double *pf_buf;         // SORT: records of the Langevin fits left to k_langevin_dd (fam_langevin_dd.h) ...
    int *pf_count;          // ... their number (device; zeroed before the launch) ...
    int pf_slot;            // ... the doubles per record (tsfa_pf_slot_doubles) ...
    int pf_cap;             // ... and the records the buffer holds (n_series x distinct (m, r) fits, hints[SORT].e)
    unsigned short *perm_buf;  // ENTROPY (bit-matrix sweep) writes / SORT reads: sample order of every series, perm_stride entries each
    int perm_stride;
    int64_t gscratch_slots; // SPECTRAL: slots of gscratch_n doubles in gscratch (one per workgroup of a launch)
    int bluestein_min;      // SPECTRAL: non-power-of-two lengths from here on take the chirp-z transform (with gscratch)
    int cwt_rowv;           // CWT peaks: bit 0: the series is staged in LDS between zero halos; bit 1: phase A on the matrix cores (TSFA_CWT_MFMA=1)
    int hint_a, hint_b, hint_c, hint_d, hint_e;  // tsfa_prepare_family (BASIC, SORT, SPECTRAL, AR)
    unsigned char *long_scratch;  // HBM scratch of the long-series build (tsfa_launch_family_long) ...
    size_t long_bytes;            // ... and its size: slots of one working set each, one per resident workgroup
    int ent_cnt;            // ENTROPY: per-template LDS counters (symmetric sweep)
    int ent_fast;           // ENTROPY: only m = 2 specs and ent_cnt: the kernel variant without the fallback sweeps
};

struct TsfaCwtLaunch {
    int dtype;
    const void *values;
    const int64_t *starts, *ends;
    int64_t n_series;
    const double *W;       // [Cpad][S4]
    int S4, C;
    const int *cols;       // output column per filter
    const int *coeff_idx;  // requested coefficient index per filter (NaN when >= series length)
    double *out;
    int64_t ld;
    void *stream;
};

size_t tsfa_family_lds_bytes(int fam, int maxn, int nt, int aux);
size_t tsfa_entropy_lds_bytes(int maxn, int with_cnt);
size_t tsfa_seq_lds_bytes(const TsfaSeqGroup &g);
int tsfa_launch_family(const TsfaLaunch &a);
int tsfa_launch_rows(const TsfaLaunch &a);          // BASIC / TREND: the series of at most TSFA_ROW_MAXN samples, four per wavefront (k_basic_rows / k_trend_rows)
int tsfa_launch_family_long(const TsfaLaunch &a);   // working set in a.long_scratch instead of LDS (any length <= 65535)
// k_general (fam_general.h): a.specs / a.nspecs = the plan's GENERAL columns, every series of the batch in one launch
size_t tsfa_general_slot_doubles(int maxn, const TsfaGenPlan &g);
int tsfa_launch_general(const TsfaLaunch &a, const TsfaGenPlan &g, double *scratch, size_t slot_doubles, int slots, const double *pool);
int tsfa_launch_ar_degenerate(const TsfaLaunch &a);
int tsfa_launch_langevin_dd(const TsfaLaunch &a);     // second pass of TSFA_FAM_SORT: the ill-conditioned Langevin fits k_sort recorded
int tsfa_launch_perm(const TsfaLaunch &a);            // beside TSFA_FAM_SORT: every permutation_entropy column (k_perm, fam_perm.h); a.hint_d = (stride << 8) | dimensions, a.nt threads
size_t tsfa_perm_lds_bytes(int maxn, int nt, int elem_bytes);
int tsfa_launch_stream(const TsfaLaunch &a);         // BASIC closed forms + median in one read of the samples (k_stream), maxn <= 2048
int tsfa_stream_calc_ok(int calc);                   // is the calculator one of those k_stream serves?
int tsfa_launch_order_stats(const TsfaLaunch &a);    // SORT family holding only median / quantile columns, maxn <= 2048  // second pass of TSFA_FAM_AR over the series the first listed
int tsfa_launch_cwt(const TsfaCwtLaunch &a);
// value: NaN (a diagnostics run may ask for a sentinel: TSFA_DEBUG_FILL).  cols / n_fill: only these columns (device list);
// cols == nullptr: every column of the plan
int tsfa_launch_fill_nan(double *out, int64_t n_rows, int64_t n_cols, int64_t ld, void *stream, double value,
                         const int *cols, int n_fill);
int tsfa_launch_len_stats(const int64_t *starts, const int64_t *ends, int64_t n_series, long long *stats, void *stream);
int tsfa_launch_class_fill(const int64_t *starts, const int64_t *ends, int64_t n_series, const TsfaClassMap &g, int *cursor,
                           int *sel, void *stream);

#endif
