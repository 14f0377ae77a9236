// gfx950 kernels: one workgroup per series per family + the MFMA contraction for cwt_coefficients.
#include <hip/hip_runtime.h>

#include "fam_ar.h"
#include "fam_ar_dd.h"
#include "fam_basic.h"
#include "fam_cwt.h"
#include "fam_entropy.h"
#include "fam_seq.h"
#include "fam_perm.h"
#include "fam_sort.h"
#include "fam_spectral.h"
#include "fam_general.h"
#include "tsfa_launch.h"
#include "tsfa_layout.h"

#include <algorithm>

// This file is compiled twice.  tsfa_kernels.o: one workgroup per series, the working set carved from LDS.
// tsfa_kernels_long.o (tsfa_kernels_long.hip defines TSFA_LONG and renames the kernels): series too long for a CU's
// LDS -- the SAME per-series code with its working set carved from a per-workgroup slot of HBM scratch (L2 / Infinity
// Cache resident), a persistent grid that walks the series list, and workgroup barriers that also order global
// memory (tsfa_common.h: blk_sync).  Slower per sample, but no length limit below 65 535.
#if defined(TSFA_LONG)
#define TSFA_GS_PARAMS , unsigned char *__restrict__ gsc, size_t gslot
#define TSFA_GS_ARGS , gsc, gslot
#define TSFA_SERIES_BEGIN                                                           \
    unsigned char *const tsfa_base = gsc + (size_t)blockIdx.x * gslot;             \
    for (int64_t wi_ = blockIdx.x; wi_ < n_series; wi_ += gridDim.x) {             \
        const int64_t sidx = sel ? (int64_t)sel[wi_] : wi_;
#define TSFA_SERIES_END \
        __syncthreads(); \
    }
#define TSFA_SERIES_SKIP continue
#else
extern __shared__ __attribute__((aligned(16))) unsigned char tsfa_smem[];
#define TSFA_GS_PARAMS
#define TSFA_GS_ARGS
#define TSFA_SERIES_BEGIN                                                                                                   \
    if ((int64_t)blockIdx.x >= n_series) return;                                                                            \
    unsigned char *const tsfa_base = tsfa_smem;                                                                             \
    {                                                                                                                       \
        const int64_t sidx = sel ? (int64_t)sel[blockIdx.x] : (int64_t)blockIdx.x; /* length-class launch: its series list */
#define TSFA_SERIES_END }
#define TSFA_SERIES_SKIP return
#endif

template <typename T>
__device__ __forceinline__ void stage_series(const Blk &b, const T *__restrict__ g, int n, double *xs) {
    for (int i = b.tid; i < n; i += b.nt) xs[i] = (double)g[i];
    blk_sync();
}

// ---------------------------------------------------------------------------------------------
// PART 1: the BASIC family; PART 2 (k_trend below): the TREND family -- same series staging, separate register
// and LDS budgets (fam_basic.h)
template <typename T, int PART>
__device__ __forceinline__ void basic_body(const T *__restrict__ values, const int64_t *__restrict__ starts, const int64_t *__restrict__ ends, int64_t n_series, const int *__restrict__ sel,
                        const TsfaSpec *__restrict__ specs, int nspecs, double *__restrict__ out, int64_t ld,
                        const double *__restrict__ dectab, int maxn, int hint_a, int hint_b,
                        const double *__restrict__ times, const TsfaAltPlan &alt, int n_loop, int n_count, int n_sum,
                        double *__restrict__ stats_out, int skip_le TSFA_GS_PARAMS) {
    TSFA_SERIES_BEGIN
    const int64_t off = starts[sidx];
    const int n = (int)(ends[sidx] - off);
    if (n <= skip_le) TSFA_SERIES_SKIP;   // evaluated by the row form (k_basic_rows / k_trend_rows), whatever the launch group
    BasicLds L;
    L.carve(tsfa_base, maxn, blockDim.x, (int)sizeof(T), PART, (PART == 2) ? alt.small_w : 0);
    TSFA_TICKS_BEGIN();
    Blk b{(int)threadIdx.x, (int)blockDim.x, L.red, L.np};
    T *xs = (T *)L.xs;  // resident in the input precision (half the LDS for float32), read as float64
    {
        const T *__restrict__ g = values + off;
        for (int i = b.tid; i < n; i += b.nt) xs[i] = g[i];
        blk_sync();
    }
    fam_basic_series<PART>(b, XsView<T>{xs}, n, specs, nspecs, out + sidx * ld, L.w, L.cum, L.altc, L.iw, dectab, hint_a,
                           hint_b, alt, L.stage, times ? times + off : nullptr, n_loop, L.ctx, n_count, n_sum,
                           stats_out ? stats_out + sidx * TSFA_STATS_N : nullptr);
    TSFA_TICKS_END();
    TSFA_SERIES_END
}

template <typename T>
__global__ void __launch_bounds__(1024) k_basic(const T *__restrict__ values, const int64_t *__restrict__ starts, const int64_t *__restrict__ ends, int64_t n_series, const int *__restrict__ sel,
                        const TsfaSpec *__restrict__ specs, int nspecs, double *__restrict__ out, int64_t ld,
                        const double *__restrict__ dectab, int maxn, int hint_a, int n_loop, int n_count, int n_sum,
                        double *__restrict__ stats_out, int skip_le TSFA_GS_PARAMS) {
    TsfaAltPlan alt;
    alt.nkeys = 0; alt.want_p = 0; alt.nq = 0; alt.small_w = 0;
    basic_body<T, 1>(values, starts, ends, n_series, sel, specs, nspecs, out, ld, dectab, maxn, hint_a, 0, nullptr, alt, n_loop,
                     n_count, n_sum, stats_out, skip_le TSFA_GS_ARGS);
}

// A plan whose BASIC columns are all closed forms of the per-series statistics (MinimalFCParameters: sum, mean, length,
// std, variance, rms, max, |max|, min): statistics + lane = column epilogue only.  A fraction of k_basic's registers, so
// twice the resident wavefronts for a kernel that waits on the numpy-order sums.
template <typename T>
__global__ void __launch_bounds__(256, 4) k_basic_lite(const T *__restrict__ values, const int64_t *__restrict__ starts, const int64_t *__restrict__ ends, int64_t n_series, const int *__restrict__ sel,
                        const TsfaSpec *__restrict__ specs, int nspecs, double *__restrict__ out, int64_t ld,
                        const double *__restrict__ dectab, int maxn, int hint_a TSFA_GS_PARAMS) {
    TsfaAltPlan alt;
    alt.nkeys = 0; alt.want_p = 0; alt.nq = 0; alt.small_w = 0;
#if defined(TSFA_LONG)
    basic_body<T, 5>(values, starts, ends, n_series, sel, specs, nspecs, out, ld, dectab, maxn, hint_a, 0, nullptr, alt, 0, 0, 0, nullptr, 0 TSFA_GS_ARGS);
#else
    // a persistent grid: the per-series work is a few thousand cycles, less than the dispatch of a workgroup costs
    unsigned char *const tsfa_base = tsfa_smem;
    BasicLds L;
    L.carve(tsfa_base, maxn, blockDim.x, (int)sizeof(T), 1);
    // ... and the samples of the NEXT series are fetched into registers while the current one is evaluated: with one
    // series per wavefront in flight the kernel waits on HBM latency, not bandwidth (0.4 GB in 0.33 ms).
    const bool prefetch = (maxn <= 16 * (int)blockDim.x);
    Blk b{(int)threadIdx.x, (int)blockDim.x, L.red, L.np};
    T *xs = (T *)L.xs;
    T pre[16];
    int64_t sidx_n = 0;
    int n_n = 0;
    if ((int64_t)blockIdx.x < n_series) {
        sidx_n = sel ? (int64_t)sel[blockIdx.x] : (int64_t)blockIdx.x;
        const int64_t off = starts[sidx_n];
        n_n = (int)(ends[sidx_n] - off);
        if (prefetch) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int i = b.tid + u * b.nt;
                pre[u] = (i < n_n) ? values[off + i] : (T)0;
            }
        }
    }
    for (int64_t wi = blockIdx.x; wi < n_series; wi += gridDim.x) {
        const int64_t sidx = sidx_n;
        const int n = n_n;
        if (prefetch) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int i = b.tid + u * b.nt;
                if (i < n) xs[i] = pre[u];
            }
        } else {
            const T *__restrict__ g = values + starts[sidx];
            for (int i = b.tid; i < n; i += b.nt) xs[i] = g[i];
        }
        blk_sync();
        const int64_t wn = wi + gridDim.x;
        if (wn < n_series) {
            sidx_n = sel ? (int64_t)sel[wn] : wn;
            const int64_t off = starts[sidx_n];
            n_n = (int)(ends[sidx_n] - off);
            if (prefetch) {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int i = b.tid + u * b.nt;
                    pre[u] = (i < n_n) ? values[off + i] : (T)0;
                }
            }
        }
        fam_basic_series<5>(b, XsView<T>{xs}, n, specs, nspecs, out + sidx * ld, L.w, L.cum, L.altc, L.iw, dectab, hint_a, 0,
                            alt, L.stage, nullptr, 0, L.ctx, 0, 0);
        blk_sync();
    }
#endif
}

template <typename T>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4))) k_trend(const T *__restrict__ values, const int64_t *__restrict__ starts, const int64_t *__restrict__ ends, int64_t n_series, const int *__restrict__ sel,
                        const TsfaSpec *__restrict__ specs, int nspecs, double *__restrict__ out, int64_t ld, int maxn,
                        int hint_b, const double *__restrict__ times, const TsfaAltPlan alt, int n_loop, int skip_le TSFA_GS_PARAMS) {
    basic_body<T, 2>(values, starts, ends, n_series, sel, specs, nspecs, out, ld, nullptr, maxn, 0, hint_b, times, alt, n_loop, 0, 0, nullptr, skip_le TSFA_GS_ARGS);
}

#if !defined(TSFA_LONG)
// Row form of the two kernels above (tsfa_common.h: BlkRow): the series of at most TSFA_ROW_MAXN samples, one per 16-lane row,
// four per one-wavefront workgroup; longer series of the launch leave their row idle (k_basic / k_trend evaluate them).
template <typename T, int PART>
__device__ __forceinline__ void basic_rows_body(const T *__restrict__ values, const int64_t *__restrict__ starts, const int64_t *__restrict__ ends, int64_t n_series, const int *__restrict__ sel,
                        const TsfaSpec *__restrict__ specs, int nspecs, double *__restrict__ out, int64_t ld,
                        const double *__restrict__ dectab, int maxn, int hint_a, int hint_b,
                        const double *__restrict__ times, const TsfaAltPlan &alt, int n_loop, int n_count, int n_sum,
                        double *__restrict__ stats_out, int row_bytes) {
    const int row = (int)(threadIdx.x >> 4);
    const int64_t wi = (int64_t)blockIdx.x * (64 / TSFA_ROW_LANES) + row;
    if (wi >= n_series) return;
    const int64_t sidx = sel ? (int64_t)sel[wi] : wi;
    const int64_t off = starts[sidx];
    const int n = (int)(ends[sidx] - off);
    if (n > TSFA_ROW_MAXN) return;
    BasicRowLds L;
    L.carve(tsfa_smem + (size_t)row * (size_t)row_bytes, maxn, (int)sizeof(T), PART, (PART == 2) ? alt.small_w : 0);
    BlkRow b;
    b.tid = (int)(threadIdx.x & (TSFA_ROW_LANES - 1));
    b.nt = TSFA_ROW_LANES;
    b.red = L.red;
    b.np = nullptr;
    T *xs = (T *)L.xs;
    {
        const T *__restrict__ g = values + off;
        for (int i = b.tid; i < n; i += TSFA_ROW_LANES) xs[i] = g[i];
        blk_sync();
    }
    fam_basic_series<PART>(b, XsView<T>{xs}, n, specs, nspecs, out + sidx * ld, L.w, L.cum, L.altc, L.iw, dectab, hint_a,
                           hint_b, alt, (TsfaSpec *)nullptr, times ? times + off : nullptr, n_loop, L.ctx, n_count, n_sum,
                           stats_out ? stats_out + sidx * TSFA_STATS_N : nullptr);
}

#if !defined(TSFA_ROWS_WPE)
#define TSFA_ROWS_WPE 4
#endif
template <typename T>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(TSFA_ROWS_WPE))) k_basic_rows(const T *__restrict__ values, const int64_t *__restrict__ starts, const int64_t *__restrict__ ends, int64_t n_series, const int *__restrict__ sel,
                        const TsfaSpec *__restrict__ specs, int nspecs, double *__restrict__ out, int64_t ld,
                        const double *__restrict__ dectab, int maxn, int hint_a, int n_loop, int n_count, int n_sum,
                        double *__restrict__ stats_out, int row_bytes) {
    TsfaAltPlan alt;
    alt.nkeys = 0; alt.want_p = 0; alt.nq = 0; alt.small_w = 0;
    basic_rows_body<T, 1>(values, starts, ends, n_series, sel, specs, nspecs, out, ld, dectab, maxn, hint_a, 0, nullptr, alt, n_loop,
                          n_count, n_sum, stats_out, row_bytes);
}

template <typename T>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(TSFA_ROWS_WPE))) k_trend_rows(const T *__restrict__ values, const int64_t *__restrict__ starts, const int64_t *__restrict__ ends, int64_t n_series, const int *__restrict__ sel,
                        const TsfaSpec *__restrict__ specs, int nspecs, double *__restrict__ out, int64_t ld, int maxn,
                        int hint_b, const double *__restrict__ times, const TsfaAltPlan alt, int n_loop, int row_bytes) {
    basic_rows_body<T, 2>(values, starts, ends, n_series, sel, specs, nspecs, out, ld, nullptr, maxn, 0, hint_b, times, alt, n_loop, 0, 0,
                          nullptr, row_bytes);
}
#endif

template <typename T>
__global__ void __launch_bounds__(1024) k_sort(const T *__restrict__ values, const int64_t *__restrict__ starts, const int64_t *__restrict__ ends, int64_t n_series, const int *__restrict__ sel,
                       const TsfaSpec *__restrict__ specs, int nspecs, double *__restrict__ out, int64_t ld, int maxn,
                       const TsfaCqPlan cqplan, int n_loop, int w_doubles, double *__restrict__ pf_buf,
                       int *__restrict__ pf_count, int pf_slot, int pf_cap, const unsigned short *__restrict__ perm_buf, int perm_stride,
                       const double *__restrict__ stats_in, int pe_hint TSFA_GS_PARAMS) {
    TSFA_SERIES_BEGIN
    const int64_t off = starts[sidx];
    const int n = (int)(ends[sidx] - off);
    SortLds L;
    L.carve(tsfa_base, maxn, blockDim.x, (int)sizeof(T), w_doubles);
    TSFA_TICKS_BEGIN();
    Blk b{(int)threadIdx.x, (int)blockDim.x, L.red, L.np};
    T *xs = (T *)L.xs;  // resident in the input precision, like its sorted copy
    {
        const T *__restrict__ g = values + off;
        for (int i = b.tid; i < n; i += b.nt) xs[i] = g[i];
        blk_sync();
    }
    fam_sort_series<T>(b, xs, n, specs, nspecs, out + sidx * ld, (T *)L.srt, L.w, L.iw, cqplan, L.cq, L.stage,
                       n_loop, L.ctx, w_doubles, FrDefer{pf_buf, pf_count, pf_slot, (long long)sidx, 0, pf_cap},
                       perm_buf ? perm_buf + (size_t)sidx * (size_t)perm_stride : nullptr,
                       stats_in ? stats_in + sidx * TSFA_STATS_N : nullptr, pe_hint);
    TSFA_TICKS_END();
    TSFA_SERIES_END
}

#if !defined(TSFA_LONG)
// every permutation_entropy column of the plan (fam_perm.h): the SORT family's spec list, its other columns left to k_sort
template <typename T>
__global__ void __launch_bounds__(1024) k_perm(const T *__restrict__ values, const int64_t *__restrict__ starts, const int64_t *__restrict__ ends, int64_t n_series, const int *__restrict__ sel,
                       const TsfaSpec *__restrict__ specs, int nspecs, double *__restrict__ out, int64_t ld, int maxn, int tau) {
    TSFA_SERIES_BEGIN
    const int64_t off = starts[sidx];
    const int n = (int)(ends[sidx] - off);
    PermLds L;
    L.carve(tsfa_base, maxn, blockDim.x, (int)sizeof(T), TSFA_PE_HIST_WORDS, TSFA_PE_LOGS + TSFA_PE_MAXD + 1);
    TSFA_TICKS_BEGIN();
    Blk b{(int)threadIdx.x, (int)blockDim.x, L.red, nullptr};
    T *xs = (T *)L.xs;
    {
        const T *__restrict__ g = values + off;
        for (int i = b.tid; i < n; i += b.nt) xs[i] = g[i];
        blk_sync();
    }
    fam_perm_series<T>(b, xs, n, specs, nspecs, out + sidx * ld, tau, L.iw, L.ltab);
    TSFA_TICKS_END();
    TSFA_SERIES_END
}
#endif

// BL: with the chirp-z transform of long non-power-of-two series (launched when the plan gave the group HBM scratch)
template <typename T, bool BL>
__global__ void __launch_bounds__(1024) k_spectral(const T *__restrict__ values, const int64_t *__restrict__ starts, const int64_t *__restrict__ ends, int64_t n_series, const int *__restrict__ sel,
                           const TsfaSpec *__restrict__ specs, int nspecs, double *__restrict__ out, int64_t ld,
                           int maxn, int dft_n, double *__restrict__ gscratch, int gscratch_n, int bl_min,
                           const double *__restrict__ twc, const double *__restrict__ tws, int hint_a, int hint_b,
                           const double *__restrict__ consts TSFA_GS_PARAMS) {
    TSFA_SERIES_BEGIN
    const int64_t off = starts[sidx];
    const int n = (int)(ends[sidx] - off);
    SpectralLds L;
    L.carve(tsfa_base, maxn, dft_n, (int)sizeof(T));
    TSFA_TICKS_BEGIN();
    Blk b{(int)threadIdx.x, (int)blockDim.x, L.red, nullptr};
    T *xs = (T *)L.xs;  // resident in the input precision
    {
        const T *__restrict__ g = values + off;
        for (int i = b.tid; i < n; i += b.nt) xs[i] = g[i];
        blk_sync();
    }
    // gscratch: one slot of gscratch_n doubles per workgroup for the Bluestein FFTs of long series (or null)
    double *gs = gscratch ? gscratch + (size_t)blockIdx.x * (size_t)gscratch_n : nullptr;
    fam_spectral_series<T, BL>(b, xs, n, specs, nspecs, out + sidx * ld, L.Xr, L.Xi, L.tc, L.ts, L.win, L.pxx, L.iw, twc, tws,
                        hint_a, hint_b, gs, consts ? consts + TSFA_CONSTS_HANN : nullptr, L.chirp_tab(), bl_min);
    TSFA_TICKS_END();
    TSFA_SERIES_END
}

template <typename T>
__global__ void __launch_bounds__(1024) k_ar(const T *__restrict__ values, const int64_t *__restrict__ starts, const int64_t *__restrict__ ends, int64_t n_series, const int *__restrict__ sel,
                     const TsfaSpec *__restrict__ specs, int nspecs, double *__restrict__ out, int64_t ld, int maxn,
                     int P, int hint_acf, int hint_pacf, int hint_adf, int n_loop, long long *__restrict__ deg_list,
                     int *__restrict__ deg_count, const double *__restrict__ stats_in TSFA_GS_PARAMS) {
    TSFA_SERIES_BEGIN
    const int64_t off = starts[sidx];
    const int n = (int)(ends[sidx] - off);
    ArLds L;
    L.carve(tsfa_base, maxn, P, (int)sizeof(T));
    TSFA_TICKS_BEGIN();
    Blk b{(int)threadIdx.x, (int)blockDim.x, L.red, L.np};
    const T *g = values + off;
    const int flags = fam_ar_series<T>(b, [=](int i) { return (double)g[i]; }, n, specs, nspecs, out + sidx * ld,
                                       (void *)L.xc, L.aw, P, hint_acf, hint_pacf, hint_adf, n_loop,
                                       stats_in ? stats_in + sidx * TSFA_STATS_N : nullptr);
    // rank-deficient / ill-conditioned regressions: list the series for k_ar_degenerate
    if (flags && threadIdx.x == 0) deg_list[atomicAdd(deg_count, 1)] = ((long long)sidx << 2) | flags;
    TSFA_TICKS_END();
    TSFA_SERIES_END
}

#if !defined(TSFA_LONG)
// second pass of the SORT family (fam_langevin_dd.h): one LANE per recorded fit, np.polyfit's scaled design + rank cut in
// double-double; overwrites the friedrich_coefficients / max_langevin_fixed_point columns of the same (m, r) that k_sort
// wrote from its float64 QR.
__global__ void __launch_bounds__(64) k_langevin_dd(const double *__restrict__ pf_buf, const int *__restrict__ pf_count, int pf_slot, int pf_cap,
                     const TsfaSpec *__restrict__ specs, int nspecs, double *__restrict__ out, int64_t ld) {
    const int cnt = min(*pf_count, pf_cap);
    const int rmax = (pf_slot - TSFA_PF_HDR) / 2;
    for (int i = blockIdx.x * 64 + threadIdx.x; i < cnt; i += gridDim.x * 64) {
        const double *rec = pf_buf + (size_t)i * (size_t)pf_slot;
        const int64_t sidx = (int64_t)rec[0];
        const int sp0 = (int)rec[1], k = (int)rec[2];
        const TsfaSpec s0 = specs[sp0];
        int m, r;
        if (s0.calc == TSFA_C_FRIEDRICH_COEFFICIENTS) { m = (int)s0.p[1]; r = (int)s0.p[2]; }
        else { m = (int)s0.p[0]; r = (int)s0.p[1]; }
        double coef[TSFA_PF_MAXC];
        const double *xm = rec + TSFA_PF_HDR, *ym = rec + TSFA_PF_HDR + rmax;
        polyfit_svd_dd([=](int j) { return xm[j]; }, [=](int j) { return ym[j]; }, k, m, coef);
        double *row = out + sidx * ld;
        for (int s = 0; s < nspecs; ++s) {
            const TsfaSpec sp = specs[s];
            if (sp.calc == TSFA_C_FRIEDRICH_COEFFICIENTS && (int)sp.p[1] == m && (int)sp.p[2] == r) {
                const int c = (int)sp.p[0];
                row[sp.col] = (c >= 0 && c <= m) ? coef[c] : TSFA_NAN;
            } else if (sp.calc == TSFA_C_MAX_LANGEVIN_FIXED_POINT && (int)sp.p[0] == m && (int)sp.p[1] == r) {
                row[sp.col] = max_real_root_deg3(coef, m + 1);
            }
        }
    }
}

// second pass of the AR family (fam_ar_dd.h): the listed series, one workgroup each, double-double normal equations.
// The list order is arbitrary (atomics); every listed series writes only its own row, so the result is not.
template <typename T>
__global__ void __launch_bounds__(64) k_ar_degenerate(const T *__restrict__ values, const int64_t *__restrict__ starts, const int64_t *__restrict__ ends,
                     const TsfaSpec *__restrict__ specs, int nspecs, double *__restrict__ out, int64_t ld, int P,
                     const long long *__restrict__ deg_list, const int *__restrict__ deg_count, int adf_mode,
                     double *__restrict__ gscratch) {
    ArDdLds L;
    L.carve(tsfa_smem, P);
    // matrices beyond a CU's LDS (ADF's maxlag grows with the length: P = 71 at 100 000 samples is 173 KB of double-double
    // normal equations): a slot of HBM scratch per workgroup instead, the reduction scratch stays in LDS
    if (gscratch != nullptr) L.scratch = gscratch + (size_t)blockIdx.x * (size_t)ArDdLds::scratch_doubles(P);
    Blk b{(int)threadIdx.x, (int)blockDim.x, L.red, nullptr};
    const int cnt = *deg_count;
    for (int i = blockIdx.x; i < cnt; i += gridDim.x) {
        const long long e = deg_list[i];
        const int64_t sidx = e >> 2;
        const int64_t off = starts[sidx];
        const int n = (int)(ends[sidx] - off);
        const T *g = values + off;
        fam_ar_degenerate_series(b, [=](int k) { return (double)g[k]; }, n, specs, nspecs, out + sidx * ld, L.scratch, P,
                                 (int)(e & 3), adf_mode);
        blk_sync();
    }
}

#endif  // !TSFA_LONG

// FAST: symmetric sweep only (m = 2 specs, LDS counters fit) -- see fam_entropy_series
template <typename T, bool FAST>
__global__ void __launch_bounds__(1024) k_entropy(const T *__restrict__ values, const int64_t *__restrict__ starts, const int64_t *__restrict__ ends, int64_t n_series, const int *__restrict__ sel,
                          const TsfaSpec *__restrict__ specs, int nspecs, double *__restrict__ out, int64_t ld,
                          int maxn, int with_cnt, const double *__restrict__ stats_in TSFA_GS_PARAMS) {
    TSFA_SERIES_BEGIN
    const int64_t off = starts[sidx];
    const int n = (int)(ends[sidx] - off);
    EntropyLds L;
#if defined(TSFA_LONG)
    typedef unsigned int ent_idx;   // the working set lives in HBM: a 32-bit sample order, series of any length
#else
    typedef unsigned short ent_idx;
#endif
    L.carve(tsfa_base, maxn, with_cnt, (int)sizeof(ent_idx));
    TSFA_TICKS_BEGIN();
    Blk b{(int)threadIdx.x, (int)blockDim.x, L.red, L.np};
    stage_series(b, values + off, n, L.xs);
    fam_entropy_series<double, FAST, sizeof(T) == 4, ent_idx>(b, L.xs, n, specs, nspecs, out + sidx * ld, L.thr, (ent_idx *)(void *)L.perm,
                                                               L.refs, L.cnt, 1, stats_in ? stats_in + sidx * TSFA_STATS_N : nullptr);
    TSFA_TICKS_END();
    TSFA_SERIES_END
}

#if defined(TSFA_LONG)
// Bit-matrix sweep beyond a CU's LDS (fam_entropy_hbits.h): every spec has m = 2, every series of the launch at most
// TSFA_ENTH_MAXN samples.  1024 threads, one workgroup per CU (the table of a column part fills its LDS); a persistent grid
// over the series, one HBM slot (EntropyHugeSlot) per workgroup.
extern __shared__ __attribute__((aligned(16))) unsigned char tsfa_hsmem[];
template <typename T>
__global__ void __launch_bounds__(1024) kl_entropy_hbits(const T *__restrict__ values, const int64_t *__restrict__ starts, const int64_t *__restrict__ ends, int64_t n_series, const int *__restrict__ sel,
                          const TsfaSpec *__restrict__ specs, int nspecs, double *__restrict__ out, int64_t ld, int maxn,
                          const double *__restrict__ stats_in, unsigned char *__restrict__ gsc, size_t gslot) {
    EntropyHugeSlot H;
    H.carve(gsc + (size_t)blockIdx.x * gslot, maxn);
    double *red = (double *)(void *)tsfa_hsmem;
    unsigned int *table = (unsigned int *)(void *)(tsfa_hsmem + TSFA_RED_DOUBLES * sizeof(double));
    unsigned int *wtot = (unsigned int *)(void *)(tsfa_hsmem + entropy_huge_lds_bytes(maxn) - 64 - (size_t)2 * TSFA_ENTB_MAXWAVES * TSFA_ENTH_S * sizeof(unsigned int));
    Blk b{(int)threadIdx.x, (int)blockDim.x, red, (NpScratch *)(void *)table};
    for (int64_t wi = blockIdx.x; wi < n_series; wi += gridDim.x) {
        const int64_t sidx = sel ? (int64_t)sel[wi] : wi;
        const int64_t off = starts[sidx];
        const int n = (int)(ends[sidx] - off);
        fam_entropy_series_hbits<T>(b, values + off, n, specs, nspecs, out + sidx * ld, H, table, wtot,
                                    stats_in ? stats_in + sidx * TSFA_STATS_N : nullptr);
        __syncthreads();
    }
}
#endif

#if !defined(TSFA_LONG)
// Bit-matrix sweep (fam_entropy_bits.h): every spec has m = 2 and every series of the launch 3 .. TSFA_ENTB_MAXN samples
// (shorter ones take the closed forms); workgroup = entb_waves_for(maxn, nspecs) wavefronts.
template <typename T, int QW_>
__global__ void __launch_bounds__(1024) k_entropy_bits(const T *__restrict__ values, const int64_t *__restrict__ starts, const int64_t *__restrict__ ends, int64_t n_series, const int *__restrict__ sel,
                          const TsfaSpec *__restrict__ specs, int nspecs, double *__restrict__ out, int64_t ld, int maxn,
                          unsigned short *__restrict__ perm_buf, int perm_stride, const double *__restrict__ stats_in) {
    TSFA_SERIES_BEGIN
    const int64_t off = starts[sidx];
    const int n = (int)(ends[sidx] - off);
    EntropyLds L;
    L.carve(tsfa_base, maxn, QW_ == TSFA_ENTB_QW ? 2 : 3);
    TSFA_TICKS_BEGIN();
    Blk b{(int)threadIdx.x, (int)blockDim.x, L.red, L.np};
    stage_series(b, values + off, n, L.xs);
    // tolerances per round: what EntropyLds sized the work region for (the longest series of the launch)
    const int kcap = (QW_ == TSFA_ENTB_QW) ? TSFA_ENTB_MAXK : entb_kround(maxn, TSFA_ENTB_MAXK, TSFA_ENTB_MAXWAVES);
    fam_entropy_series_bits<sizeof(T) == 4, QW_>(b, L.xs, n, specs, nspecs, out + sidx * ld, L.thr, L.perm, L.cnt,
                                            perm_buf ? perm_buf + (size_t)sidx * (size_t)perm_stride : nullptr, kcap,
                                            stats_in ? stats_in + sidx * TSFA_STATS_N : nullptr);
    TSFA_TICKS_END();
    TSFA_SERIES_END
}
#endif

// GROWS (LDS build only): the symbol rows in HBM -- rows + blockIdx.x * g.stride -- instead of LDS: the tables alone decide how
// many series a CU holds (8192 samples: 75 KB -> 40 KB, four series per CU instead of two; the family is latency bound)
template <typename T, bool GROWS>
__global__ void __launch_bounds__(256) k_seq(const T *__restrict__ values, const int64_t *__restrict__ starts, const int64_t *__restrict__ ends, int64_t n_series, const int *__restrict__ sel,
                      double *__restrict__ out, int64_t ld, const TsfaSeqGroup g, const double *__restrict__ stats_in,
                      unsigned char *__restrict__ rows TSFA_GS_PARAMS) {
    TSFA_SERIES_BEGIN
    const int64_t off = starts[sidx];
    const int n = (int)(ends[sidx] - off);
    SeqLds L;
    L.carve(tsfa_base, GROWS ? 0 : 1, g.stride, g.ttotal, g.etotal, (int)(blockDim.x + 63) >> 6);
    TSFA_TICKS_BEGIN();
    Blk b{(int)threadIdx.x, (int)blockDim.x, L.red, nullptr};
    const T *gv = values + off;
    fam_seq_series<GROWS>(b, [=](int i) { return (double)gv[i]; }, n, g, out + sidx * ld,
                          GROWS ? rows + (size_t)blockIdx.x * (size_t)g.stride : L.seq, L.tab, L.edges,
                          stats_in ? stats_in + sidx * TSFA_STATS_N : nullptr);
    TSFA_TICKS_END();
    TSFA_SERIES_END
}

// MAXT: 256 for series up to 2048 samples (register allocation for 4-wave workgroups), 1024 beyond
// MFMA: phase A on the float64 matrix cores (fam_cwt.h cwt_rows_mfma), the opt-in instantiation of TSFA_CWT_MFMA=1
template <typename T, int MAXT, bool MFMA>
__global__ void __launch_bounds__(MAXT, (MAXT == 256) ? 6 : 4) k_cwtpeaks(const T *__restrict__ values, const int64_t *__restrict__ starts, const int64_t *__restrict__ ends, int64_t n_series, const int *__restrict__ sel,
                           const TsfaSpec *__restrict__ specs, int nspecs, double *__restrict__ out, int64_t ld,
                           int maxn, int with_rowv, const double *__restrict__ consts TSFA_GS_PARAMS) {
    TSFA_SERIES_BEGIN
    const int64_t off = starts[sidx];
    const int n = (int)(ends[sidx] - off);
    CwtPeaksLayout L;
    L.carve(tsfa_base, maxn, with_rowv & 1, (int)sizeof(T));
    L.p.rk = consts ? consts + TSFA_CONSTS_RICKER : nullptr;
    TSFA_TICKS_BEGIN();
    Blk b{(int)threadIdx.x, (int)blockDim.x, L.p.red, nullptr};
    const T *g = values + off;
    fam_cwtpeaks_series<T, MFMA>(b, [=](int i) { return (double)g[i]; }, n, specs, nspecs, out + sidx * ld, L.p);
    TSFA_TICKS_END();
    TSFA_SERIES_END
}

#if !defined(TSFA_LONG)
// ---------------------------------------------------------------------------------------------
// Order statistics by selection: the SORT family of a plan that only asks for median / quantile columns (the
// streaming configurations, MinimalFCParameters: one median) does not need the sorted copy -- and with it the LDS
// residency, the barriers and 43 k cycles of latency per series.  One WAVEFRONT per series, the series in registers
// (E keys per lane, E = 16 at n = 1024) as order-preserving unsigned integers; the k-th smallest is found bit by bit
// from the top (#{key < prefix | bit} <= k decides the bit: E compares + one wavefront sum per bit), its successor by
// one more pass.  No LDS, no barrier, four series per 256-thread workgroup.  np.median / np.quantile("linear") then
// take their usual expressions on the (at most two) order statistics each needs.
// ---------------------------------------------------------------------------------------------
template <typename T> struct OsKey;
template <> struct OsKey<float> {
    typedef unsigned int key_t;
    static constexpr int BITS = 32;
    static __device__ __forceinline__ key_t enc(float v) {
        const unsigned int u = __float_as_uint(v);
        return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    }
    static __device__ __forceinline__ double dec(key_t k) {
        const unsigned int u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
        return (double)__uint_as_float(u);
    }
    static __device__ __forceinline__ key_t maxkey() { return 0xFFFFFFFFu; }
};
template <> struct OsKey<double> {
    typedef unsigned long long key_t;
    static constexpr int BITS = 64;
    static __device__ __forceinline__ key_t enc(double v) {
        const unsigned long long u = (unsigned long long)__double_as_longlong(v);
        return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
    }
    static __device__ __forceinline__ double dec(key_t k) {
        const unsigned long long u = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
        return __longlong_as_double((long long)u);
    }
    static __device__ __forceinline__ key_t maxkey() { return ~0ull; }
};

// wave64 integer sum without LDS-crossbar traffic: four DPP steps inside the 16-lane rows, then four readlanes
__device__ __forceinline__ int wave_sum_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, TSFA_DPP_QUAD_XOR1, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, TSFA_DPP_QUAD_XOR2, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, TSFA_DPP_ROW_HALF_MIRROR, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, TSFA_DPP_ROW_MIRROR, 0xf, 0xf, false);
    return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) +
           (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}

// k-th smallest (0-based) key of the wavefront's E x 64 keys
template <typename K, int E, int BITS>
__device__ __forceinline__ K os_select(const K (&key)[E], int k) {
    K prefix = 0;
#pragma unroll 1
    for (int bit = BITS - 1; bit >= 0; --bit) {
        const K cand = prefix | ((K)1 << bit);
        // the count of a wavefront-wide predicate is the popcount of its compare mask: one v_cmp per register, the
        // rest on the scalar ALU (no per-lane counters, no cross-lane sum)
        int c = 0;
#pragma unroll
        for (int e = 0; e < E; ++e) c += __popcll(__ballot(key[e] < cand));
        if (c <= k) prefix = cand;
    }
    return prefix;
}

template <typename T, int E>
__global__ void __launch_bounds__(256) k_order_stats(const T *__restrict__ values, const int64_t *__restrict__ starts, const int64_t *__restrict__ ends, int64_t n_series,
                                                     const int *__restrict__ sel, const TsfaSpec *__restrict__ specs, int nspecs,
                                                     double *__restrict__ out, int64_t ld) {
    typedef OsKey<T> KC;
    typedef typename KC::key_t K;
    const int lane = threadIdx.x & 63;
    const int64_t wi = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wi >= n_series) return;  // wave-uniform
    const int64_t sidx = sel ? (int64_t)sel[wi] : wi;
    const int64_t off = starts[sidx];
    const int n = (int)(ends[sidx] - off);
    const T *__restrict__ g = values + off;
    K key[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = e * 64 + lane;
        key[e] = (i < n) ? KC::enc(g[i]) : KC::maxkey();  // pads sort behind every sample (NaN-free input)
    }
    double *row = out + sidx * ld;
    int have_k = -1;
    double v_k = 0.0, v_k1 = 0.0;  // cached order statistics k and k + 1
    // order statistic i of the series (wave-collective); np.median / np.quantile read i and i + 1 back to back
    auto os = [&](int i) -> double {
        if (i == have_k) return v_k;
        if (i == have_k + 1 && have_k >= 0) return v_k1;
        const K kk = os_select<K, E, KC::BITS>(key, i);
        v_k = KC::dec(kk);
        // successor: the (i + 1)-th smallest is v_k again if v_k occurs often enough, else the smallest larger key
        int cle = 0;
        K nxt = KC::maxkey();
#pragma unroll
        for (int e = 0; e < E; ++e) {
            cle += __popcll(__ballot(key[e] <= kk));
            if (key[e] > kk && key[e] < nxt) nxt = key[e];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const K t = (K)__shfl_xor(nxt, o);
            nxt = (t < nxt) ? t : nxt;
        }
        v_k1 = (cle >= i + 2 || i + 1 >= n) ? v_k : KC::dec(nxt);
        have_k = i;
        return v_k;
    };
    for (int s = 0; s < nspecs; ++s) {
        const TsfaSpec sp = specs[s];
        double res;
        if (sp.calc == TSFA_C_MEDIAN) {                                      // fc.py:663 np.median
            if (n & 1) res = os((n - 1) / 2);
            else { const double a0 = os(n / 2 - 1), a1 = os(n / 2); res = (0.0 + a0 + a1) / 2.0; }
        } else {                                                             // fc.py:1963 np.quantile (TSFA_C_QUANTILE)
            res = np_quantile_sorted(os, n, sp.p[0]);
        }
        if (lane == 0) row[sp.col] = res;
    }
}

// ---------------------------------------------------------------------------------------------
// Streaming plans (MinimalFCParameters): ONE read of the samples.  A plan whose BASIC columns are float closed forms of
// the per-series statistics (sum / mean / length / std / variance / rms / max / |max| / min / abs_energy / variation
// coefficient) and whose SORT columns are `median` only runs entirely from registers: one wavefront per series, the
// samples coalesced into E registers per lane (E * 64 >= n), statistics by in-register sums + wave butterflies, the
// median by WINDOW SELECTION, the row written by lane = column.  No LDS staging of the series, no barrier.
//   * Window selection: the two middle order statistics lie near the mean; count the samples below lo and up to hi for
//     [lo, hi] = mean -+ 0.06 std (counts are popcounts of compare masks: 2 v_cmp per register, the rest on the scalar
//     ALU).  If both ranks fall inside and the window holds <= 64 samples they are compacted to one per lane (LDS, 256
//     B per wavefront) and sorted across the lanes; a window that misses is slid towards the ranks, a crowded one is
//     halved, and after four tries the exact bit-by-bit selection of k_order_stats takes over (ties, heavy tails).
//     Measured on 100 000 x 1024 float32 (profiles/lab/minimal_fused_lab.hip): load + statistics 0.067 ms (6.1 TB/s),
//     + bit-by-bit selection 0.25 ms, + window selection 0.14 ms with 46 series falling back.
//   * The sums are plain tree sums, not numpy's pairwise order (fam_basic.h reproduces that order from an LDS copy, which
//     is what k_basic_lite spent its time on): none of the columns served here is compared against another quantity,
//     so the 1e-6 bar holds with margin (differences ~1e-16 relative).
// Replaces k_basic_lite + k_order_stats (0.23 + 0.25 ms, two reads of the samples).
// ---------------------------------------------------------------------------------------------
// FULL: the series fills all E * 64 register slots (n == E * 64: no pads to mask out of the sums)
template <typename T, int E, bool FULL>
__device__ __forceinline__ void stream_body(const T (&v)[E], int n, int lane, typename OsKey<T>::key_t *w,
                                            const TsfaSpec *__restrict__ bspecs, int nb, const TsfaSpec *__restrict__ sspecs, int ns,
                                            int want_median, double *__restrict__ row) {
#pragma clang fp contract(fast)   // nothing here is compared against another quantity: fused multiply-adds are fine
    typedef OsKey<T> KC;
    typedef typename KC::key_t K;
    const double dn = (double)n;
    double s = 0.0;
    T tmn = (T)TSFA_INF, tmx = (T)-TSFA_INF;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const bool in = FULL || (e * 64 + lane < n);
        s += in ? (double)v[e] : 0.0;
        tmn = in ? (v[e] < tmn ? v[e] : tmn) : tmn;
        tmx = in ? (v[e] > tmx ? v[e] : tmx) : tmx;
    }
    s = wave_sum(s);
    const double mean = s / dn;
    double ssd = 0.0, sq = 0.0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const bool in = FULL || (e * 64 + lane < n);
        const double x = (double)v[e], d = x - mean;
        ssd += in ? d * d : 0.0;
        sq += in ? x * x : 0.0;
    }
    ssd = wave_sum(ssd);
    sq = wave_sum(sq);
    const double mn = wave_min((double)tmn), mx = wave_max((double)tmx);
    const double var = ssd / dn, sd = sqrt(var);

    double med = TSFA_NAN;
    if (want_median) {
        // order statistics k and k1 (k1 == k for odd n)
        const int k = (n - 1) / 2, k1 = n / 2;
        T lo = (T)(mean - 0.06 * sd), hi = (T)(mean + 0.06 * sd);
        int c_lo = 0, c_hi = 0;   // #{v < lo}, #{v <= hi}
        bool ok = false;
        if (sd > 0.0 && sd < TSFA_INF) {
#pragma unroll 1
            for (int it = 0; it < 4; ++it) {
                if (!((double)hi < TSFA_INF) || !((double)lo > -TSFA_INF)) break;   // (the +inf pads must stay outside the window)
                c_lo = 0; c_hi = 0;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    c_lo += __popcll(__ballot(v[e] < lo));
                    c_hi += __popcll(__ballot(v[e] <= hi));
                }
                if (k >= c_lo && k1 < c_hi) {
                    if (c_hi - c_lo <= 64) { ok = true; break; }
                    // crowded: keep the half that holds both ranks (a rank pair straddling the cut gives up)
                    const T mid = (T)(0.5 * ((double)lo + (double)hi));
                    int c_mid = 0;
#pragma unroll
                    for (int e = 0; e < E; ++e) c_mid += __popcll(__ballot(v[e] <= mid));
                    if (k1 < c_mid) hi = mid;
                    else if (k >= c_mid)   // both ranks above mid: the window restarts at the next value after it
                        lo = (sizeof(T) == 4) ? (T)nextafterf((float)mid, INFINITY) : (T)nextafter((double)mid, (double)INFINITY);
                    else break;
                } else {
                    const double wd = (double)hi - (double)lo;   // slide towards the ranks, overlapping the old window
                    if (k < c_lo) { hi = (T)((double)lo + 0.5 * wd); lo = (T)((double)hi - 2.0 * wd); }
                    else { lo = (T)((double)hi - 0.5 * wd); hi = (T)((double)lo + 2.0 * wd); }
                }
            }
        }
        if (ok) {
            int base = 0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const unsigned long long m = __ballot(v[e] >= lo && v[e] <= hi);
                const bool in = (m >> lane) & 1ull;
                const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
                if (in) w[pos] = KC::enc(v[e]);
                base += __popcll(m);
            }
            __builtin_amdgcn_wave_barrier();
            K key = (lane < c_hi - c_lo) ? w[lane] : KC::maxkey();
#pragma unroll
            for (int kk = 2; kk <= 64; kk <<= 1) {
#pragma unroll
                for (int j = kk >> 1; j > 0; j >>= 1) {
                    const K o = (K)__shfl_xor(key, j);
                    const bool up = ((lane & kk) == 0), lower = ((lane & j) == 0);
                    const K a = key < o ? key : o, c = key < o ? o : key;
                    key = (lower == up) ? a : c;
                }
            }
            const K q0 = (K)__shfl(key, k - c_lo), q1 = (K)__shfl(key, k1 - c_lo);
            med = (n & 1) ? KC::dec(q0) : (0.0 + KC::dec(q0) + KC::dec(q1)) / 2.0;   // np.median
        } else {
            K key[E];
#pragma unroll
            for (int e = 0; e < E; ++e) key[e] = (FULL || e * 64 + lane < n) ? KC::enc(v[e]) : KC::maxkey();
            const K q0 = os_select<K, E, KC::BITS>(key, k);
            double a0 = KC::dec(q0), a1 = a0;
            if (k1 != k) {
                int cle = 0;
                K nxt = KC::maxkey();
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    cle += __popcll(__ballot(key[e] <= q0));
                    if (key[e] > q0 && key[e] < nxt) nxt = key[e];
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const K t = (K)__shfl_xor(nxt, o);
                    nxt = (t < nxt) ? t : nxt;
                }
                a1 = (cle >= k1 + 1) ? a0 : KC::dec(nxt);
            }
            med = (n & 1) ? a0 : (0.0 + a0 + a1) / 2.0;
        }
    }
    // lane = column
    for (int c = lane; c < nb + ns; c += 64) {
        const TsfaSpec sp = (c < nb) ? bspecs[c] : sspecs[c - nb];
        double r = TSFA_NAN;
        switch (sp.calc) {
        case TSFA_C_SUM_VALUES: r = s; break;
        case TSFA_C_MEAN: r = mean; break;
        case TSFA_C_LENGTH: r = dn; break;
        case TSFA_C_STANDARD_DEVIATION: r = sd; break;
        case TSFA_C_VARIANCE: r = var; break;
        case TSFA_C_ROOT_MEAN_SQUARE: r = sqrt(sq / dn); break;
        case TSFA_C_MAXIMUM: r = mx; break;
        case TSFA_C_ABSOLUTE_MAXIMUM: r = fmax(fabs(mx), fabs(mn)); break;
        case TSFA_C_MINIMUM: r = mn; break;
        case TSFA_C_ABS_ENERGY: r = sq; break;
        case TSFA_C_VARIATION_COEFFICIENT: r = (mean == 0.0) ? TSFA_NAN : sd / mean; break;
        case TSFA_C_MEDIAN: r = med; break;
        default: break;
        }
        row[sp.col] = r;
    }
}

template <typename T, int E>
__global__ void __launch_bounds__(256) k_stream(const T *__restrict__ values, const int64_t *__restrict__ starts, const int64_t *__restrict__ ends, int64_t n_series,
                                                const int *__restrict__ sel, const TsfaSpec *__restrict__ bspecs, int nb,
                                                const TsfaSpec *__restrict__ sspecs, int ns, int want_median,
                                                double *__restrict__ out, int64_t ld) {
    typedef OsKey<T> KC;
    typedef typename KC::key_t K;
    __shared__ K win[4][64];
    const int lane = threadIdx.x & 63;
    const int64_t wi = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wi >= n_series) return;  // wave-uniform
    const int64_t sidx = sel ? (int64_t)sel[wi] : wi;
    const int64_t off = starts[sidx];
    const int n = (int)(ends[sidx] - off);
    const T *__restrict__ g = values + off;
    T v[E];
    K *w = win[threadIdx.x >> 6];
    double *row = out + sidx * ld;
    if (n == E * 64) {
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = g[e * 64 + lane];
        stream_body<T, E, true>(v, n, lane, w, bspecs, nb, sspecs, ns, want_median, row);
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = e * 64 + lane;
            v[e] = (i < n) ? g[i] : (T)TSFA_INF;   // pads sort behind every sample and are masked out of the sums
        }
        stream_body<T, E, false>(v, n, lane, w, bspecs, nb, sspecs, ns, want_median, row);
    }
}

// ---------------------------------------------------------------------------------------------
// cwt_coefficients (fc.py:1370): pywt.cwt(x, widths, "mexh")[i, coeff] only ever reads output positions
// coeff < ~15, i.e. a dot product of the first S samples with a fixed filter column.  For a batch that is the
// dense contraction  X[n_series x S] . W[S x C]  -> float64 MFMA (v_mfma_f64_16x16x4_f64), one wavefront per
// 16 series, C padded to a multiple of 16, S padded to a multiple of 4.
//   W is stored [Cpad][S4] (column c contiguous in k).
// ---------------------------------------------------------------------------------------------
typedef double tsfa_d4 __attribute__((ext_vector_type(4)));

// Workgroup = 4 wavefronts = 64 series (one 16-series row tile per wavefront); K in chunks of 16: the sample chunk
// [64 x 16] (coalesced 64-byte rows, converted to float64 on the way) and the filter chunk [16 x 16 CT] are staged in LDS,
// double buffered, and every wavefront feeds 4 CT MFMAs per chunk from 4 + 4 CT conflict-free ds_read_b64.  (Rounds 1-3:
// one wavefront per 16 series reading A and B straight from global memory, one 4- or 8-byte load per lane and MFMA
// operand: matrix pipe 13.6 % busy.)
#define TSFA_CWT_KC 16
template <typename T, int CT>
__global__ void __launch_bounds__(256)
k_cwt_gemm(const T *__restrict__ values, const int64_t *__restrict__ starts, const int64_t *__restrict__ ends, int64_t n_series,
           const double *__restrict__ W, int S4, int C, const int *__restrict__ cols,
           const int *__restrict__ coeff_idx, double *__restrict__ out, int64_t ld) {
    constexpr int KC = TSFA_CWT_KC, NC = CT * 16;
    __shared__ double sA[2][64][KC + 1];
    __shared__ double sB[2][KC][NC + 2];
    __shared__ int lens[64];
    __shared__ long long offs[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const int64_t base = (int64_t)blockIdx.x * 64;
    if (tid < 64) {
        const int64_t s = base + tid;
        long long off = 0;
        int len = 0;
        if (s < n_series) {
            off = starts[s];
            len = (int)(ends[s] - off);
        }
        lens[tid] = len;
        offs[tid] = off;
    }
    __syncthreads();
    // a thread's share of a chunk: 4 samples (A: 16 consecutive samples of a series per 16 threads) and CT filter taps (B: W
    // is [Cpad][S4], column c contiguous in k); fetched into registers BEFORE the MFMAs of the current chunk and written to
    // the other LDS buffer AFTER them, so the global-memory latency hides behind the matrix pipe
    double ra[4], rb[CT];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + u * 256, m = i / KC, kk = i - m * KC, k = k0 + kk;
            ra[u] = (k < lens[m]) ? (double)values[offs[m] + k] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < CT; ++u) {
            const int i = tid + u * 256, c = i / KC, kk = i - c * KC, k = k0 + kk;
            rb[u] = (k < S4) ? W[(size_t)c * S4 + k] : 0.0;
        }
    };
    auto put = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + u * 256, m = i / KC, kk = i - m * KC;
            sA[buf][m][kk] = ra[u];
        }
#pragma unroll
        for (int u = 0; u < CT; ++u) {
            const int i = tid + u * 256, c = i / KC, kk = i - c * KC;
            sB[buf][kk][c] = rb[u];
        }
    };
    tsfa_d4 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[ct] = (tsfa_d4){0.0, 0.0, 0.0, 0.0};
    const int nchunks = (S4 + KC - 1) / KC;
    fetch(0);
    put(0);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunks) fetch((ch + 1) * KC);
#pragma unroll
        for (int ks = 0; ks < KC; ks += 4) {
            const double a = sA[buf][wave * 16 + r][ks + kq];         // A[i = r][k = kq]
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const double bv = sB[buf][ks + kq][ct * 16 + r];      // B[k = kq][j = r]
                acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc[ct], 0, 0, 0);
            }
        }
        if (ch + 1 < nchunks) put(buf ^ 1);
        __syncthreads();
    }
    // D[i = 4*v + kq][j = r]: the f64 16x16x4 accumulator interleaves rows across the four 16-lane groups
    // (composable_kernel xdlops_gemm.hpp mfma_f64_16x16x4f64: group_size 1, 4 groups), unlike the f32 shapes
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int c = ct * 16 + r;
        if (c >= C) continue;
        const int col = cols[c], ci = coeff_idx[c];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int i = wave * 16 + 4 * v + kq;
            const int64_t srow = base + i;
            if (srow < n_series) out[srow * ld + col] = (ci < lens[i]) ? acc[ct][v] : TSFA_NAN;
        }
    }
}

// the plan's n_cols columns of every row (leading dimension ld >= n_cols: the cells beyond belong to the caller)
__global__ void k_fill_nan(double *__restrict__ out, int64_t n_rows, int64_t n_cols, int64_t ld, double value) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (ld == n_cols) {
        for (int64_t k = i; k < n_rows * n_cols; k += stride) out[k] = value;
    } else {
        for (int64_t k = i; k < n_rows * n_cols; k += stride) {
            const int64_t r = k / n_cols;
            out[r * ld + (k - r * n_cols)] = value;
        }
    }
}
// the listed columns only (a plan whose kernels write every other cell themselves)
__global__ void k_fill_cols(double *__restrict__ out, int64_t n_rows, const int *__restrict__ cols, int n_fill, int64_t ld, double value) {
    const int64_t total = n_rows * (int64_t)n_fill;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = k / n_fill;
        out[r * ld + cols[k - r * n_fill]] = value;
    }
}

// per-batch length statistics: [0] = max length, [1] = min length, [2] = max non-power-of-two length, then per
// length class c (tsfa_len_class: lengths in (2^(c+5), 2^(c+6)], the first class from 1):
//   [3 + c] = number of series, [3 + NC + c] = longest, [3 + 2 NC + c] = longest non-power-of-two
__device__ __forceinline__ int tsfa_len_class_dev(long long l) {
    int c = 0;
    while (c < TSFA_N_LEN_CLASSES - 1 && l > (64LL << c)) ++c;
    return c;
}
__global__ void __launch_bounds__(256) k_len_stats(const int64_t *__restrict__ starts, const int64_t *__restrict__ ends, int64_t n_series, long long *__restrict__ stats) {
    __shared__ long long acc[TSFA_LEN_STATS];
    for (int i = threadIdx.x; i < TSFA_LEN_STATS; i += blockDim.x) acc[i] = (i == 1) ? (1LL << 62) : 0;
    __syncthreads();
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_series;
         s += (int64_t)gridDim.x * blockDim.x) {
        const long long l = ends[s] - starts[s];
        const bool np2 = l > 0 && (l & (l - 1)) != 0;
        const int c = tsfa_len_class_dev(l);
        atomicMax(&acc[0], l);
        atomicMin(&acc[1], l);
        if (np2) atomicMax(&acc[2], l);
        atomicAdd((unsigned long long *)&acc[3 + c], 1ULL);
        atomicMax(&acc[3 + TSFA_N_LEN_CLASSES + c], l);
        if (np2) atomicMax(&acc[3 + 2 * TSFA_N_LEN_CLASSES + c], l);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TSFA_LEN_STATS; i += blockDim.x) {
        if (i == 1) atomicMin(&stats[1], acc[1]);
        else if (i >= 3 && i < 3 + TSFA_N_LEN_CLASSES) atomicAdd((unsigned long long *)&stats[i], (unsigned long long)acc[i]);
        else atomicMax(&stats[i], acc[i]);
    }
}

// series index lists of the launch groups: sel[g.base[group] ...] receives the series whose length class maps to it
__global__ void __launch_bounds__(256) k_class_fill(const int64_t *__restrict__ starts, const int64_t *__restrict__ ends, int64_t n_series,
                                                    const TsfaClassMap g, int *__restrict__ cursor, int *__restrict__ sel) {
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_series;
         s += (int64_t)gridDim.x * blockDim.x) {
        const int grp = g.group_of[tsfa_len_class_dev(ends[s] - starts[s])];
        sel[g.base[grp] + atomicAdd(&cursor[grp], 1)] = (int)s;
    }
}

#endif  // !TSFA_LONG

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
#define TSFA_LAUNCH_CHECK()                       \
    do {                                          \
        hipError_t e_ = hipGetLastError();        \
        if (e_ != hipSuccess) return (int)e_;     \
    } while (0)

template <typename K>
static int set_lds(K kern, size_t bytes) {
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

// One family launch.  LDS build: a workgroup per series, `lds` bytes of dynamic LDS.  TSFA_LONG build: a persistent grid
// of as many workgroups as the scratch holds slots of `lds` bytes (a.long_scratch / a.long_bytes), no dynamic LDS.
#if defined(TSFA_LONG)
#define TSFA_KLAUNCH(kern, lds, ...)                                                                       \
    do {                                                                                                   \
        const size_t slot_ = ((size_t)(lds) + 255) & ~(size_t)255;                                          \
        const int64_t slots_ = (int64_t)(a.long_bytes / slot_);                                            \
        if (!a.long_scratch || slots_ < 1) return -2;                                                      \
        const dim3 lgrid_((unsigned)std::min<int64_t>(a.n_series, std::min<int64_t>(slots_, 2048)));       \
        kern<<<lgrid_, nt, 0, st>>>(__VA_ARGS__, a.long_scratch, slot_);                                   \
    } while (0)
#else
#define TSFA_KLAUNCH(kern, lds, ...)                           \
    do {                                                       \
        if ((rc = set_lds(kern, lds))) return rc;              \
        kern<<<grid, nt, lds, st>>>(__VA_ARGS__);              \
    } while (0)
#endif

template <typename T>
static int launch_all_t(const TsfaLaunch &a, const T *values) {
    hipStream_t st = (hipStream_t)a.stream;
    const dim3 grid((unsigned)a.n_series);
    const int nt = a.nt;
    int rc = 0;
    (void)grid; (void)rc;
    if (a.fam == TSFA_FAM_BASIC) {
        BasicLds L;
        const size_t lds = L.carve(nullptr, a.maxn, nt, (int)sizeof(T), 1);
        if (a.hint_c == 0 && nt <= 256) {  // no column of the loop / count / sum kinds: statistics + epilogue only
#if defined(TSFA_LONG)
            TSFA_KLAUNCH(k_basic_lite<T>, lds, values, a.starts, a.ends, a.n_series, a.sel, a.specs, a.nspecs, a.out, a.ld,
                         a.dectab, a.maxn, a.hint_a);
#else
            if ((rc = set_lds(k_basic_lite<T>, lds))) return rc;
            const int64_t resident = std::max<int64_t>(1, std::min<int64_t>(32, (int64_t)(TSFA_LDS_LIMIT / std::max<size_t>(lds, 1))));
            const dim3 pgrid((unsigned)std::min<int64_t>(a.n_series, 256 * resident));
            k_basic_lite<T><<<pgrid, nt, lds, st>>>(values, a.starts, a.ends, a.n_series, a.sel, a.specs, a.nspecs, a.out, a.ld,
                                                     a.dectab, a.maxn, a.hint_a);
#endif
        } else {
            TSFA_KLAUNCH(k_basic<T>, lds, values, a.starts, a.ends, a.n_series, a.sel, a.specs, a.nspecs, a.out, a.ld, a.dectab,
                         a.maxn, a.hint_a, a.hint_c, a.hint_d, a.hint_e, a.stats_out, a.skip_le);
        }
    } else if (a.fam == TSFA_FAM_TREND) {
        BasicLds L;
        const size_t lds = L.carve(nullptr, a.maxn, nt, (int)sizeof(T), 2, a.alt.small_w);
        TSFA_KLAUNCH(k_trend<T>, lds, values, a.starts, a.ends, a.n_series, a.sel, a.specs, a.nspecs, a.out, a.ld, a.maxn,
                     a.hint_b, a.times, a.alt, a.hint_c, a.skip_le);
    } else if (a.fam == TSFA_FAM_SORT) {
        SortLds L;
        const int wd = (a.hint_a >= 320) ? a.hint_a : 1280;
        const size_t lds = L.carve(nullptr, a.maxn, nt, (int)sizeof(T), wd);
        TSFA_KLAUNCH(k_sort<T>, lds, values, a.starts, a.ends, a.n_series, a.sel, a.specs, a.nspecs, a.out, a.ld, a.maxn, a.cq,
                     a.hint_c, wd, a.pf_buf, a.pf_count, a.pf_slot, a.pf_cap, a.perm_buf, a.perm_stride, a.stats_in, a.hint_d);
    } else if (a.fam == TSFA_FAM_SPECTRAL) {
        SpectralLds L;
        const size_t lds = L.carve(nullptr, a.maxn, a.dft_n, (int)sizeof(T));
        if (a.gscratch != nullptr) {
            auto kfn = k_spectral<T, true>;
#if defined(TSFA_LONG)
            // (a persistent grid of at most 2048 workgroups: the plan allocated a scratch slot for each)
            TSFA_KLAUNCH(kfn, lds, values, a.starts, a.ends, a.n_series, a.sel, a.specs, a.nspecs, a.out, a.ld, a.maxn,
                         a.dft_n, a.gscratch, a.gscratch_n, a.bluestein_min, a.twc, a.tws, a.hint_a, a.hint_b, a.consts);
#else
            // one workgroup per series, one scratch slot per workgroup: the group goes out in launches of as many series as
            // the plan's scratch has slots (configs[4] per GPU: 12 500 series x 512 KB would be 6.4 GB in one launch)
            if ((rc = set_lds(kfn, lds))) return rc;
            const int64_t slots = std::max<int64_t>(a.gscratch_slots, 1);
            for (int64_t c0 = 0; c0 < a.n_series; c0 += slots) {
                const int64_t nc = std::min<int64_t>(slots, a.n_series - c0);
                const int64_t s0 = a.sel ? 0 : c0;   // without a series list the workgroup index IS the series
                kfn<<<dim3((unsigned)nc), nt, lds, st>>>(values, a.starts + s0, a.ends + s0, nc, a.sel ? a.sel + c0 : nullptr, a.specs,
                                                        a.nspecs, a.out + s0 * a.ld, a.ld, a.maxn, a.dft_n, a.gscratch,
                                                        a.gscratch_n, a.bluestein_min, a.twc, a.tws, a.hint_a, a.hint_b, a.consts);
            }
#endif
        } else {
            auto kfn = k_spectral<T, false>;
            TSFA_KLAUNCH(kfn, lds, values, a.starts, a.ends, a.n_series, a.sel, a.specs, a.nspecs, a.out, a.ld, a.maxn,
                         a.dft_n, a.gscratch, a.gscratch_n, a.bluestein_min, a.twc, a.tws, a.hint_a, a.hint_b, a.consts);
        }
    } else if (a.fam == TSFA_FAM_AR) {
        ArLds L;
        const size_t lds = L.carve(nullptr, a.maxn, a.ar_P, (int)sizeof(T));
        TSFA_KLAUNCH(k_ar<T>, lds, values, a.starts, a.ends, a.n_series, a.sel, a.specs, a.nspecs, a.out, a.ld, a.maxn, a.ar_P,
                     a.hint_a, a.hint_b, a.hint_c, a.hint_d, a.deg_list, a.deg_count, a.stats_in);
    } else if (a.fam == TSFA_FAM_ENTROPY) {
#if defined(TSFA_LONG)
        if (a.ent_cnt == 4) {   // fam_entropy_hbits.h: the table in LDS, everything per sample in the workgroup's HBM slot
            EntropyHugeSlot H;
            const size_t slot_ = (H.carve(nullptr, a.maxn) + 255) & ~(size_t)255;
            const int64_t slots_ = (int64_t)(a.long_bytes / slot_);
            if (!a.long_scratch || slots_ < 1) return -2;
            const size_t hl = entropy_huge_lds_bytes(a.maxn);
            auto kfn = kl_entropy_hbits<T>;
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)hl);
            if (e != hipSuccess) return (int)e;
            const dim3 hgrid((unsigned)std::min<int64_t>(a.n_series, std::min<int64_t>(slots_, 512)));
            kfn<<<hgrid, 1024, hl, st>>>(values, a.starts, a.ends, a.n_series, a.sel, a.specs, a.nspecs, a.out, a.ld, a.maxn, a.stats_in,
                                         a.long_scratch, slot_);
            TSFA_LAUNCH_CHECK();
            return 0;
        }
#endif
        EntropyLds L;
#if defined(TSFA_LONG)
        const size_t lds = L.carve(nullptr, a.maxn, a.ent_cnt, 4);   // (32-bit sample order: k_entropy above)
#else
        const size_t lds = L.carve(nullptr, a.maxn, a.ent_cnt);
#endif
#if !defined(TSFA_LONG)
        if (a.ent_cnt == 3) {
            auto kfn = k_entropy_bits<T, TSFA_ENTB_QW_LONG>;
            TSFA_KLAUNCH(kfn, lds, values, a.starts, a.ends, a.n_series, a.sel, a.specs, a.nspecs, a.out, a.ld, a.maxn,
                         (unsigned short *)nullptr, 0, a.stats_in);
        } else if (a.ent_cnt == 2) {
            auto kfn = k_entropy_bits<T, TSFA_ENTB_QW>;
            TSFA_KLAUNCH(kfn, lds, values, a.starts, a.ends, a.n_series, a.sel, a.specs, a.nspecs, a.out, a.ld, a.maxn,
                         a.perm_buf, a.perm_stride, a.stats_in);
        } else
#endif
        if (a.ent_fast) {
            auto kfn = k_entropy<T, true>;
            TSFA_KLAUNCH(kfn, lds, values, a.starts, a.ends, a.n_series, a.sel, a.specs, a.nspecs, a.out, a.ld, a.maxn, a.ent_cnt, a.stats_in);
        } else {
            auto kfn = k_entropy<T, false>;
            TSFA_KLAUNCH(kfn, lds, values, a.starts, a.ends, a.n_series, a.sel, a.specs, a.nspecs, a.out, a.ld, a.maxn, a.ent_cnt, a.stats_in);
        }
    } else if (a.fam == TSFA_FAM_SEQ) {
        SeqLds L;
        const size_t lds = L.carve(nullptr, a.seq.grows ? 0 : 1, a.seq.stride, a.seq.ttotal, a.seq.etotal, (nt + 63) >> 6);
#if !defined(TSFA_LONG)
        if (a.seq.grows) {
            if (!a.seq_rows) return -2;
            auto kfn = k_seq<T, true>;
            TSFA_KLAUNCH(kfn, lds, values, a.starts, a.ends, a.n_series, a.sel, a.out, a.ld, a.seq, a.stats_in, a.seq_rows);
        } else
#endif
        {
            auto kfn = k_seq<T, false>;
            TSFA_KLAUNCH(kfn, lds, values, a.starts, a.ends, a.n_series, a.sel, a.out, a.ld, a.seq, a.stats_in, (unsigned char *)nullptr);
        }
    } else if (a.fam == TSFA_FAM_CWT) {  // number_cwt_peaks
        CwtPeaksLayout L;
        const size_t lds = L.carve(nullptr, a.maxn, a.cwt_rowv & 1, (int)sizeof(T));
#if !defined(TSFA_LONG)
        if (a.cwt_rowv & 2) {   // TSFA_CWT_MFMA=1
            if (nt <= 256) {
                auto kfn = k_cwtpeaks<T, 256, true>;
                TSFA_KLAUNCH(kfn, lds, values, a.starts, a.ends, a.n_series, a.sel, a.specs, a.nspecs, a.out, a.ld, a.maxn, a.cwt_rowv, a.consts);
            } else {
                auto kfn = k_cwtpeaks<T, 1024, true>;
                TSFA_KLAUNCH(kfn, lds, values, a.starts, a.ends, a.n_series, a.sel, a.specs, a.nspecs, a.out, a.ld, a.maxn, a.cwt_rowv, a.consts);
            }
        } else
#endif
        if (nt <= 256) {
            auto kfn = k_cwtpeaks<T, 256, false>;
            TSFA_KLAUNCH(kfn, lds, values, a.starts, a.ends, a.n_series, a.sel, a.specs, a.nspecs, a.out, a.ld, a.maxn, a.cwt_rowv, a.consts);
        } else {
            auto kfn = k_cwtpeaks<T, 1024, false>;
            TSFA_KLAUNCH(kfn, lds, values, a.starts, a.ends, a.n_series, a.sel, a.specs, a.nspecs, a.out, a.ld, a.maxn, a.cwt_rowv, a.consts);
        }
    } else {
        return -1;
    }
    TSFA_LAUNCH_CHECK();
    return 0;
}

#if !defined(TSFA_LONG)
template <typename T>
static int launch_rows_t(const TsfaLaunch &a, const T *values) {
    hipStream_t st = (hipStream_t)a.stream;
    const int per = 64 / TSFA_ROW_LANES;
    const dim3 grid((unsigned)((a.n_series + per - 1) / per));
    const int maxn = std::min(a.maxn, TSFA_ROW_MAXN);
    int rc = 0;
    if (a.fam == TSFA_FAM_BASIC) {
        const size_t rb = BasicRowLds::row_bytes(maxn, (int)sizeof(T), 1, 0);
        if ((rc = set_lds(k_basic_rows<T>, rb * per))) return rc;
        k_basic_rows<T><<<grid, 64, rb * per, st>>>(values, a.starts, a.ends, a.n_series, a.sel, a.specs, a.nspecs, a.out, a.ld, a.dectab,
                                                    maxn, a.hint_a, a.hint_c, a.hint_d, a.hint_e, a.stats_out, (int)rb);
    } else if (a.fam == TSFA_FAM_TREND) {
        const size_t rb = BasicRowLds::row_bytes(maxn, (int)sizeof(T), 2, a.alt.small_w);
        if ((rc = set_lds(k_trend_rows<T>, rb * per))) return rc;
        k_trend_rows<T><<<grid, 64, rb * per, st>>>(values, a.starts, a.ends, a.n_series, a.sel, a.specs, a.nspecs, a.out, a.ld, maxn,
                                                    a.hint_b, a.times, a.alt, a.hint_c, (int)rb);
    } else {
        return -1;
    }
    TSFA_LAUNCH_CHECK();
    return 0;
}
int tsfa_launch_rows(const TsfaLaunch &a) {
    if (a.dtype == 0) return launch_rows_t<float>(a, (const float *)a.values);
    return launch_rows_t<double>(a, (const double *)a.values);
}
#endif

#if defined(TSFA_LONG)
int tsfa_launch_family_long(const TsfaLaunch &a) {
    if (a.dtype == 0) return launch_all_t<float>(a, (const float *)a.values);
    return launch_all_t<double>(a, (const double *)a.values);
}

// k_general (fam_general.h): the calculators of a plan whose parameters lie beyond the tuned kernels' tables.  One wavefront
// per series on a persistent grid, the series read where it lies, every working array in the workgroup's slot of HBM scratch
// (this translation unit's blk_sync orders global memory).
template <typename T>
__global__ void __launch_bounds__(64) k_general(const T *__restrict__ values, const int64_t *__restrict__ starts, const int64_t *__restrict__ ends,
                                               int64_t n_series, const TsfaSpec *__restrict__ specs, int nspecs, double *__restrict__ out, int64_t ld,
                                               int maxn, const TsfaGenPlan g, double *__restrict__ scratch, size_t slot_doubles,
                                               const double *__restrict__ pool) {
    __shared__ double red[TSFA_RED_DOUBLES];
    __shared__ NpScratch nps;
    Blk b{(int)threadIdx.x, (int)blockDim.x, red, &nps};
    GenSlot S;
    S.carve(scratch + (size_t)blockIdx.x * slot_doubles, maxn, g);
    for (int64_t sidx = blockIdx.x; sidx < n_series; sidx += gridDim.x) {
        const int64_t off = starts[sidx];
        const int n = (int)(ends[sidx] - off);
        const T *gv = values + off;
        fam_general_series(b, [=](int i) { return (double)gv[i]; }, n, specs, nspecs, out + sidx * ld, S, g, pool);
        __syncthreads();
    }
}

size_t tsfa_general_slot_doubles(int maxn, const TsfaGenPlan &g) {
    GenSlot S;
    return S.carve(nullptr, maxn, g);
}

int tsfa_launch_general(const TsfaLaunch &a, const TsfaGenPlan &g, double *scratch, size_t slot_doubles, int slots, const double *pool) {
    hipStream_t st = (hipStream_t)a.stream;
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(a.n_series, slots));
    if (a.dtype == 0)
        k_general<float><<<grid, 64, 0, st>>>((const float *)a.values, a.starts, a.ends, a.n_series, a.specs, a.nspecs, a.out, a.ld, a.maxn, g,
                                              scratch, slot_doubles, pool);
    else
        k_general<double><<<grid, 64, 0, st>>>((const double *)a.values, a.starts, a.ends, a.n_series, a.specs, a.nspecs, a.out, a.ld, a.maxn, g,
                                               scratch, slot_doubles, pool);
    TSFA_LAUNCH_CHECK();
    return 0;
}
#else
// second pass of the AR family over the series k_ar listed (after tsfa_launch_family / _long of TSFA_FAM_AR)
template <typename T>
static int launch_ar_degenerate_t(const TsfaLaunch &a, const T *values) {
    hipStream_t st = (hipStream_t)a.stream;
    int rc;
    ArDdLds D;
    const int P = a.ar_P_dd > 0 ? a.ar_P_dd : a.ar_P;
    size_t dlds = D.carve(nullptr, P);
    unsigned dgrid = (unsigned)std::min<int64_t>(a.n_series, 4096);
    if (a.dd_scratch != nullptr) {   // the plan found the matrices too large for LDS (tsfa_api.cpp): one HBM slot per workgroup
        dlds = (size_t)TSFA_RED_DOUBLES * sizeof(double) + 64;
        dgrid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(a.n_series, a.dd_slots));
    }
    if ((rc = set_lds(k_ar_degenerate<T>, dlds))) return rc;
    k_ar_degenerate<T><<<dgrid, 64, dlds, st>>>(values, a.starts, a.ends, a.specs, a.nspecs, a.out, a.ld, P, a.deg_list,
                                                a.deg_count, (a.hint_c >> 1) & 3, a.dd_scratch);
    TSFA_LAUNCH_CHECK();
    return 0;
}
int tsfa_launch_ar_degenerate(const TsfaLaunch &a) {
    if (!(a.hint_c || a.ar_has_coef)) return 0;  // no ADF / ar_coefficient column: nothing can degenerate
    if (a.dtype == 0) return launch_ar_degenerate_t<float>(a, (const float *)a.values);
    return launch_ar_degenerate_t<double>(a, (const double *)a.values);
}

// second pass of the SORT family over the fits k_sort recorded (after tsfa_launch_family / _long of TSFA_FAM_SORT)
int tsfa_launch_langevin_dd(const TsfaLaunch &a) {
    if (!a.pf_buf) return 0;   // the plan holds no Langevin fit
    hipStream_t st = (hipStream_t)a.stream;
    const unsigned grid = (unsigned)std::min<int64_t>((a.n_series + 63) / 64, 4096);
    k_langevin_dd<<<grid, 64, 0, st>>>(a.pf_buf, a.pf_count, a.pf_slot, a.pf_cap, a.specs, a.nspecs, a.out, a.ld);
    TSFA_LAUNCH_CHECK();
    return 0;
}

size_t tsfa_entropy_lds_bytes(int maxn, int with_cnt) {
    if (with_cnt == 5) {   // the pair sweep in the long-series build: 32-bit sample order
        EntropyLds L;
        return L.carve(nullptr, maxn, 0, 4);
    }
    if (with_cnt == 4) {   // fam_entropy_hbits.h: the HBM slot of a workgroup (its LDS is entropy_huge_lds_bytes)
        EntropyHugeSlot H;
        return H.carve(nullptr, maxn);
    }
    EntropyLds L;
    return L.carve(nullptr, maxn, with_cnt);
}

size_t tsfa_seq_lds_bytes(const TsfaSeqGroup &g) {
    SeqLds L;
    return L.carve(nullptr, g.grows ? 0 : 1, g.stride, g.ttotal, g.etotal);
}

size_t tsfa_family_lds_bytes(int fam, int maxn, int nt, int aux) {
    switch (fam) {
    case TSFA_FAM_BASIC: { BasicLds L; return L.carve(nullptr, maxn, nt, 8, 1); }
    case TSFA_FAM_TREND: { BasicLds L; return L.carve(nullptr, maxn, nt, 8, 2, aux); }
    case TSFA_FAM_SORT: { SortLds L; return L.carve(nullptr, maxn, nt); }
    case TSFA_FAM_SPECTRAL: { SpectralLds L; return L.carve(nullptr, maxn, aux); }
    case TSFA_FAM_AR: { ArLds L; return L.carve(nullptr, maxn, aux); }
    case TSFA_FAM_CWT: {   // beyond a CU's LDS the long-series build runs, with 32-bit column indices
        CwtPeaksLayout L;
        const size_t lds16 = L.carve(nullptr, maxn, aux, 8, 2);
        return lds16 > TSFA_LDS_LIMIT ? L.carve(nullptr, maxn, aux, 8, 4) : lds16;
    }
    default: return 0;
    }
}

int tsfa_launch_family(const TsfaLaunch &a) {
    if (a.dtype == 0) return launch_all_t<float>(a, (const float *)a.values);
    return launch_all_t<double>(a, (const double *)a.values);
}

// SORT family of a plan that holds only median / quantile columns, series of at most 2048 samples: selection in registers
template <typename T>
static int launch_order_stats_t(const TsfaLaunch &a, const T *values) {
    hipStream_t st = (hipStream_t)a.stream;
    const dim3 grid((unsigned)((a.n_series + 3) / 4));
    if (a.maxn <= 256)
        k_order_stats<T, 4><<<grid, 256, 0, st>>>(values, a.starts, a.ends, a.n_series, a.sel, a.specs, a.nspecs, a.out, a.ld);
    else if (a.maxn <= 1024)
        k_order_stats<T, 16><<<grid, 256, 0, st>>>(values, a.starts, a.ends, a.n_series, a.sel, a.specs, a.nspecs, a.out, a.ld);
    else if (a.maxn <= 2048)
        k_order_stats<T, 32><<<grid, 256, 0, st>>>(values, a.starts, a.ends, a.n_series, a.sel, a.specs, a.nspecs, a.out, a.ld);
    else
        return -1;
    TSFA_LAUNCH_CHECK();
    return 0;
}
// the fused streaming kernel: a.specs / a.nspecs = the SORT family's (median) columns, a.bspecs / a.nbspecs the BASIC ones
template <typename T>
static int launch_stream_t(const TsfaLaunch &a, const T *values) {
    hipStream_t st = (hipStream_t)a.stream;
    const dim3 grid((unsigned)((a.n_series + 3) / 4));
    const int wm = a.nspecs > 0 ? 1 : 0;
#define TSFA_STREAM_CASE(EE) k_stream<T, EE><<<grid, 256, 0, st>>>(values, a.starts, a.ends, a.n_series, a.sel, a.bspecs, a.nbspecs, a.specs, a.nspecs, wm, a.out, a.ld)
    if (a.maxn <= 256) TSFA_STREAM_CASE(4);
    else if (a.maxn <= 512) TSFA_STREAM_CASE(8);
    else if (a.maxn <= 1024) TSFA_STREAM_CASE(16);
    else if (a.maxn <= 2048) TSFA_STREAM_CASE(32);
    else return -1;
#undef TSFA_STREAM_CASE
    TSFA_LAUNCH_CHECK();
    return 0;
}
int tsfa_launch_stream(const TsfaLaunch &a) {
    if (a.dtype == 0) return launch_stream_t<float>(a, (const float *)a.values);
    return launch_stream_t<double>(a, (const double *)a.values);
}
int tsfa_stream_calc_ok(int calc) {
    switch (calc) {
    case TSFA_C_SUM_VALUES: case TSFA_C_MEAN: case TSFA_C_LENGTH: case TSFA_C_STANDARD_DEVIATION: case TSFA_C_VARIANCE:
    case TSFA_C_ROOT_MEAN_SQUARE: case TSFA_C_MAXIMUM: case TSFA_C_ABSOLUTE_MAXIMUM: case TSFA_C_MINIMUM:
    case TSFA_C_ABS_ENERGY: case TSFA_C_VARIATION_COEFFICIENT: case TSFA_C_QUERY_SIMILARITY_COUNT: case TSFA_C_MEDIAN:
        return 1;
    default: return 0;
    }
}

size_t tsfa_perm_lds_bytes(int maxn, int nt, int elem_bytes) {
    PermLds L;
    return L.carve(nullptr, maxn, nt, elem_bytes, TSFA_PE_HIST_WORDS, TSFA_PE_LOGS + TSFA_PE_MAXD + 1);
}
template <typename T>
static int launch_perm_t(const TsfaLaunch &a, const T *values) {
    hipStream_t st = (hipStream_t)a.stream;
    const dim3 grid((unsigned)a.n_series);
    const int nt = a.nt;
    const size_t lds = tsfa_perm_lds_bytes(a.maxn, nt, (int)sizeof(T));
    int rc = 0;
    if ((rc = set_lds(k_perm<T>, lds))) return rc;
    k_perm<T><<<grid, nt, lds, st>>>(values, a.starts, a.ends, a.n_series, a.sel, a.specs, a.nspecs, a.out, a.ld, a.maxn, a.hint_d >> 8);
    TSFA_LAUNCH_CHECK();
    return 0;
}
int tsfa_launch_perm(const TsfaLaunch &a) {
    if (((unsigned)a.hint_d & 0xFFu) != TSFA_PE_MASK || (a.hint_d >> 8) < 1) return -1;
    if (a.dtype == 0) return launch_perm_t<float>(a, (const float *)a.values);
    return launch_perm_t<double>(a, (const double *)a.values);
}

int tsfa_launch_order_stats(const TsfaLaunch &a) {
    if (a.dtype == 0) return launch_order_stats_t<float>(a, (const float *)a.values);
    return launch_order_stats_t<double>(a, (const double *)a.values);
}

template <typename T>
static int launch_cwt_t(const TsfaCwtLaunch &a, const T *values) {
    hipStream_t st = (hipStream_t)a.stream;
    const int ct = (a.C + 15) / 16;
    const dim3 grid((unsigned)((a.n_series + 63) / 64));
#define TSFA_CWT_CASE(N)                                                                                         \
    case N:                                                                                                      \
        k_cwt_gemm<T, N><<<grid, 256, 0, st>>>(values, a.starts, a.ends, a.n_series, a.W, a.S4, a.C, a.cols, a.coeff_idx, \
                                               a.out, a.ld);                                                     \
        break;
    switch (ct) {
        TSFA_CWT_CASE(1)
        TSFA_CWT_CASE(2)
        TSFA_CWT_CASE(3)
        TSFA_CWT_CASE(4)
        TSFA_CWT_CASE(5)
        TSFA_CWT_CASE(6)
        TSFA_CWT_CASE(7)
        TSFA_CWT_CASE(8)
    default: return -1;
    }
#undef TSFA_CWT_CASE
    TSFA_LAUNCH_CHECK();
    return 0;
}

int tsfa_launch_cwt(const TsfaCwtLaunch &a) {
    if (a.dtype == 0) return launch_cwt_t<float>(a, (const float *)a.values);
    return launch_cwt_t<double>(a, (const double *)a.values);
}

int tsfa_launch_fill_nan(double *out, int64_t n_rows, int64_t n_cols, int64_t ld, void *stream, double value,
                         const int *cols, int n_fill) {
    if (cols == nullptr) k_fill_nan<<<2048, 256, 0, (hipStream_t)stream>>>(out, n_rows, n_cols, ld, value);
    else if (n_fill > 0) {
        const int64_t total = n_rows * (int64_t)n_fill;
        k_fill_cols<<<(unsigned)std::min<int64_t>((total + 255) / 256, 2048), 256, 0, (hipStream_t)stream>>>(out, n_rows, cols, n_fill, ld, value);
    }
    TSFA_LAUNCH_CHECK();
    return 0;
}

int tsfa_launch_len_stats(const int64_t *starts, const int64_t *ends, int64_t n_series, long long *stats, void *stream) {
    k_len_stats<<<256, 256, 0, (hipStream_t)stream>>>(starts, ends, n_series, stats);
    TSFA_LAUNCH_CHECK();
    return 0;
}

int tsfa_launch_class_fill(const int64_t *starts, const int64_t *ends, int64_t n_series, const TsfaClassMap &g, int *cursor,
                           int *sel, void *stream) {
    k_class_fill<<<256, 256, 0, (hipStream_t)stream>>>(starts, ends, n_series, g, cursor, sel);
    TSFA_LAUNCH_CHECK();
    return 0;
}

#ifdef TSFA_TICKS
// diagnostics build only: read (and optionally clear) the phase clocks
extern "C" int tsfa_debug_ticks(unsigned long long *out, int n, int reset) {
    if (n > 256) n = 256;
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(tsfa_ticks), (size_t)n * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        static unsigned long long zeros[256];
        if (hipMemcpyToSymbol(HIP_SYMBOL(tsfa_ticks), zeros, sizeof zeros) != hipSuccess) return -1;
    }
    return 0;
}
#endif
#endif  // !TSFA_LONG
