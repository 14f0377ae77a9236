// Relevance statistics for feature selection: for every column of the extracted feature matrix the
// sufficient statistics of the reference's univariate tests against a class-coded target
//   tsfresh/feature_selection/significance_tests.py:84  target_binary_feature_real_test   (Mann-Whitney U: rank sums
//                                                        with mid-ranks + the tie term of the normal approximation)
//   tsfresh/feature_selection/significance_tests.py:43  target_binary_feature_binary_test (Fisher: 2 x 2 counts)
//   tsfresh/feature_selection/relevance.py:396          get_feature_type                  (1 / 2 / more distinct values)
// in ONE batched sweep: a workgroup owns a column, sorts its (value, row) pairs in HBM scratch (bitonic network:
// 4096-element tiles finished in LDS, only the strides >= 4096 as global passes), then ranks them (ties found by
// binary search in the sorted column) and accumulates rank sums / counts per class with LDS atomics.  The sums are
// exact (mid-ranks are multiples of 0.5, totals < 2^53), so the order of the atomics does not matter.
// The p-value tails (O(1) per feature) are the host's: tsfresh_amd/feature_selection/significance_tests.py.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <string>
#include <vector>

#include "../../include/tsfresh_amd.h"

int tsfa_fail(int code, const char *msg);  // tsfa_api.cpp: records the thread-local message, returns code

#define REL_NT 1024
#define REL_TILE 4096
#define REL_MAXC 256

__global__ void __launch_bounds__(REL_NT) k_rel_stage(const double *__restrict__ X, int64_t n, int64_t ld, int64_t c0,
                                                       double *__restrict__ keys, uint32_t *__restrict__ idx, int64_t np2) {
    const int64_t c = blockIdx.x;
    double *K = keys + c * np2;
    uint32_t *I = idx + c * np2;
    const double *col = X + c0 + c;
    for (int64_t r = threadIdx.x; r < np2; r += REL_NT) {
        K[r] = (r < n) ? col[r * ld] : __builtin_inf();
        I[r] = (uint32_t)r;
    }
}

// compare-exchange stages j = jmax, jmax/2, .. 1 of merge level k on one LDS tile whose first element has global index base
__device__ __forceinline__ void rel_lds_stages(double *sk, uint32_t *si, int tile, int64_t base, int64_t k, int jmax) {
    for (int j = jmax; j > 0; j >>= 1) {
        __syncthreads();
        for (int t = threadIdx.x; t < (tile >> 1); t += REL_NT) {
            const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
            const int l = i | j;
            const bool up = (((base + i) & k) == 0);
            const double a = sk[i], b = sk[l];
            const uint32_t ia = si[i], ib = si[l];
            // ties by row: a total order, so the +inf pads (rows >= n) stay behind real +inf values
            if (((a > b) || (a == b && ia > ib)) == up) {
                sk[i] = b; sk[l] = a;
                si[i] = ib; si[l] = ia;
            }
        }
    }
    __syncthreads();
}

__global__ void __launch_bounds__(REL_NT) k_rel_sort(double *__restrict__ keys, uint32_t *__restrict__ idx, int64_t np2, int tile) {
    extern __shared__ unsigned char rel_smem[];
    double *sk = (double *)rel_smem;
    uint32_t *si = (uint32_t *)(rel_smem + (size_t)tile * sizeof(double));
    double *K = keys + (int64_t)blockIdx.x * np2;
    uint32_t *I = idx + (int64_t)blockIdx.x * np2;
    // phase A: every tile fully sorted in LDS (direction from the element's global index, as the network requires)
    for (int64_t base = 0; base < np2; base += tile) {
        for (int t = threadIdx.x; t < tile; t += REL_NT) { sk[t] = K[base + t]; si[t] = I[base + t]; }
        for (int64_t k = 2; k <= tile; k <<= 1) rel_lds_stages(sk, si, tile, base, k, (int)(k >> 1));
        for (int t = threadIdx.x; t < tile; t += REL_NT) { K[base + t] = sk[t]; I[base + t] = si[t]; }
        __syncthreads();
    }
    // phase B: merge levels above the tile size: strides >= tile in HBM (L2-resident), the rest of the level in LDS
    for (int64_t k = 2 * (int64_t)tile; k <= np2; k <<= 1) {
        for (int64_t j = k >> 1; j >= tile; j >>= 1) {
            __syncthreads();
            for (int64_t t = threadIdx.x; t < (np2 >> 1); t += REL_NT) {
                const int64_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int64_t l = i | j;
                const bool up = ((i & k) == 0);
                const double a = K[i], b = K[l];
                const uint32_t ia = I[i], ib = I[l];
                if (((a > b) || (a == b && ia > ib)) == up) {
                    K[i] = b; K[l] = a;
                    I[i] = ib; I[l] = ia;
                }
            }
        }
        __syncthreads();
        for (int64_t base = 0; base < np2; base += tile) {
            for (int t = threadIdx.x; t < tile; t += REL_NT) { sk[t] = K[base + t]; si[t] = I[base + t]; }
            rel_lds_stages(sk, si, tile, base, k, tile >> 1);
            for (int t = threadIdx.x; t < tile; t += REL_NT) { K[base + t] = sk[t]; I[base + t] = si[t]; }
            __syncthreads();
        }
    }
}

__global__ void __launch_bounds__(REL_NT) k_rel_stats(const double *__restrict__ keys, const uint32_t *__restrict__ idx, int64_t np2, int64_t n,
                                                       const int32_t *__restrict__ y, int n_classes, int64_t c0,
                                                       tsfa_relevance_col *__restrict__ cols, double *__restrict__ rank_sums,
                                                       int64_t *__restrict__ hi_counts) {
    __shared__ double s_rs[REL_MAXC];
    __shared__ unsigned long long s_hc[REL_MAXC];
    __shared__ double s_tie;
    __shared__ unsigned long long s_uniq;
    const double *K = keys + (int64_t)blockIdx.x * np2;
    const uint32_t *I = idx + (int64_t)blockIdx.x * np2;
    for (int k = threadIdx.x; k < REL_MAXC; k += REL_NT) { s_rs[k] = 0.0; s_hc[k] = 0ull; }
    if (threadIdx.x == 0) { s_tie = 0.0; s_uniq = 0ull; }
    __syncthreads();
    const double vhi = K[n - 1];
    double tie = 0.0;
    unsigned long long uniq = 0ull;
    for (int64_t i = threadIdx.x; i < n; i += REL_NT) {
        const double key = K[i];
        const bool start = (i == 0) || (K[i - 1] != key);
        const bool end = (i == n - 1) || (K[i + 1] != key);
        int64_t lb = i, ub = i + 1;
        if (!start) {  // first position of the tie group
            int64_t lo = 0, hi = i;
            while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (K[mid] < key) lo = mid + 1; else hi = mid; }
            lb = lo;
        }
        if (!end) {    // one past its last position
            int64_t lo = i + 1, hi = n;
            while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (K[mid] <= key) lo = mid + 1; else hi = mid; }
            ub = lo;
        }
        const double midrank = 0.5 * (double)(lb + ub + 1);  // mean of the 1-based ranks lb + 1 .. ub
        if (start) {
            const double t = (double)(ub - lb);
            tie += t * t * t - t;
            ++uniq;
        }
        const int cls = y[I[i]];
        atomicAdd(&s_rs[cls], midrank);
        if (key == vhi) atomicAdd(&s_hc[cls], 1ull);
    }
    atomicAdd(&s_tie, tie);
    atomicAdd(&s_uniq, uniq);
    __syncthreads();
    const int64_t c = c0 + blockIdx.x;
    for (int k = threadIdx.x; k < n_classes; k += REL_NT) {
        rank_sums[c * n_classes + k] = s_rs[k];
        hi_counts[c * n_classes + k] = (int64_t)s_hc[k];
    }
    if (threadIdx.x == 0) {
        cols[c].n_unique = (int64_t)s_uniq;
        cols[c].v_lo = K[0];
        cols[c].v_hi = vhi;
        cols[c].tie_term = s_tie;
    }
}

// 'smir' option (significance_tests.py:121: scipy.stats.ks_2samp(x[y == label], x[y != label])): the Kolmogorov-Smirnov
// distance of the column split by each class label, walked along the sorted column (prefix counts by a block scan,
// evaluated where a tie group of the column ends).
__global__ void __launch_bounds__(REL_NT) k_rel_ks_classes(const double *__restrict__ keys, const uint32_t *__restrict__ idx, int64_t np2, int64_t n,
                                                            const int32_t *__restrict__ y, int n_classes, int64_t c0,
                                                            double *__restrict__ ks_d) {
    __shared__ long long s_cnt[REL_NT];
    __shared__ double s_max[REL_NT], s_min[REL_NT];
    __shared__ long long s_total;
    const double *K = keys + (int64_t)blockIdx.x * np2;
    const uint32_t *I = idx + (int64_t)blockIdx.x * np2;
    const int64_t chunk = (n + REL_NT - 1) / REL_NT;
    const int64_t p0 = (int64_t)threadIdx.x * chunk, p1 = (p0 + chunk < n) ? p0 + chunk : n;
    for (int k = 0; k < n_classes; ++k) {
        long long c = 0;
        for (int64_t p = p0; p < p1; ++p) c += (y[I[p]] == k) ? 1 : 0;
        __syncthreads();
        s_cnt[threadIdx.x] = c;
        __syncthreads();
        if (threadIdx.x == 0) {
            long long acc = 0;
            for (int t = 0; t < REL_NT; ++t) { const long long v = s_cnt[t]; s_cnt[t] = acc; acc += v; }
            s_total = acc;
        }
        __syncthreads();
        const double n1 = (double)s_total, n0 = (double)(n - s_total);
        long long k1 = s_cnt[threadIdx.x];
        double mx = -__builtin_inf(), mn = __builtin_inf();
        for (int64_t p = p0; p < p1; ++p) {
            k1 += (y[I[p]] == k) ? 1 : 0;
            if (p == n - 1 || K[p + 1] != K[p]) {
                const double d = (double)k1 / n1 - (double)(p + 1 - k1) / n0;
                mx = fmax(mx, d);
                mn = fmin(mn, d);
            }
        }
        s_max[threadIdx.x] = mx; s_min[threadIdx.x] = mn;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int t = 1; t < REL_NT; ++t) { mx = fmax(mx, s_max[t]); mn = fmin(mn, s_min[t]); }
            double mins = -mn;
            mins = (mins < 0.0) ? 0.0 : ((mins > 1.0) ? 1.0 : mins);
            ks_d[(c0 + blockIdx.x) * n_classes + k] = (mins > mx) ? mins : mx;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Real-valued targets (significance_tests.py:170 target_real_feature_real_test = scipy.stats.kendalltau, :135
// target_real_feature_binary_test = scipy.stats.ks_2samp).  The column is sorted by (value, dense rank of y) -- the
// payload of the sort is the y rank instead of the row -- which is the order scipy's kendalltau establishes; then
//   k_rel_xties       tie statistics of x and of the (x, y) pairs (group ends by binary search),
//   k_rel_inversions  discordant pairs = strict inversions of the y-rank sequence: bottom-up merge sort, runs up to 4096
//                     in LDS (rank of every element in the sibling run by binary search), longer runs by merge-path
//                     segments in HBM scratch (each thread merges one contiguous output segment and counts, for every
//                     element taken from the right run, the left elements still waiting),
//   k_rel_ks          for two-valued columns: the two-sample Kolmogorov-Smirnov distance of y split by the column,
//                     walked in y order (prefix counts by a block scan, evaluated where a y tie group ends).
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(REL_NT) k_rel_stage_real(const double *__restrict__ X, int64_t n, int64_t ld, int64_t c0,
                                                            const int32_t *__restrict__ yrank, double *__restrict__ keys,
                                                            uint32_t *__restrict__ idx, int64_t np2) {
    const int64_t c = blockIdx.x;
    double *K = keys + c * np2;
    uint32_t *I = idx + c * np2;
    const double *col = X + c0 + c;
    for (int64_t r = threadIdx.x; r < np2; r += REL_NT) {
        K[r] = (r < n) ? col[r * ld] : __builtin_inf();
        I[r] = (r < n) ? (uint32_t)yrank[r] : 0xFFFFFFFFu;
    }
}

__global__ void __launch_bounds__(REL_NT) k_rel_xties(const double *__restrict__ keys, const uint32_t *__restrict__ idx, int64_t np2, int64_t n,
                                                       int64_t c0, tsfa_relevance_real_col *__restrict__ cols) {
    __shared__ unsigned long long s_xtie, s_ntie, s_uniq;
    __shared__ double s_x0, s_x1;
    const double *K = keys + (int64_t)blockIdx.x * np2;
    const uint32_t *I = idx + (int64_t)blockIdx.x * np2;
    if (threadIdx.x == 0) { s_xtie = 0ull; s_ntie = 0ull; s_uniq = 0ull; s_x0 = 0.0; s_x1 = 0.0; }
    __syncthreads();
    unsigned long long xtie = 0ull, ntie = 0ull, uniq = 0ull;
    double x0 = 0.0, x1 = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += REL_NT) {
        const double key = K[i];
        const uint32_t yr = I[i];
        const bool xstart = (i == 0) || (K[i - 1] != key);
        const bool jstart = xstart || (I[i - 1] != yr);
        if (xstart) {
            ++uniq;
            if (i + 1 < n && K[i + 1] == key) {  // a tie group of x starts here: its end by binary search
                int64_t lo = i + 1, hi = n;
                while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (K[mid] <= key) lo = mid + 1; else hi = mid; }
                const unsigned long long t = (unsigned long long)(lo - i);
                xtie += t * (t - 1ull) / 2ull;
                const double td = (double)t;
                x0 += td * (td - 1.0) * (td - 2.0);
                x1 += td * (td - 1.0) * (2.0 * td + 5.0);
            }
        }
        if (jstart && i + 1 < n && K[i + 1] == key && I[i + 1] == yr) {  // joint tie group of (x, y)
            int64_t lo = i + 1, hi = n;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                const double km = K[mid];
                if (km < key || (km == key && I[mid] <= yr)) lo = mid + 1; else hi = mid;
            }
            const unsigned long long t = (unsigned long long)(lo - i);
            ntie += t * (t - 1ull) / 2ull;
        }
    }
    atomicAdd(&s_xtie, xtie); atomicAdd(&s_ntie, ntie); atomicAdd(&s_uniq, uniq);
    atomicAdd(&s_x0, x0); atomicAdd(&s_x1, x1);
    __syncthreads();
    if (threadIdx.x == 0) {
        tsfa_relevance_real_col &o = cols[c0 + blockIdx.x];
        o.n_unique = (int64_t)s_uniq;
        o.v_lo = K[0];
        o.v_hi = K[n - 1];
        o.xtie = (int64_t)s_xtie;
        o.ntie = (int64_t)s_ntie;
        o.x0 = s_x0;
        o.x1 = s_x1;
        o.dis = 0;
        o.n_hi = 0;
        o.ks_d = 0.0;
    }
}

// strict inversions of seq[0 .. np2) (padded with 0xFFFFFFFF); tmp: a second buffer of np2 words; both are clobbered
__global__ void __launch_bounds__(REL_NT) k_rel_inversions(uint32_t *__restrict__ idx, uint32_t *__restrict__ tmpbuf, int64_t np2, int tile,
                                                            int64_t c0, tsfa_relevance_real_col *__restrict__ cols) {
    extern __shared__ unsigned char rel_smem[];
    uint32_t *sa = (uint32_t *)rel_smem, *sb = sa + tile;
    __shared__ unsigned long long s_inv;
    uint32_t *A = idx + (int64_t)blockIdx.x * np2;
    uint32_t *B = tmpbuf + (int64_t)blockIdx.x * np2;
    if (threadIdx.x == 0) s_inv = 0ull;
    unsigned long long inv = 0ull;
    for (int64_t base = 0; base < np2; base += tile) {
        __syncthreads();
        for (int t = threadIdx.x; t < tile; t += REL_NT) sa[t] = A[base + t];
        uint32_t *src = sa, *dst = sb;
        for (int run = 1; run < tile; run <<= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < tile; t += REL_NT) {
                const int off = t & (2 * run - 1), lbase = t - off, rbase = lbase + run;
                const uint32_t v = src[t];
                int pos;
                if (off < run) {  // left run: elements of the right run that are smaller go first
                    int lo = 0, hi = run;
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (src[rbase + mid] < v) lo = mid + 1; else hi = mid; }
                    pos = off + lo;
                } else {          // right run: elements of the left run that are <= v go first; the others are inversions
                    int lo = 0, hi = run;
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (src[lbase + mid] <= v) lo = mid + 1; else hi = mid; }
                    pos = (off - run) + lo;
                    inv += (unsigned long long)(run - lo);
                }
                dst[lbase + pos] = v;
            }
            uint32_t *sw = src; src = dst; dst = sw;
        }
        __syncthreads();
        for (int t = threadIdx.x; t < tile; t += REL_NT) A[base + t] = src[t];
    }
    // merge levels above the tile: one contiguous output segment per thread (merge path), ping-pong A <-> B
    const int64_t seg = np2 / REL_NT;  // np2 >= 2048: >= 2
    uint32_t *src = A, *dst = B;
    for (int64_t run = tile; run < np2; run <<= 1) {
        __syncthreads();
        const int64_t o0 = (int64_t)threadIdx.x * seg;
        const int64_t pbase = o0 & ~(2 * run - 1), diag = o0 - pbase;
        const uint32_t *L = src + pbase, *R = src + pbase + run;
        int64_t lo = (diag > run) ? diag - run : 0, hi = (diag < run) ? diag : run;
        while (lo < hi) {  // smallest li with L[li] > R[diag - li - 1]  (ties: the left element goes first)
            const int64_t mid = (lo + hi) >> 1;
            if (L[mid] <= R[diag - mid - 1]) lo = mid + 1; else hi = mid;
        }
        int64_t li = lo, ri = diag - lo;
        for (int64_t k = 0; k < seg; ++k) {
            const bool take_left = (li < run) && (ri >= run || L[li] <= R[ri]);
            if (take_left) { dst[o0 + k] = L[li]; ++li; }
            else { dst[o0 + k] = R[ri]; ++ri; inv += (unsigned long long)(run - li); }
        }
        uint32_t *sw = src; src = dst; dst = sw;
    }
    atomicAdd(&s_inv, inv);
    __syncthreads();
    if (threadIdx.x == 0) cols[c0 + blockIdx.x].dis = (int64_t)s_inv;
}

// two-valued columns: sup |F_hi - F_lo| of y over the rows with the larger / the smaller value of the column
__global__ void __launch_bounds__(REL_NT) k_rel_ks(const double *__restrict__ X, int64_t n, int64_t ld, const int32_t *__restrict__ yperm,
                                                    const unsigned char *__restrict__ yend, tsfa_relevance_real_col *__restrict__ cols) {
    __shared__ long long s_cnt[REL_NT];
    __shared__ double s_max[REL_NT], s_min[REL_NT];
    tsfa_relevance_real_col &o = cols[blockIdx.x];
    if (o.n_unique != 2) return;
    const double vhi = o.v_hi;
    const double *col = X + blockIdx.x;
    const int64_t chunk = (n + REL_NT - 1) / REL_NT;
    const int64_t p0 = (int64_t)threadIdx.x * chunk, p1 = (p0 + chunk < n) ? p0 + chunk : n;
    long long c = 0;
    for (int64_t p = p0; p < p1; ++p) c += (col[(int64_t)yperm[p] * ld] == vhi) ? 1 : 0;
    s_cnt[threadIdx.x] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long acc = 0;
        for (int t = 0; t < REL_NT; ++t) { const long long v = s_cnt[t]; s_cnt[t] = acc; acc += v; }
        o.n_hi = acc;
    }
    __syncthreads();
    const double n1 = (double)o.n_hi, n0 = (double)(n - o.n_hi);
    long long k1 = s_cnt[threadIdx.x];
    double mx = -__builtin_inf(), mn = __builtin_inf();
    for (int64_t p = p0; p < p1; ++p) {
        k1 += (col[(int64_t)yperm[p] * ld] == vhi) ? 1 : 0;
        if (yend[p]) {
            const double d = (double)k1 / n1 - (double)(p + 1 - k1) / n0;  // searchsorted(side="right") / n, as scipy
            mx = fmax(mx, d);
            mn = fmin(mn, d);
        }
    }
    s_max[threadIdx.x] = mx; s_min[threadIdx.x] = mn;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int t = 1; t < REL_NT; ++t) { mx = fmax(mx, s_max[t]); mn = fmin(mn, s_min[t]); }
        double mins = -mn;
        mins = (mins < 0.0) ? 0.0 : ((mins > 1.0) ? 1.0 : mins);
        o.ks_d = (mins > mx) ? mins : mx;
    }
}

// columns per batch: bound the sort scratch (12 B per padded row and column) to ~4 GB; tsfa_plan_set_option(NULL,
// "relevance_batch", n) overrides (tests)
int64_t tsfa_relevance_batch_override();
static int64_t rel_batch_columns(int64_t np2, int64_t n_cols) {
    int64_t batch = (int64_t)(4.0e9 / (12.0 * (double)np2));
    if (tsfa_relevance_batch_override() > 0) batch = tsfa_relevance_batch_override();
    if (batch < 1) batch = 1;
    if (batch > n_cols) batch = n_cols;
    return batch;
}

#define REL_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) { rc = tsfa_fail(TSFA_ERR_HIP, (std::string(#call) + ": " + hipGetErrorString(e_)).c_str()); goto done; } \
    } while (0)

static int relevance_classes_impl(const double *X, int64_t n_rows, int64_t n_cols, int64_t ld, int32_t space,
                                  const int32_t *y_codes, int32_t n_classes, int32_t device, tsfa_relevance_col *cols,
                                  double *rank_sums, int64_t *hi_counts, double *ks_d);

extern "C" int tsfa_relevance_classes(const double *X, int64_t n_rows, int64_t n_cols, int64_t ld, int32_t space,
                                      const int32_t *y_codes, int32_t n_classes, int32_t device, tsfa_relevance_col *cols,
                                      double *rank_sums, int64_t *hi_counts) {
    return relevance_classes_impl(X, n_rows, n_cols, ld, space, y_codes, n_classes, device, cols, rank_sums, hi_counts, nullptr);
}

extern "C" int tsfa_relevance_classes_ks(const double *X, int64_t n_rows, int64_t n_cols, int64_t ld, int32_t space,
                                         const int32_t *y_codes, int32_t n_classes, int32_t device, tsfa_relevance_col *cols,
                                         double *rank_sums, int64_t *hi_counts, double *ks_d) {
    if (!ks_d) return tsfa_fail(TSFA_ERR_INVALID, "tsfa_relevance_classes_ks: null ks_d");
    return relevance_classes_impl(X, n_rows, n_cols, ld, space, y_codes, n_classes, device, cols, rank_sums, hi_counts, ks_d);
}

static int relevance_classes_impl(const double *X, int64_t n_rows, int64_t n_cols, int64_t ld, int32_t space,
                                  const int32_t *y_codes, int32_t n_classes, int32_t device, tsfa_relevance_col *cols,
                                  double *rank_sums, int64_t *hi_counts, double *ks_d) {
    if (!X || !y_codes || !cols || !rank_sums || !hi_counts || n_rows < 1 || n_cols < 0 || ld < n_cols)
        return tsfa_fail(TSFA_ERR_INVALID, "tsfa_relevance_classes: null pointer or bad shape");
    if (n_classes < 1 || n_classes > REL_MAXC) return tsfa_fail(TSFA_ERR_UNSUPPORTED, "tsfa_relevance_classes: 1 .. 256 classes");
    if (n_rows >= (1ll << 31)) return tsfa_fail(TSFA_ERR_TOO_LONG, "tsfa_relevance_classes: more than 2^31 rows");
    for (int64_t r = 0; r < n_rows; ++r)
        if (y_codes[r] < 0 || y_codes[r] >= n_classes) return tsfa_fail(TSFA_ERR_INVALID, "tsfa_relevance_classes: class code out of range");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || device < 0 || device >= ndev)
        return tsfa_fail(TSFA_ERR_NO_DEVICE, "tsfa_relevance_classes: no such HIP device (there is no CPU path)");
    if (n_cols == 0) return TSFA_OK;
    int rc = TSFA_OK;
    int64_t np2 = 2048;
    while (np2 < n_rows) np2 <<= 1;
    const int tile = (int)((np2 < REL_TILE) ? np2 : REL_TILE);
    int64_t batch = rel_batch_columns(np2, n_cols);
    double *dX = nullptr, *dkeys = nullptr, *drs = nullptr, *dks = nullptr;
    uint32_t *didx = nullptr;
    int32_t *dy = nullptr;
    int64_t *dhc = nullptr;
    tsfa_relevance_col *dcols = nullptr;
    const double *Xd = X;
    int64_t ldd = ld;   // leading dimension of the matrix the kernels work on (a host matrix is staged dense)
    REL_HIP(hipSetDevice(device));
    // a host matrix may be a row-strided VIEW (ld > n_cols) that ends with its last row: only the n_cols-wide rows are
    // the caller's -- the device copy is dense (leading dimension n_cols) and the columns between the rows are never
    // read nor written back
    if (space == TSFA_HOST) {
        ldd = n_cols;
        REL_HIP(hipMalloc((void **)&dX, (size_t)n_rows * n_cols * sizeof(double)));
        REL_HIP(hipMemcpy2D(dX, (size_t)n_cols * sizeof(double), X, (size_t)ld * sizeof(double), (size_t)n_cols * sizeof(double),
                            (size_t)n_rows, hipMemcpyHostToDevice));
        Xd = dX;
    }
    REL_HIP(hipMalloc((void **)&dkeys, (size_t)batch * np2 * sizeof(double)));
    REL_HIP(hipMalloc((void **)&didx, (size_t)batch * np2 * sizeof(uint32_t)));
    REL_HIP(hipMalloc((void **)&dy, (size_t)n_rows * sizeof(int32_t)));
    REL_HIP(hipMalloc((void **)&drs, (size_t)n_cols * n_classes * sizeof(double)));
    REL_HIP(hipMalloc((void **)&dhc, (size_t)n_cols * n_classes * sizeof(int64_t)));
    REL_HIP(hipMalloc((void **)&dcols, (size_t)n_cols * sizeof(tsfa_relevance_col)));
    if (ks_d) REL_HIP(hipMalloc((void **)&dks, (size_t)n_cols * n_classes * sizeof(double)));
    REL_HIP(hipMemcpy(dy, y_codes, (size_t)n_rows * sizeof(int32_t), hipMemcpyHostToDevice));
    {
        const size_t lds = (size_t)tile * (sizeof(double) + sizeof(uint32_t));
        REL_HIP(hipFuncSetAttribute((const void *)k_rel_sort, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        for (int64_t c0 = 0; c0 < n_cols; c0 += batch) {
            const int64_t nb = (n_cols - c0 < batch) ? (n_cols - c0) : batch;
            k_rel_stage<<<dim3((unsigned)nb), REL_NT, 0, 0>>>(Xd, n_rows, ldd, c0, dkeys, didx, np2);
            k_rel_sort<<<dim3((unsigned)nb), REL_NT, lds, 0>>>(dkeys, didx, np2, tile);
            k_rel_stats<<<dim3((unsigned)nb), REL_NT, 0, 0>>>(dkeys, didx, np2, n_rows, dy, n_classes, c0, dcols, drs, dhc);
            if (ks_d) k_rel_ks_classes<<<dim3((unsigned)nb), REL_NT, 0, 0>>>(dkeys, didx, np2, n_rows, dy, n_classes, c0, dks);
            REL_HIP(hipGetLastError());
        }
    }
    REL_HIP(hipMemcpy(cols, dcols, (size_t)n_cols * sizeof(tsfa_relevance_col), hipMemcpyDeviceToHost));
    REL_HIP(hipMemcpy(rank_sums, drs, (size_t)n_cols * n_classes * sizeof(double), hipMemcpyDeviceToHost));
    REL_HIP(hipMemcpy(hi_counts, dhc, (size_t)n_cols * n_classes * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (ks_d) REL_HIP(hipMemcpy(ks_d, dks, (size_t)n_cols * n_classes * sizeof(double), hipMemcpyDeviceToHost));
done:
    (void)hipFree(dX); (void)hipFree(dkeys); (void)hipFree(didx); (void)hipFree(dy); (void)hipFree(drs); (void)hipFree(dhc); (void)hipFree(dcols);
    (void)hipFree(dks);
    return rc;
}

extern "C" int tsfa_relevance_real(const double *X, int64_t n_rows, int64_t n_cols, int64_t ld, int32_t space,
                                   const int32_t *y_rank, const int32_t *y_perm, const unsigned char *y_end, int32_t device,
                                   tsfa_relevance_real_col *cols) {
    if (!X || !y_rank || !y_perm || !y_end || !cols || n_rows < 1 || n_cols < 0 || ld < n_cols)
        return tsfa_fail(TSFA_ERR_INVALID, "tsfa_relevance_real: null pointer or bad shape");
    if (n_rows >= (1ll << 31)) return tsfa_fail(TSFA_ERR_TOO_LONG, "tsfa_relevance_real: more than 2^31 rows");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || device < 0 || device >= ndev)
        return tsfa_fail(TSFA_ERR_NO_DEVICE, "tsfa_relevance_real: no such HIP device (there is no CPU path)");
    if (n_cols == 0) return TSFA_OK;
    int rc = TSFA_OK;
    int64_t np2 = 2048;
    while (np2 < n_rows) np2 <<= 1;
    const int tile = (int)((np2 < REL_TILE) ? np2 : REL_TILE);
    int64_t batch = rel_batch_columns(np2, n_cols);
    double *dX = nullptr, *dkeys = nullptr;
    uint32_t *didx = nullptr;
    int32_t *dyr = nullptr, *dyp = nullptr;
    unsigned char *dye = nullptr;
    tsfa_relevance_real_col *dcols = nullptr;
    const double *Xd = X;
    int64_t ldd = ld;   // leading dimension of the matrix the kernels work on (a host matrix is staged dense)
    REL_HIP(hipSetDevice(device));
    // a host matrix may be a row-strided VIEW (ld > n_cols) that ends with its last row: only the n_cols-wide rows are
    // the caller's -- the device copy is dense (leading dimension n_cols) and the columns between the rows are never
    // read nor written back
    if (space == TSFA_HOST) {
        ldd = n_cols;
        REL_HIP(hipMalloc((void **)&dX, (size_t)n_rows * n_cols * sizeof(double)));
        REL_HIP(hipMemcpy2D(dX, (size_t)n_cols * sizeof(double), X, (size_t)ld * sizeof(double), (size_t)n_cols * sizeof(double),
                            (size_t)n_rows, hipMemcpyHostToDevice));
        Xd = dX;
    }
    REL_HIP(hipMalloc((void **)&dkeys, (size_t)batch * np2 * sizeof(double)));
    REL_HIP(hipMalloc((void **)&didx, (size_t)batch * np2 * sizeof(uint32_t)));
    REL_HIP(hipMalloc((void **)&dyr, (size_t)n_rows * sizeof(int32_t)));
    REL_HIP(hipMalloc((void **)&dyp, (size_t)n_rows * sizeof(int32_t)));
    REL_HIP(hipMalloc((void **)&dye, (size_t)n_rows));
    REL_HIP(hipMalloc((void **)&dcols, (size_t)n_cols * sizeof(tsfa_relevance_real_col)));
    REL_HIP(hipMemcpy(dyr, y_rank, (size_t)n_rows * sizeof(int32_t), hipMemcpyHostToDevice));
    REL_HIP(hipMemcpy(dyp, y_perm, (size_t)n_rows * sizeof(int32_t), hipMemcpyHostToDevice));
    REL_HIP(hipMemcpy(dye, y_end, (size_t)n_rows, hipMemcpyHostToDevice));
    {
        const size_t lds_sort = (size_t)tile * (sizeof(double) + sizeof(uint32_t));
        const size_t lds_inv = (size_t)tile * 2 * sizeof(uint32_t);
        REL_HIP(hipFuncSetAttribute((const void *)k_rel_sort, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sort));
        REL_HIP(hipFuncSetAttribute((const void *)k_rel_inversions, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_inv));
        for (int64_t c0 = 0; c0 < n_cols; c0 += batch) {
            const int64_t nb = (n_cols - c0 < batch) ? (n_cols - c0) : batch;
            k_rel_stage_real<<<dim3((unsigned)nb), REL_NT, 0, 0>>>(Xd, n_rows, ldd, c0, dyr, dkeys, didx, np2);
            k_rel_sort<<<dim3((unsigned)nb), REL_NT, lds_sort, 0>>>(dkeys, didx, np2, tile);
            k_rel_xties<<<dim3((unsigned)nb), REL_NT, 0, 0>>>(dkeys, didx, np2, n_rows, c0, dcols);
            // the sorted values are dead now: their storage is the second buffer of the merge sort (np2 words per column
            // out of the np2 doubles)
            k_rel_inversions<<<dim3((unsigned)nb), REL_NT, lds_inv, 0>>>(didx, (uint32_t *)dkeys, np2, tile, c0, dcols);
            REL_HIP(hipGetLastError());
        }
        k_rel_ks<<<dim3((unsigned)n_cols), REL_NT, 0, 0>>>(Xd, n_rows, ldd, dyp, dye, dcols);
        REL_HIP(hipGetLastError());
    }
    REL_HIP(hipMemcpy(cols, dcols, (size_t)n_cols * sizeof(tsfa_relevance_real_col), hipMemcpyDeviceToHost));
done:
    (void)hipFree(dX); (void)hipFree(dkeys); (void)hipFree(didx); (void)hipFree(dyr); (void)hipFree(dyp); (void)hipFree(dye); (void)hipFree(dcols);
    return rc;
}

// ---------------------------------------------------------------------------------------------------------------
// impute (tsfresh/utilities/dataframe_functions.py:49-214): per column the largest / smallest / median FINITE value
// (get_range_values_per_column :142-180: np.ma.masked_invalid + np.max / np.min / np.ma.median; a column without any
// finite value counts as all zeros), then +inf -> max, -inf -> min, NaN -> median (impute_dataframe_range :96-139).
// The column statistics come from the same staged HBM sort as the relevance tests: non-finite cells are staged as +inf
// and therefore sort behind the `cnt` finite ones.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(REL_NT) k_imp_stage(const double *__restrict__ X, int64_t n, int64_t ld, int64_t c0,
                                                       double *__restrict__ keys, uint32_t *__restrict__ idx, int64_t np2,
                                                       int *__restrict__ finite_cnt) {
    const int64_t c = blockIdx.x;
    double *K = keys + c * np2;
    uint32_t *I = idx + c * np2;
    const double *col = X + c0 + c;
    int cnt = 0;
    for (int64_t r = threadIdx.x; r < np2; r += REL_NT) {
        double v = __builtin_inf();
        if (r < n) {
            const double x = col[r * ld];
            if (x == x && x != __builtin_inf() && x != -__builtin_inf()) { v = x; ++cnt; }
        }
        K[r] = v;
        I[r] = (uint32_t)r;
    }
    __shared__ int red[REL_NT / 64];
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < REL_NT / 64; ++w) t += red[w];
        finite_cnt[c0 + c] = t;
    }
}

__global__ void k_imp_stats(const double *__restrict__ keys, int64_t np2, const int *__restrict__ finite_cnt, int64_t c0, int64_t nb,
                            double *__restrict__ cmax, double *__restrict__ cmin, double *__restrict__ cmed) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nb) return;
    const double *K = keys + c * np2;
    const int cnt = finite_cnt[c0 + c];
    double mx = 0.0, mn = 0.0, md = 0.0;
    if (cnt > 0) {
        mn = K[0];
        mx = K[cnt - 1];
        md = (cnt & 1) ? K[cnt / 2] : (K[cnt / 2 - 1] + K[cnt / 2]) / 2.0;  // np.ma.median: mean of the two middle values
    }
    cmax[c0 + c] = mx;
    cmin[c0 + c] = mn;
    cmed[c0 + c] = md;
}

__global__ void __launch_bounds__(256) k_imp_apply(double *__restrict__ X, int64_t n, int64_t ld, int64_t n_cols,
                                                   const double *__restrict__ cmax, const double *__restrict__ cmin,
                                                   const double *__restrict__ cmed) {
    const int64_t total = n * n_cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / n_cols, c = i - r * n_cols;
        const double v = X[r * ld + c];
        if (v != v) X[r * ld + c] = cmed[c];
        else if (v == __builtin_inf()) X[r * ld + c] = cmax[c];
        else if (v == -__builtin_inf()) X[r * ld + c] = cmin[c];
    }
}

extern "C" int tsfa_impute(double *X, int64_t n_rows, int64_t n_cols, int64_t ld, int32_t space, int32_t device,
                           double *col_max, double *col_min, double *col_median, int32_t *finite_count) {
    if (!X || n_rows < 0 || n_cols < 0 || ld < n_cols) return tsfa_fail(TSFA_ERR_INVALID, "tsfa_impute: null pointer or bad shape");
    if (n_rows >= (1ll << 31)) return tsfa_fail(TSFA_ERR_TOO_LONG, "tsfa_impute: more than 2^31 rows");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || device < 0 || device >= ndev)
        return tsfa_fail(TSFA_ERR_NO_DEVICE, "tsfa_impute: no such HIP device (there is no CPU path)");
    if (n_cols == 0 || n_rows == 0) return TSFA_OK;
    int rc = TSFA_OK;
    int64_t np2 = 2048;
    while (np2 < n_rows) np2 <<= 1;
    const int tile = (int)((np2 < REL_TILE) ? np2 : REL_TILE);
    const int64_t batch = rel_batch_columns(np2, n_cols);
    double *dX = nullptr, *dkeys = nullptr, *dstat = nullptr;
    uint32_t *didx = nullptr;
    int *dcnt = nullptr;
    double *Xd = X;
    int64_t ldd = ld;   // leading dimension of the matrix the kernels work on
    REL_HIP(hipSetDevice(device));
    // a host matrix may be a row-strided VIEW (ld > n_cols) that ends with its last row: only the n_cols-wide rows are
    // the caller's -- the device copy is dense (leading dimension n_cols) and the columns between the rows are never
    // read nor written back
    if (space == TSFA_HOST) {
        ldd = n_cols;
        REL_HIP(hipMalloc((void **)&dX, (size_t)n_rows * n_cols * sizeof(double)));
        REL_HIP(hipMemcpy2D(dX, (size_t)n_cols * sizeof(double), X, (size_t)ld * sizeof(double), (size_t)n_cols * sizeof(double),
                            (size_t)n_rows, hipMemcpyHostToDevice));
        Xd = dX;
    }
    REL_HIP(hipMalloc((void **)&dkeys, (size_t)batch * np2 * sizeof(double)));
    REL_HIP(hipMalloc((void **)&didx, (size_t)batch * np2 * sizeof(uint32_t)));
    REL_HIP(hipMalloc((void **)&dcnt, (size_t)n_cols * sizeof(int)));
    REL_HIP(hipMalloc((void **)&dstat, (size_t)3 * n_cols * sizeof(double)));
    {
        const size_t lds = (size_t)tile * (sizeof(double) + sizeof(uint32_t));
        REL_HIP(hipFuncSetAttribute((const void *)k_rel_sort, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        for (int64_t c0 = 0; c0 < n_cols; c0 += batch) {
            const int64_t nb = (n_cols - c0 < batch) ? (n_cols - c0) : batch;
            k_imp_stage<<<dim3((unsigned)nb), REL_NT, 0, 0>>>(Xd, n_rows, ldd, c0, dkeys, didx, np2, dcnt);
            k_rel_sort<<<dim3((unsigned)nb), REL_NT, lds, 0>>>(dkeys, didx, np2, tile);
            k_imp_stats<<<dim3((unsigned)((nb + 63) / 64)), 64, 0, 0>>>(dkeys, np2, dcnt, c0, nb, dstat, dstat + n_cols, dstat + 2 * n_cols);
            REL_HIP(hipGetLastError());
        }
        k_imp_apply<<<2048, 256, 0, 0>>>(Xd, n_rows, ldd, n_cols, dstat, dstat + n_cols, dstat + 2 * n_cols);
        REL_HIP(hipGetLastError());
    }
    if (col_max) REL_HIP(hipMemcpy(col_max, dstat, (size_t)n_cols * sizeof(double), hipMemcpyDeviceToHost));
    if (col_min) REL_HIP(hipMemcpy(col_min, dstat + n_cols, (size_t)n_cols * sizeof(double), hipMemcpyDeviceToHost));
    if (col_median) REL_HIP(hipMemcpy(col_median, dstat + 2 * n_cols, (size_t)n_cols * sizeof(double), hipMemcpyDeviceToHost));
    if (finite_count) REL_HIP(hipMemcpy(finite_count, dcnt, (size_t)n_cols * sizeof(int), hipMemcpyDeviceToHost));
    if (space == TSFA_HOST)
        REL_HIP(hipMemcpy2D(X, (size_t)ld * sizeof(double), dX, (size_t)n_cols * sizeof(double), (size_t)n_cols * sizeof(double),
                            (size_t)n_rows, hipMemcpyDeviceToHost));
    REL_HIP(hipDeviceSynchronize());
done:
    (void)hipFree(dX); (void)hipFree(dkeys); (void)hipFree(didx); (void)hipFree(dcnt); (void)hipFree(dstat);
    return rc;
}

// ---------------------------------------------------------------------------------------------
// Device-resident matrices: extract -> impute -> select without the feature matrix crossing PCIe
// between the steps.  tsfa_extract / tsfa_impute / tsfa_relevance_* take TSFA_DEVICE pointers; these four entry points
// give a host in any language the buffer, the transfers and the final gather of the selected columns.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_gather_columns(const double *__restrict__ X, int64_t n_rows, int64_t ld,
                                                         const int32_t *__restrict__ cols, int64_t n_sel,
                                                         double *__restrict__ out) {
    const int64_t total = n_rows * n_sel;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / n_sel, c = i - r * n_sel;
        out[i] = X[r * ld + cols[c]];
    }
}

__global__ void __launch_bounds__(256) k_scatter_columns(double *__restrict__ dst, int64_t ld_dst, const int32_t *__restrict__ cols,
                                                          const double *__restrict__ src, int64_t n_rows, int64_t n_src) {
    const int64_t total = n_rows * n_src;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / n_src, c = i - r * n_src;
        dst[r * ld_dst + cols[c]] = src[i];
    }
}

static int rel_check_device(int32_t device, const char *who) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || device < 0 || device >= ndev)
        return tsfa_fail(TSFA_ERR_NO_DEVICE, (std::string(who) + ": no such HIP device").c_str());
    return TSFA_OK;
}

extern "C" int tsfa_device_alloc(void **ptr, size_t bytes, int32_t device) {
    if (!ptr) return tsfa_fail(TSFA_ERR_INVALID, "tsfa_device_alloc: ptr is NULL");
    *ptr = nullptr;
    int rc = rel_check_device(device, "tsfa_device_alloc");
    if (rc) return rc;
    REL_HIP(hipSetDevice(device));
    REL_HIP(hipMalloc(ptr, bytes ? bytes : 1));
done:
    return rc;
}

extern "C" int tsfa_device_free(void *ptr, int32_t device) {
    if (!ptr) return TSFA_OK;
    int rc = rel_check_device(device, "tsfa_device_free");
    if (rc) return rc;
    REL_HIP(hipSetDevice(device));
    REL_HIP(hipFree(ptr));
done:
    return rc;
}

extern "C" int tsfa_device_copy(void *dst, const void *src, size_t bytes, int32_t to_device, int32_t device) {
    if ((!dst || !src) && bytes) return tsfa_fail(TSFA_ERR_INVALID, "tsfa_device_copy: null pointer");
    int rc = rel_check_device(device, "tsfa_device_copy");
    if (rc || !bytes) return rc;
    REL_HIP(hipSetDevice(device));
    REL_HIP(hipMemcpy(dst, src, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost));
done:
    return rc;
}

extern "C" int tsfa_gather_columns(const double *X, int64_t n_rows, int64_t ld, const int32_t *cols, int64_t n_sel,
                                   double *out_host, int32_t device) {
    if (n_rows < 0 || n_sel < 0 || ((n_rows && n_sel) && (!X || !cols || !out_host)))
        return tsfa_fail(TSFA_ERR_INVALID, "tsfa_gather_columns: null pointer or bad shape");
    int rc = rel_check_device(device, "tsfa_gather_columns");
    if (rc || n_rows == 0 || n_sel == 0) return rc;
    for (int64_t c = 0; c < n_sel; ++c)
        if (cols[c] < 0 || cols[c] >= ld) return tsfa_fail(TSFA_ERR_INVALID, "tsfa_gather_columns: column index out of range");
    int32_t *dcols = nullptr;
    double *dout = nullptr;
    REL_HIP(hipSetDevice(device));
    REL_HIP(hipMalloc((void **)&dcols, (size_t)n_sel * sizeof(int32_t)));
    REL_HIP(hipMalloc((void **)&dout, (size_t)n_rows * n_sel * sizeof(double)));
    REL_HIP(hipMemcpy(dcols, cols, (size_t)n_sel * sizeof(int32_t), hipMemcpyHostToDevice));
    k_gather_columns<<<2048, 256, 0, 0>>>(X, n_rows, ld, dcols, n_sel, dout);
    REL_HIP(hipGetLastError());
    REL_HIP(hipMemcpy(out_host, dout, (size_t)n_rows * n_sel * sizeof(double), hipMemcpyDeviceToHost));
done:
    (void)hipFree(dcols); (void)hipFree(dout);
    return rc;
}

extern "C" int tsfa_scatter_columns(double *dst, int64_t ld_dst, const int32_t *cols, const double *src, int64_t n_rows,
                                    int64_t n_src, int32_t device) {
    if (n_rows < 0 || n_src < 0 || ((n_rows && n_src) && (!dst || !cols || !src)))
        return tsfa_fail(TSFA_ERR_INVALID, "tsfa_scatter_columns: null pointer or bad shape");
    int rc = rel_check_device(device, "tsfa_scatter_columns");
    if (rc || n_rows == 0 || n_src == 0) return rc;
    for (int64_t c = 0; c < n_src; ++c)
        if (cols[c] < 0 || cols[c] >= ld_dst) return tsfa_fail(TSFA_ERR_INVALID, "tsfa_scatter_columns: column index out of range");
    int32_t *dcols = nullptr;
    REL_HIP(hipSetDevice(device));
    REL_HIP(hipMalloc((void **)&dcols, (size_t)n_src * sizeof(int32_t)));
    REL_HIP(hipMemcpy(dcols, cols, (size_t)n_src * sizeof(int32_t), hipMemcpyHostToDevice));
    k_scatter_columns<<<2048, 256, 0, 0>>>(dst, ld_dst, dcols, src, n_rows, n_src);
    REL_HIP(hipGetLastError());
    REL_HIP(hipDeviceSynchronize());
done:
    (void)hipFree(dcols);
    return rc;
}

// Pr(D >= h / lcm(m, n)) for the two-sided two-sample Kolmogorov-Smirnov statistic, m != n: the proportion of lattice
// paths (0,0) -> (m,n) that do not stay strictly inside |x/m - y/n| < h/lcm (Hodges 1958; the column recurrence scipy
// uses in ks_2samp(method="exact"), computed on the complement so small probabilities keep their relative accuracy).
// A scalar function of four integers: no data passes through it.
extern "C" double tsfa_ks_outer_prob(int64_t m, int64_t n, int64_t g, int64_t h) {
    if (m < n) { const int64_t t = m; m = n; n = t; }
    const int64_t mg = m / g, ng = n / g;
    int64_t minj = 0, maxj = (h + mg - 1) / mg;
    if (maxj > n + 1) maxj = n + 1;
    int64_t curlen = maxj - minj;
    int64_t lenA = 2 * maxj + 2;
    if (lenA > n + 1) lenA = n + 1;
    std::vector<double> A((size_t)lenA + 2, 1.0);
    for (int64_t j = minj; j < maxj; ++j) A[(size_t)j] = 0.0;
    for (int64_t i = 1; i <= m; ++i) {
        const int64_t lastminj = minj, lastlen = curlen;
        // floor((ng * i - h) / mg) + 1 and ceil((ng * i + h) / mg) in exact integer arithmetic
        const int64_t num = ng * i - h;
        int64_t fl = num / mg;
        if (num % mg != 0 && num < 0) --fl;
        minj = fl + 1;
        if (minj < 0) minj = 0;
        if (minj > n) minj = n;
        maxj = (ng * i + h + mg - 1) / mg;
        if (maxj > n + 1) maxj = n + 1;
        if (maxj <= minj) return 1.0;
        double val = (minj == 0) ? 0.0 : 1.0;
        for (int64_t jj = 0; jj < maxj - minj; ++jj) {
            const int64_t j = jj + minj;
            val = (A[(size_t)(jj + minj - lastminj)] * (double)i + val * (double)j) / (double)(i + j);
            A[(size_t)jj] = val;
        }
        curlen = maxj - minj;
        if (lastlen > curlen)
            for (int64_t q = maxj - minj; q < maxj - minj + (lastlen - curlen); ++q) A[(size_t)q] = 1.0;
    }
    return A[(size_t)(maxj - minj - 1)];
}
