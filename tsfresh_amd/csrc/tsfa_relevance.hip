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

#define REL_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) { rc = tsfa_fail(TSFA_ERR_HIP, (std::string(#call) + ": " + hipGetErrorString(e_)).c_str()); goto done; } \
    } while (0)

extern "C" int tsfa_relevance_classes(const double *X, int64_t n_rows, int64_t n_cols, int64_t ld, int32_t space,
                                      const int32_t *y_codes, int32_t n_classes, int32_t device, tsfa_relevance_col *cols,
                                      double *rank_sums, int64_t *hi_counts) {
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
    // columns per batch: bound the sort scratch (12 B per padded row and column) to ~4 GB
    int64_t batch = (int64_t)(4.0e9 / (12.0 * (double)np2));
    if (batch < 1) batch = 1;
    if (batch > n_cols) batch = n_cols;
    double *dX = nullptr, *dkeys = nullptr, *drs = nullptr;
    uint32_t *didx = nullptr;
    int32_t *dy = nullptr;
    int64_t *dhc = nullptr;
    tsfa_relevance_col *dcols = nullptr;
    const double *Xd = X;
    REL_HIP(hipSetDevice(device));
    if (space == TSFA_HOST) {
        REL_HIP(hipMalloc((void **)&dX, (size_t)n_rows * ld * sizeof(double)));
        REL_HIP(hipMemcpy(dX, X, (size_t)n_rows * ld * sizeof(double), hipMemcpyHostToDevice));
        Xd = dX;
    }
    REL_HIP(hipMalloc((void **)&dkeys, (size_t)batch * np2 * sizeof(double)));
    REL_HIP(hipMalloc((void **)&didx, (size_t)batch * np2 * sizeof(uint32_t)));
    REL_HIP(hipMalloc((void **)&dy, (size_t)n_rows * sizeof(int32_t)));
    REL_HIP(hipMalloc((void **)&drs, (size_t)n_cols * n_classes * sizeof(double)));
    REL_HIP(hipMalloc((void **)&dhc, (size_t)n_cols * n_classes * sizeof(int64_t)));
    REL_HIP(hipMalloc((void **)&dcols, (size_t)n_cols * sizeof(tsfa_relevance_col)));
    REL_HIP(hipMemcpy(dy, y_codes, (size_t)n_rows * sizeof(int32_t), hipMemcpyHostToDevice));
    {
        const size_t lds = (size_t)tile * (sizeof(double) + sizeof(uint32_t));
        REL_HIP(hipFuncSetAttribute((const void *)k_rel_sort, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        for (int64_t c0 = 0; c0 < n_cols; c0 += batch) {
            const int64_t nb = (n_cols - c0 < batch) ? (n_cols - c0) : batch;
            k_rel_stage<<<dim3((unsigned)nb), REL_NT, 0, 0>>>(Xd, n_rows, ld, c0, dkeys, didx, np2);
            k_rel_sort<<<dim3((unsigned)nb), REL_NT, lds, 0>>>(dkeys, didx, np2, tile);
            k_rel_stats<<<dim3((unsigned)nb), REL_NT, 0, 0>>>(dkeys, didx, np2, n_rows, dy, n_classes, c0, dcols, drs, dhc);
            REL_HIP(hipGetLastError());
        }
    }
    REL_HIP(hipMemcpy(cols, dcols, (size_t)n_cols * sizeof(tsfa_relevance_col), hipMemcpyDeviceToHost));
    REL_HIP(hipMemcpy(rank_sums, drs, (size_t)n_cols * n_classes * sizeof(double), hipMemcpyDeviceToHost));
    REL_HIP(hipMemcpy(hi_counts, dhc, (size_t)n_cols * n_classes * sizeof(int64_t), hipMemcpyDeviceToHost));
done:
    (void)hipFree(dX); (void)hipFree(dkeys); (void)hipFree(didx); (void)hipFree(dy); (void)hipFree(drs); (void)hipFree(dhc); (void)hipFree(dcols);
    return rc;
}
