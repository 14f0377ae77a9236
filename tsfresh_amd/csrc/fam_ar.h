// Family AR: calculators whose arithmetic lives in statsmodels in the reference (fc.py:387, 440, 499, 1459):
// adjusted autocovariance, Levinson-Durbin PACF, conditional-OLS AR(k), augmented Dickey-Fuller with AIC lag
// selection.  Restated from statsmodels/tsa/stattools.py (acovf, acf, pacf, levinson_durbin, adfuller, _autolag),
// statsmodels/tsa/ar_model.py (AutoReg = OLS on lagged regressors) and statsmodels/tsa/adfvalues.py (mackinnonp).
//
// Both regressions have LAGGED designs: every regressor (and the target) is the same sequence s shifted by j.
// Their normal matrix T(i, j) = sum_t s[t-i] s[t-j] therefore satisfies
//     T(i+1, j+1) = T(i, j) + s[t0-1-i] s[t0-1-j] - s[t1-1-i] s[t1-1-j]
// so only the first column (one wavefront reduction per lag) is summed; every other entry is an O(1) update
// along its diagonal (one lane per diagonal).  ADF needs 2*maxlag+5 reductions instead of (maxlag+3)^2/2.
#ifndef TSFA_FAM_AR_H
#define TSFA_FAM_AR_H

#include "tsfa_common.h"

// ADF regressors at series length n: const, level, maxlag lagged differences (+ the target as "lag 0")
TSFA_DEV int adf_maxlag_for(int n) {
    int maxlag = (int)ceil(12.0 * pow((double)n / 100.0, 0.25));
    const int cap = n / 2 - 1 - 1;
    if (cap < maxlag) maxlag = cap;
    return maxlag;
}

// Cholesky factorization G = L L^T (lower triangle, in place, leading dimension ld).  Serial.
TSFA_DEV bool chol_factor(double *G, int p, int ld) {
    for (int j = 0; j < p; ++j) {
        double d = G[j + j * ld];
        for (int k = 0; k < j; ++k) d -= G[j + k * ld] * G[j + k * ld];
        if (!(d > 0.0)) return false;
        d = sqrt(d);
        G[j + j * ld] = d;
        for (int i = j + 1; i < p; ++i) {
            double s = G[i + j * ld];
            for (int k = 0; k < j; ++k) s -= G[i + k * ld] * G[j + k * ld];
            G[i + j * ld] = s / d;
        }
    }
    return true;
}
TSFA_DEV void chol_solve(const double *L, int p, int ld, const double *rhs, double *x) {
    for (int i = 0; i < p; ++i) {
        double s = rhs[i];
        for (int k = 0; k < i; ++k) s -= L[i + k * ld] * x[k];
        x[i] = s * (1.0 / L[i + i * ld]);   // reciprocal pivots, as in blk_chol_solve: one division per pivot, not per use
    }
    for (int i = p - 1; i >= 0; --i) {
        double s = x[i];
        for (int k = i + 1; k < p; ++k) s -= L[k + i * ld] * x[k];
        x[i] = s * (1.0 / L[i + i * ld]);
    }
}

// x = (L L^T)^-1 rhs for p <= 64, called by every thread.  GPU: lane i of every wavefront holds entry i; the two
// triangular solves are column-oriented (x_k is broadcast with readlane, every lane below / above subtracts its
// L entry times x_k), so a solve costs 2p dependent steps instead of p^2 dependent LDS reads on one thread; all
// wavefronts compute the same result, thread 0 stores it.  The caller provides the barriers around it.
TSFA_DEV void blk_chol_solve(const Blk &b, const double *L, int p, int ld, const double *rhs, double *x) {
#if TSFA_GPU
    const int lane = b.tid & 63;
    const bool live = lane < p;
    double s = live ? rhs[lane] : 0.0;
    // a float64 division is ~30 instructions and the family is issue bound: one reciprocal per pivot (lane), then products
    const double dg = live ? 1.0 / L[lane + lane * ld] : 1.0;
    for (int k = 0; k < p; ++k) {  // forward: L w = rhs
        const double wk = readlane_f64(s, k) * readlane_f64(dg, k);
        if (lane == k) s = wk;
        else if (live && lane > k) s -= L[lane + k * ld] * wk;
    }
    for (int k = p - 1; k >= 0; --k) {  // backward: L^T x = w
        const double xk = readlane_f64(s, k) * readlane_f64(dg, k);
        if (lane == k) s = xk;
        else if (live && lane < k) s -= L[k + lane * ld] * xk;
    }
    if (b.tid < p) x[b.tid] = s;
#else
    if (b.tid == 0) chol_solve(L, p, ld, rhs, x);
#endif
}

// Workgroup-cooperative Cholesky (right-looking): after step j, column j holds L(:, j) and the trailing block has had
// L(:, j) L(:, j)^T subtracted.  Every entry receives exactly the subtractions of the serial left-looking loop, in the
// same order (k ascending), so the factor is bit-identical to chol_factor's; only the p-1 dependent steps remain
// serial.
// Returns false (uniformly) when a pivot keeps less than TSFA_AR_PIVOT_TOL of its column's squared norm (diag0
// receives the diagonal of G): the design is rank-deficient (constant / linear / periodic series) or conditioned
// worse than ~1e3, where float64 normal equations lose the digits the reference's SVD still has.  Such series are
// listed for the double-double second pass (fam_ar_dd.h); an absolute `d > 0` test lets round-off pass for a pivot.
#define TSFA_AR_PIVOT_TOL 1e-6   // (1e-9 until round 3: a noiseless float32 sine + offset keeps 16 pivots at 1.5e-9 of their columns,
                                 //  the float64 factor then loses the ADF statistic's 4th digit -- found by the fuzz, adjudicated in 60 digits)
TSFA_DEV bool blk_chol_factor(const Blk &b, double *G, int p, int ld, double *diag0, double *dmin = nullptr) {
    for (int a = b.tid; a < p; a += b.nt) diag0[a] = G[a + a * ld];
    double dm = TSFA_INF;
    for (int j = 0; j < p; ++j) {
        blk_sync();
        const double d = G[j + j * ld];
        if (!(d > TSFA_AR_PIVOT_TOL * diag0[j])) return false;
        dm = fmin(dm, d);
        if (dmin) *dmin = dm;   // uniform: the smallest pivot so far
        const double sd = sqrt(d);
        blk_sync();
        const double rsd = 1.0 / sd;
        for (int i = j + b.tid; i < p; i += b.nt) G[i + j * ld] = (i == j) ? sd : G[i + j * ld] * rsd;
        blk_sync();
        // trailing update, (row, column) = (tid % 32, tid / 32) strides: shifts instead of an integer division and a
        // modulo per element (the family is VALU-issue bound, and those two cost ~80 instructions)
        const int rsh = (b.nt >= 32) ? 5 : 0;  // nt is 1 (emulation) or a multiple of 64
        for (int i = j + 1 + (b.tid & ((1 << rsh) - 1)); i < p; i += (1 << rsh)) {
            const double lij = G[i + j * ld];
            for (int k = j + 1 + (b.tid >> rsh); k <= i; k += (b.nt >> rsh))
                G[i + k * ld] = G[i + k * ld] - lij * G[k + j * ld];
        }
    }
    blk_sync();
    return true;
}
// w = L^-1 rhs by column-oriented forward substitution (same subtraction order as the serial row loop).
// `w` is overwritten in place: pass a copy of rhs.  GPU (p <= 64): lane i of every wavefront holds w_i, w_j travels by
// readlane -- p dependent steps without a barrier (every wavefront computes the same values, thread i < p stores).
TSFA_DEV void blk_chol_forward(const Blk &b, const double *L, int p, int ld, double *w) {
#if TSFA_GPU
    if (p <= 64) {
        const int lane = b.tid & 63;
        const bool live = lane < p;
        double s = live ? w[lane] : 0.0;
        const double dg = live ? 1.0 / L[lane + lane * ld] : 1.0;
        for (int k = 0; k < p; ++k) {
            const double wk = readlane_f64(s, k) * readlane_f64(dg, k);
            if (lane == k) s = wk;
            else if (live && lane > k) s = s - L[lane + k * ld] * wk;
        }
        blk_sync();
        if (b.tid < p) w[b.tid] = s;
        blk_sync();
        return;
    }
#endif
    for (int i = 0; i < p; ++i) {
        blk_sync();
        const double wi = w[i] * (1.0 / L[i + i * ld]);
        blk_sync();
        if (b.tid == 0) w[i] = wi;
        for (int k = i + 1 + b.tid; k < p; k += b.nt) w[k] = w[k] - L[k + i * ld] * wi;
    }
    blk_sync();
}

// (Measured and dropped, round 3: the same factorization by ONE wavefront with the matrix in registers -- lane = row,
//  columns by readlane, no barrier.  Fewer instructions, but the other wavefront of the workgroup idles through p
//  dependent sqrt / reciprocal chains: k_ar 5.90 -> 6.56 ms per 100 000 series.)

// ---------------------------------------------------------------------------------------------
// Lagged sums on register tiles.  Every sum of the family has the form  S[j] = sum_t A(t) B(t - j)  over a row range,
// j = 0 .. 60.  A thread owns EIGHT CONSECUTIVE rows (a tile: A(tb .. tb+7) in registers); the eight lags of a sweep then
// need B(tb - j0 - 8 .. tb - j0 + 7), two aligned blocks of eight, of which the upper one is the previous sweep's lower
// one.  A sweep is 64 fused multiply-adds on registers plus ONE block load (two 128-bit LDS reads for float32 samples);
// the strided form it replaces (row = tid + u nt) spent ten instructions per product on index clamps, conversions,
// centring and selects.  The series sits in LDS between TSFA_AR_PADL zero samples and TSFA_AR_PADR zero samples, so every
// block a tile can ask for exists and is finite: rows outside the sum are masked in A alone.
// ---------------------------------------------------------------------------------------------
#define TSFA_AR_PADL 64   // >= 8 * ceil((max lag + 1) / 8): ADF maxlag <= 61 at n = 65 535, agg_autocorrelation <= 60
#define TSFA_AR_PADR 16   // tiles cover [0, 8 ceil(n / 8)); a block of differences reads one sample more

template <class ST>
TSFA_DEV void ar_load8(const ST *p, double (&r)[8]) {   // p: block of eight samples, index a multiple of 8
#if TSFA_GPU
    p = (const ST *)__builtin_assume_aligned(p, 16);
#endif
#pragma unroll
    for (int q = 0; q < 8; ++q) r[q] = (double)p[q];
}

// the resident series as blocks: y = centred samples (double)x - mean, d = first differences x[i + 1] - x[i] of the RAW
// samples (np.diff(x): the mean cancels, and for float32 input the difference of two samples is exact in float64)
template <class ST>
struct ArBlocks {
    const ST *xs;   // sample 0 (TSFA_AR_PADL zero samples before it)
    double mean;
    TSFA_MEM void y8(int i0, double (&r)[8]) const {
        ar_load8(xs + i0, r);
#pragma unroll
        for (int q = 0; q < 8; ++q) r[q] -= mean;
    }
    TSFA_MEM void d8(int i0, double (&r)[8]) const {
        double v[8];
        ar_load8(xs + i0, v);
        const double nx = (double)xs[i0 + 8];
#pragma unroll
        for (int q = 0; q < 7; ++q) r[q] = v[q + 1] - v[q];
        r[7] = nx - v[7];
    }
};

// rows outside [r0, r1) of the tile starting at tb read as zero
TSFA_DEV void tile_mask_rows(double (&A)[8], int tb, int r0, int r1) {
#pragma unroll
    for (int u = 0; u < 8; ++u) A[u] = (tb + u >= r0 && tb + u < r1) ? A[u] : 0.0;
}

// s[j] += sum_u A[u] B(tb + u - j0 - j), j < 8:  hi = B(tb - j0 ..), lo = B(tb - j0 - 8 ..)
TSFA_DEV void tile_dots8(const double (&A)[8], const double (&lo)[8], const double (&hi)[8], double (&s)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int u = 0; u < 8; ++u) s[j] = fma(A[u], (u >= j) ? hi[u - j] : lo[u - j + 8], s[j]);
    }
}
// r[u] -= sum_j cf[j] B(tb + u - j0 - j): the same window, transposed (residuals of a lagged regression)
TSFA_DEV void tile_axpy8(const double (&cf)[8], const double (&lo)[8], const double (&hi)[8], double (&r)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int u = 0; u < 8; ++u) r[u] = fma(-cf[j], (u >= j) ? hi[u - j] : lo[u - j + 8], r[u]);
    }
}

// store(j, sum_t A(t) B(t - j)) for j < nlags, rows t in [0, nrows):
//   loadA(tb, A): A(tb .. tb + 7), zero on the rows that are not part of the sum
//   loadB(i0, B): B(i0 .. i0 + 7), i0 a multiple of 8, >= -TSFA_AR_PADL
//   sums (or null): sums[0] = sum_t A(t), sums[1] = sum_t A(t)^2  (every thread receives them)
//   part (or null): LDS scratch of part_doubles doubles.  With it the wavefronts leave their partial sums of ALL sweeps
//                   there and meet once (two barriers per call instead of two per sweep; lane = lag adds them up).
// store() is called by one thread per lag; the caller provides the barrier before anybody else reads what it wrote.
// One tile per thread (n <= 8 nt, the short-series launch): A and the upper block stay in registers across the sweeps.
#define TSFA_AR_PART_STRIDE 72   // 64 lags + the two row sums, per wavefront
template <class LA, class LB, class SF>
TSFA_DEV void blk_tile_lagdots(const Blk &b, int nrows, int nlags, LA loadA, LB loadB, SF store, double *sums = nullptr,
                               double *part = nullptr, int part_doubles = 0) {
    const int ntiles = (nrows + 7) >> 3;
    const bool single = ntiles <= b.nt;
#if TSFA_GPU
    const int nwv = (b.nt + 63) >> 6, wv = b.tid >> 6, lane = b.tid & 63;
#else
    const int nwv = 1, wv = 0;
#endif
    const bool deferred = (part != nullptr) && (nwv * TSFA_AR_PART_STRIDE <= part_doubles) && (nlags <= 64);
    if (deferred) blk_sync();   // the previous user of `part`
    double A[8], lo[8], hi[8];
    for (int j0 = 0; j0 < nlags; j0 += 8) {
        double s[8], sa[2] = {0.0, 0.0};
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = 0.0;
        for (int tile = b.tid; tile < ntiles; tile += b.nt) {
            const int tb = tile << 3;
            if (single && j0 > 0) {
#pragma unroll
                for (int q = 0; q < 8; ++q) hi[q] = lo[q];
            } else {
                loadA(tb, A);
                loadB(tb - j0, hi);
            }
            loadB(tb - j0 - 8, lo);
            tile_dots8(A, lo, hi, s);
            if (sums != nullptr && j0 == 0) {
#pragma unroll
                for (int u = 0; u < 8; ++u) { sa[0] += A[u]; sa[1] = fma(A[u], A[u], sa[1]); }
            }
        }
        if (deferred) {
            double *slot = part + wv * TSFA_AR_PART_STRIDE;
#if TSFA_GPU
            const double held = wave_reduce_scatter<8>(s);
            if (lane < 8) slot[j0 + wave_reduce_scatter_index<8>(lane)] = held;
            if (sums != nullptr && j0 == 0) {
                sa[0] = wave_sum(sa[0]);
                sa[1] = wave_sum(sa[1]);
                if (lane == 0) { slot[64] = sa[0]; slot[65] = sa[1]; }
            }
#else
#pragma unroll
            for (int j = 0; j < 8; ++j) slot[j0 + j] = s[j];
            if (sums != nullptr && j0 == 0) { slot[64] = sa[0]; slot[65] = sa[1]; }
#endif
            continue;
        }
        blk_sum_multi<8>(b, s);
        if (b.tid == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j0 + j < nlags) store(j0 + j, s[j]);
        }
        if (sums != nullptr && j0 == 0) {
            blk_sum_multi<2>(b, sa);
            sums[0] = sa[0];
            sums[1] = sa[1];
        }
    }
    if (deferred) {
        blk_sync();
        for (int j = b.tid; j < nlags; j += b.nt) {
            double v = part[j];
            for (int w = 1; w < nwv; ++w) v += part[w * TSFA_AR_PART_STRIDE + j];
            store(j, v);
        }
        if (sums != nullptr) {
            double v0 = part[64], v1 = part[65];
            for (int w = 1; w < nwv; ++w) { v0 += part[w * TSFA_AR_PART_STRIDE + 64]; v1 += part[w * TSFA_AR_PART_STRIDE + 65]; }
            sums[0] = v0;
            sums[1] = v1;
        }
    }
}

// Lag-product matrix of sequence s over rows t in [t0, t1):  T[i + j*ld] = sum_t s(t-i) s(t-j), 0 <= j <= i <= Lg,
// and column sums C[j] = sum_t s(t-j), from the first column T[0 .. Lg] and C[0] (blk_tile_lagdots; thread 0 wrote
// them).  Requires t0 >= Lg.  S(u) returns s[u].
template <class S>
TSFA_DEV void blk_lag_products_rest(const Blk &b, S s, int Lg, int t0, int t1, double *T, int ld, double *C) {
    if (b.tid == 0) {
        for (int j = 0; j < Lg; ++j) C[j + 1] = C[j] + s(t0 - 1 - j) - s(t1 - 1 - j);
    }
    blk_sync();
    for (int dg = b.tid; dg <= Lg; dg += b.nt) {  // one lane per diagonal i - j = dg
        double v = T[dg];
        for (int j = 0; dg + j + 1 <= Lg; ++j) {
            const int i = dg + j;
            v = v + s(t0 - 1 - i) * s(t0 - 1 - j) - s(t1 - 1 - i) * s(t1 - 1 - j);
            T[(i + 1) + (j + 1) * ld] = v;
        }
    }
    blk_sync();
}

// statsmodels.tsa.adfvalues.mackinnonp(teststat, regression="c", N=1)
TSFA_DEV double mackinnon_p_c1(double t) {
    if (t > 2.74) return 1.0;
    if (t < -18.83) return 0.0;
    double z;
    if (t <= -1.61) z = 2.1659 + t * (1.4412 + t * (3.8269 * 1e-2));
    else z = 1.7339 + t * (9.3202 * 1e-1 + t * (-1.2745 * 1e-1 + t * (-1.0368 * 1e-2)));
    return tsfa_norm_cdf(z);
}

// LDS scratch of the family, carved from `aw`:  P = max regressors + 1 (leading dimension of the matrices)
//   T (P*P) | G (P*P) | 7 vectors of P | acv 64 | res 16 | pac 48 | arres 40 | pacw 128
TSFA_DEV int ar_scratch_doubles(int P) { return 2 * P * P + 7 * P + 64 + 16 + 48 + 40 + 128; }

// Evaluate the AR specs of one series.
//   xv   : sample accessor
//   xc_raw : LDS, n + 2 elements of ST (the series in its input precision)
//   aw   : LDS, ar_scratch_doubles(P) doubles;  P >= max(adf_maxlag_for(n) + 3, max AR order + 2)
// The series stays in LDS in its INPUT precision ST (float32 samples: half the LDS, one more resident series per CU);
// the mean-centred value is formed where it is read -- (double)x - mean is the very expression that used to be stored.
// statsmodels' pinv drops singular values <= 1e-15 s_max of the RAW design (fam_ar_dd.h).  The first pass works on the
// centred series, whose design X_c is well conditioned however large the mean is; the raw one is X_c S with S = I +
// e_const mu^T, so  s_min(raw) >= s_min(X_c) / (1 + |mu|)  and  s_max(raw)^2 <= trace.  With s_min(X_c) ~ the square
// root of the smallest Cholesky pivot this bounds the raw ratio from below; a series whose bound is within 1000x of
// the cut is listed for the second pass, which measures the ratio and takes the eigen route if it has to
// (1e6 + N(0, 1) is listed, 1e5 + N(0, 1) is not).
// A factorization without pivoting does not reveal rank: a design can keep every relative pivot above 1e-2 and still have
// cond(X) = 2e7 (a stuck sensor after a noisy start: the lag columns differ from the constant in their first rows only;
// eleven pivots of ~0.05 each multiply to a determinant of 1e-11).  The float64 normal equations then lose eps cond^2 --
// the intercept came out 4e-7 off where the reference's SVD is good to 4e-9 (found by the fuzz).  What does reveal it is
// the refinement step both regressions take anyway: its correction IS the error of the first solve.  A correction above
// 1e-8 of the solution (both weighted by the column norms) lists the series for the double-double pass.
#define TSFA_AR_REFINE_TOL 1e-8
TSFA_DEV bool ar_refinement_suspect(const Blk &b, const double *beta, const double *corr, const double *diag0, int p) {
    double mb = 0.0, mc = 0.0;
#if TSFA_GPU
    // lane = regressor, the two maxima by a wavefront reduction (a maximum does not depend on the order: the same bits as the
    // loop below, which every lane used to walk -- a float64 square root per regressor, ~35 instructions each)
    for (int a = b.tid & 63; a < p; a += 64) {
        const double w = sqrt(diag0[a]);
        mb = fmax(mb, fabs(beta[a]) * w);
        mc = fmax(mc, fabs(corr[a]) * w);
    }
    mb = wave_max(mb);
    mc = wave_max(mc);
#else
    (void)b;
    for (int a = 0; a < p; ++a) {
        const double w = sqrt(diag0[a]);
        mb = fmax(mb, fabs(beta[a]) * w);
        mc = fmax(mc, fabs(corr[a]) * w);
    }
#endif
    return mc > TSFA_AR_REFINE_TOL * mb;
}

// AR orders above this also get a condition estimate (fam_ar_series, ar_coefficient).  12 until round 5 ("none seen at
// AR(10) / AR(12)" in the fuzz); a targeted batch of stuck-sensor designs then failed at AR(12) (1 of 60 series wrong in the
// first digit; round-4 ADVICE, tests/test_ar_stuck.py::test_stuck_sensor_designs_at_low_orders).  Not 0: for the AR(10) of
// the settings objects the estimate costs 0.5 ms per 100 000 series (k_ar 5.71 -> 6.22 ms, profiles/r05_z2) and 180
// stuck-sensor designs at each of AR(2) .. AR(10) pass without it.
#define TSFA_AR_COND_CHECK_K 10
#define TSFA_AR_RAW_RATIO 1e-12
TSFA_DEV bool ar_raw_design_suspect(double dmin, double mu_norm, double trace_raw) {
    return sqrt(dmin) < TSFA_AR_RAW_RATIO * (1.0 + mu_norm) * sqrt(trace_raw);
}

template <class ST>
struct ArCentred {
    const ST *p;
    double mean;
    TSFA_MEM double operator[](int i) const { return (double)p[i] - mean; }
};
// Returns the calculators the float64 normal equations could not serve (bit 0: ar_coefficient, bit 1:
// augmented_dickey_fuller); their columns are left NaN for the second pass (fam_ar_dd.h).
template <class ST, class X>
TSFA_DEV int fam_ar_series(const Blk &b, X xv, int n, const TsfaSpec *specs, int nspecs, double *out_row, void *xc_raw,
                            double *aw, int P, int hint_acf, int hint_pacf, int hint_adf, int n_loop = -1,
                            const double *stats = nullptr) {
    const int nloop = (n_loop >= 0) ? n_loop : nspecs;  // columns [nloop, nspecs): lane = column epilogue
    const double dn = (double)n;
    TSFA_TICKER(tk, 0);
    // x.mean() in numpy's summation order: statsmodels demeans with it, and on (near-)constant series the
    // autocovariances are pure round-off of x - x.mean(), so the order decides what comes out
    // (stats: the record k_basic left for this series, TSFA_STATS_*: the same sum in the same order)
    const double mean = stats ? stats[TSFA_STATS_MEAN] : np_sum(b, n, [=](int i) { return xv(i); }) / dn;
    blk_sync();
    ST *xs_lds = (ST *)xc_raw + TSFA_AR_PADL;   // zero samples on either side: the blocks of the register tiles
    for (int i = b.tid; i < n + TSFA_AR_PADL + TSFA_AR_PADR; i += b.nt) {
        const int j = i - TSFA_AR_PADL;
        ((ST *)xc_raw)[i] = (j >= 0 && j < n) ? (ST)xv(j) : (ST)0;
    }
    const ArCentred<ST> xc{xs_lds, mean};
    const ArBlocks<ST> blk{xs_lds, mean};
    blk_sync();
    double v0 = 0.0;
    for (int i = b.tid; i < n; i += b.nt) v0 += xc[i] * xc[i];
    const double var = blk_sum(b, v0) / dn;

    double *T = aw;               // lag products
    double *G = T + P * P;        // normal matrix / Cholesky factor
    double *g = G + P * P;        // rhs
    double *beta = g + P;
    double *tmp1 = beta + P;
    double *tmp2 = tmp1 + P;
    double *C = tmp2 + P;         // column sums
    double *V = C + P;            // level products (ADF)
    double *diag0 = V + P;        // diagonal of the matrix being factored (relative pivot test)
    double *acv = diag0 + P;      // 64
    double *res = acv + 64;       // 16
    double *pac = res + 16;       // 48
    double *arres = pac + 48;     // 40: cached AR solution
    double *pacw = arres + 40;    // 128: Levinson-Durbin columns
    const ArCentred<ST> xcc = xc;
    int degenerate = 0;

    TSFA_TICK(tk, b, 120);
    // largest agg_autocorrelation maxlag / partial_autocorrelation lag of the plan (-1: none) and whether ADF is
    // requested: tsfa_prepare_family on the host (a scan of the spec list here costs a scalar-load round trip per spec)
    int max_acf_lag = hint_acf, max_pacf_lag = hint_pacf;
    const bool need_adf = (hint_adf != 0);
    const int adf_mode = (hint_adf >> 1) & 3;   // TSFA_AUTOLAG_*: the plan's lag selection
    if (max_acf_lag > 60) max_acf_lag = 60;
    if (max_pacf_lag > 40) max_pacf_lag = 40;

    // ---- adjusted autocovariance acv[k] = sum_t xc[t] xc[t+k] / (n - k)  (stattools.acovf, adjusted=True) ----
    int pacf_maxlag = 0;
    if (max_pacf_lag >= 0 && n > 1) pacf_maxlag = (max_pacf_lag >= n / 2) ? (n / 2 - 1) : max_pacf_lag;  // fc.py:472-475
    int nacv = -1;
    if (max_acf_lag >= 0) nacv = (max_acf_lag < n - 1) ? max_acf_lag : (n - 1);
    if (pacf_maxlag > nacv) nacv = pacf_maxlag;
    if (nacv >= 0) {
        // sum_t y(t) y(t - k) over t in [0, n) with y = 0 before the series: the blocks below sample 0 read as zeros
        blk_tile_lagdots(b, n, nacv + 1,
                         [=](int tb, double (&A)[8]) { blk.y8(tb, A); tile_mask_rows(A, tb, 0, n); },
                         [=](int i0, double (&B)[8]) {
                             blk.y8(i0, B);
#pragma unroll
                             for (int q = 0; q < 8; ++q) B[q] = (i0 >= 0) ? B[q] : 0.0;
                         },
                         [=](int k, double v) { acv[k] = v / (double)(n - k); }, nullptr, G, P * P);
    }
    blk_sync();

    TSFA_TICK(tk, b, 121);
    // ---- PACF by Levinson-Durbin (stattools.levinson_durbin, isacov=True) ----
    if (max_pacf_lag >= 0) {
#if TSFA_GPU
        {
            // lane j holds phi[j] of the current order (j <= 40): the products of the dot are formed in parallel and
            // added in the serial order (readlane), phi[k - j] comes through the LDS crossbar; every wavefront computes
            // the same values, bit-identical to the serial recursion below.  (Serial on one lane: 21 k cycles per series.)
            const int lane = b.tid & 63;
            for (int k = b.tid; k <= max_pacf_lag; k += b.nt) pac[k] = TSFA_NAN;
            blk_sync();
            if (n > 1 && pacf_maxlag > 0) {
                const int ord = pacf_maxlag;
                const double a0 = acv[0], a1 = acv[1];
                const double p1v = a1 / a0;
                double phi = (lane == 1) ? p1v : 0.0;
                double sig = a0 - p1v * a1;
                if (b.tid == 0) pac[1] = p1v;
                for (int k = 2; k <= ord; ++k) {
                    const int back = k - lane;
                    const double ak = acv[(back >= 0 && back <= ord) ? back : 0];
                    const double prod = phi * ak;
                    double dot = 0.0;
                    for (int j = 1; j < k; ++j) dot += readlane_f64(prod, j);
                    const double pkk = (acv[k] - dot) / sig;
                    const double rev = __shfl(phi, (back >= 0 && back < 64) ? back : 0);
                    phi = (lane >= 1 && lane < k) ? phi - pkk * rev : ((lane == k) ? pkk : phi);
                    sig = sig * (1.0 - pkk * pkk);
                    if (b.tid == 0) pac[k] = pkk;
                }
                if (b.tid == 0) pac[0] = 1.0;
            }
        }
#else
        if (b.tid == 0) {
            for (int k = 0; k <= max_pacf_lag; ++k) pac[k] = TSFA_NAN;
            if (n > 1 && pacf_maxlag > 0) {
                const int ord = pacf_maxlag;
                // only the previous column of phi is ever read: keep two rolling columns
                double *prev = pacw, *cur = pacw + 42, *sig = pacw + 84;
                prev[1] = acv[1] / acv[0];
                pac[1] = prev[1];
                sig[1] = acv[0] - prev[1] * acv[1];
                for (int k = 2; k <= ord; ++k) {
                    double dot = 0.0;
                    for (int j = 1; j < k; ++j) dot += prev[j] * acv[k - j];
                    const double pkk = (acv[k] - dot) / sig[k - 1];
                    for (int j = 1; j < k; ++j) cur[j] = prev[j] - pkk * prev[k - j];
                    cur[k] = pkk;
                    sig[k] = sig[k - 1] * (1.0 - pkk * pkk);
                    pac[k] = pkk;
                    for (int j = 1; j <= k; ++j) prev[j] = cur[j];
                }
                pac[0] = 1.0;
            }
        }
#endif
        blk_sync();
    }

    TSFA_TICK(tk, b, 122);
    // ---- augmented Dickey-Fuller, regression="c", autolag="AIC" (stattools.adfuller) ----
    // rows are indexed by t with d[t] = x[t+1] - x[t]; regressors: 1, x[t] (level), d[t-1..t-maxlag]; target d[t]
    double adf_stat = TSFA_NAN, adf_p = TSFA_NAN, adf_lag = TSFA_NAN;
    if (need_adf) {
        const int maxlag = adf_maxlag_for(n);
        // The float64 fit holds one matrix entry per lane (p1 = maxlag + 2 <= 64: series up to 75 969 samples).  Longer series
        // -- maxlag = ceil(12 (n / 100)^(1/4)) keeps growing -- go to the second pass, whose double-double normal equations
        // take any p (fam_ar_dd.h; its matrices move to HBM beyond P = 64: tsfa_api.cpp).
        if (maxlag + 2 > 64) degenerate |= 2;
        else if (maxlag >= 0 && maxlag + 3 <= P) {
            auto dif = [=](int t) { return (double)xs_lds[t + 1] - (double)xs_lds[t]; };
            const int t0 = maxlag, t1 = n - 1;
            const double nobs = (double)(t1 - t0);
            // lag products of d (lag 0 = the target) and level products: first columns on register tiles
            auto d_blocks = [=](int i0, double (&B)[8]) { blk.d8(i0, B); };
            double sd[2], sl[2];
            blk_tile_lagdots(b, n, maxlag + 1,
                             [=](int tb, double (&A)[8]) { blk.d8(tb, A); tile_mask_rows(A, tb, t0, t1); }, d_blocks,
                             [=](int j, double v) { T[j] = v; }, sd, G, P * P);
            blk_tile_lagdots(b, n, maxlag + 1,
                             [=](int tb, double (&A)[8]) { blk.y8(tb, A); tile_mask_rows(A, tb, t0, t1); }, d_blocks,
                             [=](int j, double v) { V[j] = v; }, sl, G, P * P);
            if (b.tid == 0) C[0] = sd[0];
            blk_lag_products_rest(b, dif, maxlag, t0, t1, T, P, C);
            const double sx = sl[0], sxx = sl[1];
            // assemble the normal matrix in the autolag column order [const, level, d-lag1 .. d-lag maxlag]
            const int p1 = maxlag + 2;
            for (int e = b.tid; e < 64 * p1; e += b.nt) {  // (a, c) = (e % 64, e / 64): p1 <= 63 for n <= 65535
                const int a = e & 63, c = e >> 6;
                if (a >= p1 || c > a) continue;
                double v;
                if (a == 0) v = nobs;
                else if (a == 1) v = (c == 0) ? sx : sxx;
                else {
                    const int la = a - 1;
                    if (c == 0) v = C[la];
                    else if (c == 1) v = V[la];
                    else v = T[la + (c - 1) * P];
                }
                G[a + c * P] = v;
            }
            for (int a = b.tid; a < p1; a += b.nt) g[a] = (a == 0) ? C[0] : (a == 1 ? V[0] : T[a - 1]);
            blk_sync();
            const double yy = T[0];
            TSFA_TICK(tk, b, 123);
            {
                // all nested fits from one factorization: w = L^-1 g, SSR_p = yy - sum_{i<p} w_i^2
                double dmin1 = 0.0;
                const bool okf = blk_chol_factor(b, G, p1, P, diag0, &dmin1);
                if (!okf) degenerate |= 2;
                else {
                    // raw level column: sum (xc + mean)^2 over the rows; the lag columns are differences (no offset)
                    const double mu = xcc.mean;
                    const double tr = nobs + (sxx + 2.0 * mu * sx + nobs * mu * mu) + (double)maxlag * yy;
                    if (ar_raw_design_suspect(dmin1, fabs(mu), tr)) degenerate |= 2;
                }
                if (okf) {
                    for (int a = b.tid; a < p1; a += b.nt) tmp1[a] = g[a];
                    blk_chol_forward(b, G, p1, P, tmp1);
                    if (b.tid == 0) {  // running sum of squares in the serial order
                        double acc = 0.0;
                        for (int i = 0; i < p1; ++i) { acc += tmp1[i] * tmp1[i]; tmp2[i] = acc; }
                    }
                    blk_sync();
                    for (int i = b.tid; i < p1; i += b.nt) {  // AIC (BIC) of the fit with i + 1 columns
                        const int pcols = i + 1;
                        const double ssr = yy - tmp2[i];
                        const double llf = -0.5 * nobs * log(2.0 * M_PI) - 0.5 * nobs * log(ssr / nobs) - 0.5 * nobs;
                        const double wi = tmp1[i];
                        double ic = -2.0 * llf + ((adf_mode == TSFA_AUTOLAG_BIC) ? log(nobs) : 2.0) * (double)pcols;
                        // "t-stat": |t| of the LAST coefficient of that fit -- its value is w_i / L_ii and its variance
                        // sigma^2 / L_ii^2, so the Cholesky diagonal cancels
                        if (adf_mode == TSFA_AUTOLAG_TSTAT) ic = fabs(wi / sqrt(ssr / (nobs - (double)pcols)));
                        tmp1[i] = ic;
                    }
                    blk_sync();
                }
                if (b.tid == 0) {
                    int best = -1;
                    double best_aic = 0.0;
                    if (okf && adf_mode == TSFA_AUTOLAG_TSTAT) {
                        // from the largest lag down: the first fit whose last coefficient is significant, else the smallest
                        for (int i = p1 - 1; i >= 1; --i) {
                            best = i + 1;
                            if (tmp1[i] >= TSFA_AUTOLAG_TSTAT_STOP) break;
                        }
                    } else if (okf && adf_mode == TSFA_AUTOLAG_NONE) {
                        best = p1;
                    } else if (okf)
                        for (int i = 1; i < p1; ++i) {
                            const double aic = tmp1[i];
                            if (best < 0 || aic < best_aic) { best = i + 1; best_aic = aic; }
                        }
                    res[0] = (double)best;
                }
            }
            blk_sync();
            const int bestcols = (int)res[0];
            blk_sync();
            TSFA_TICK(tk, b, 124);
            if (bestcols >= 2) {
                const int usedlag = bestcols - 2;
                const int p2 = usedlag + 2;
                const int u0 = usedlag;
                const double nobs2 = (double)(t1 - u0);
                // final design [level, d-lag1..usedlag, const] over rows [usedlag, n-1): the autolag matrix restricted
                // to those columns plus the rows t in [usedlag, maxlag) that the first regression had trimmed
                auto reg2 = [=](int a, int t) { return a == 0 ? xcc[t] : (a == p2 - 1 ? 1.0 : dif(t - a)); };
                for (int e = b.tid; e < p2 * p2 + p2; e += b.nt) {
                    const bool is_rhs = e >= p2 * p2;
                    const int a = is_rhs ? e - p2 * p2 : e % p2;
                    const int c = is_rhs ? 0 : e / p2;
                    if (!is_rhs && c > a) continue;
                    double v;
                    // base value from the maxlag-row sums (same quantity, autolag ordering)
                    auto lagidx = [=](int q) { return q == 0 ? -2 : (q == p2 - 1 ? -1 : q); };  // -2 level, -1 const
                    const int A = lagidx(a), Cc = is_rhs ? 0 : lagidx(c);
                    if (is_rhs) {
                        v = (A == -2) ? V[0] : (A == -1 ? C[0] : T[A]);
                    } else {
                        const int hi = a, lo = c;  // a >= c in final ordering; map each unordered pair
                        (void)hi; (void)lo;
                        if (A == -1 && Cc == -1) v = nobs;
                        else if ((A == -1 && Cc == -2) || (A == -2 && Cc == -1)) v = sx;
                        else if (A == -2 && Cc == -2) v = sxx;
                        else if (A == -1) v = C[Cc];
                        else if (Cc == -1) v = C[A];
                        else if (A == -2) v = V[Cc];
                        else if (Cc == -2) v = V[A];
                        else v = (A >= Cc) ? T[A + Cc * P] : T[Cc + A * P];
                    }
                    for (int t = u0; t < t0; ++t) v += reg2(a, t) * (is_rhs ? dif(t) : reg2(c, t));
                    if (is_rhs) g[a] = v; else G[a + c * P] = v;
                }
                blk_sync();
                const bool ok2 = blk_chol_factor(b, G, p2, P, diag0);
                if (!ok2) degenerate |= 2;
                if (ok2) blk_chol_solve(b, G, p2, P, g, beta);
                blk_sync();
                if (ok2) {
                    // one step of iterative refinement on the true residuals, then SSR
                    // (the residual of a row is recomputed where it is used: a few multiply-adds instead of an
                    // n-double LDS buffer, which would cost the family two resident series per CU)
                    auto resid2 = [=](int t) {
                        double r = dif(t);
                        for (int c = 0; c < p2; ++c) r -= reg2(c, t) * beta[c];
                        return r;
                    };
                    for (int a0 = 0; a0 < p2; a0 += 8) {  // X^T r, eight regressors per sweep
                        double s8[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) s8[j] = 0.0;
                        for (int t = u0 + b.tid; t < t1; t += b.nt) {
                            const double rt = resid2(t);
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                if (a0 + j < p2) s8[j] += reg2(a0 + j, t) * rt;
                        }
                        blk_sum_multi<8>(b, s8);
                        if (b.tid == 0) {
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                if (a0 + j < p2) tmp1[a0 + j] = s8[j];
                        }
                    }
                    blk_sync();
                    blk_chol_solve(b, G, p2, P, tmp1, tmp2);
                    blk_sync();
                    if (ar_refinement_suspect(b, beta, tmp2, diag0, p2)) degenerate |= 2;
                    blk_sync();
                    for (int a = b.tid; a < p2; a += b.nt) beta[a] += tmp2[a];
                    blk_sync();
                    double ssr = 0.0;
                    for (int t = u0 + b.tid; t < t1; t += b.nt) {
                        double r = dif(t);
                        for (int c = 0; c < p2; ++c) r -= reg2(c, t) * beta[c];
                        ssr += r * r;
                    }
                    ssr = blk_sum(b, ssr);
                    for (int i = b.tid; i < p2; i += b.nt) tmp1[i] = (i == 0) ? 1.0 : 0.0;  // (X'X)^-1 [level, level]
                    blk_sync();
                    blk_chol_solve(b, G, p2, P, tmp1, tmp2);
                    blk_sync();
                    if (b.tid == 0) {
                        const double sigma2 = ssr / (nobs2 - (double)p2);
                        const double tstat = beta[0] / sqrt(sigma2 * tmp2[0]);
                        res[2] = tstat;
                        res[3] = mackinnon_p_c1(tstat);
                        res[4] = (double)usedlag;
                    }
                    blk_sync();
                    adf_stat = res[2];
                    adf_p = res[3];
                    adf_lag = res[4];
                    if (adf_stat != adf_stat) adf_p = TSFA_NAN;
                    blk_sync();
                }
            }
        }
    }

    int ar_cached_k = -1;
    bool ar_ok = false;
    TSFA_TICK(tk, b, 125);
    for (int s = 0; s < nloop; ++s) {
        const TsfaSpec sp = specs[s];
        double v = TSFA_NAN;
        switch (sp.calc) {
        case TSFA_C_AGG_AUTOCORRELATION: {                               // fc.py:387
            const int agg = (int)sp.p[0];
            const int ml = (int)sp.p[1];
            double r = TSFA_NAN;
#if TSFA_GPU
            {
                // lane = lag (at most 60 of them): one division per lane, mean / var by wave reductions, the median by
                // ranking every lag against the others with readlane broadcasts -- registers only, every wavefront
                // computes the same value.  (Serial on one lane this was ~100k cycles per series for the 3 columns.)
                int len = (ml < n - 1) ? ml : (n - 1);
                if (len > 60) len = 60;
                const int lane = b.tid & 63;
                const bool live = lane < len;
                if (fabs(var) < 1e-10 || n == 1) {
                    r = 0.0;
                } else if (len > 0) {
                    const double a0 = acv[0];
                    const double ak = live ? acv[lane + 1] / a0 : 0.0;
                    if (agg == TSFA_AGG_MEAN) {
                        r = wave_sum(ak) / (double)len;
                    } else if (agg == TSFA_AGG_VAR) {
                        const double m = wave_sum(ak) / (double)len;
                        const double d = live ? ak - m : 0.0;
                        r = wave_sum(d * d) / (double)len;
                    } else {
                        int rank = 0;
                        for (int j = 0; j < len; ++j) {
                            const double aj = readlane_f64(ak, j);
                            rank += (aj < ak || (aj == ak && j < lane)) ? 1 : 0;
                        }
                        const unsigned long long mlo = __ballot(live && rank == (len - 1) / 2);
                        const unsigned long long mhi = __ballot(live && rank == len / 2);
                        const double lo = readlane_f64(ak, __ffsll((long long)mlo) - 1);
                        const double hi = readlane_f64(ak, __ffsll((long long)mhi) - 1);
                        r = (len & 1) ? lo : (0.0 + lo + hi) / 2.0;
                    }
                }
            }
            v = r;
            break;
#endif
            if (b.tid == 0) {
                if (fabs(var) < 1e-10 || n == 1) {
                    r = 0.0;  // f_agg over zeros
                } else {
                    int len = (ml < n - 1) ? ml : (n - 1);  // a = acf[1:], a[:maxlag]
                    if (len > 60) len = 60;
                    double abuf[64];
                    double *a = abuf;
                    for (int k = 0; k < len; ++k) a[k] = acv[k + 1] / acv[0];
                    if (len <= 0) {
                        r = TSFA_NAN;
                    } else if (agg == TSFA_AGG_MEAN) {
                        r = np_leaf_sum(0, len, [=](int i) { return a[i]; }) / (double)len;
                    } else if (agg == TSFA_AGG_VAR) {
                        const double m = np_leaf_sum(0, len, [=](int i) { return a[i]; }) / (double)len;
                        r = np_leaf_sum(0, len, [=](int i) { const double d = a[i] - m; return d * d; }) / (double)len;
                    } else {  // median
                        for (int i = 1; i < len; ++i) {
                            const double key = a[i];
                            int j = i - 1;
                            while (j >= 0 && a[j] > key) { a[j + 1] = a[j]; --j; }
                            a[j + 1] = key;
                        }
                        r = (len & 1) ? a[(len - 1) / 2] : (0.0 + a[len / 2 - 1] + a[len / 2]) / 2.0;
                    }
                }
            }
            v = blk_bcast0(b, r);
        } break;
        case TSFA_C_PARTIAL_AUTOCORRELATION: {                           // fc.py:440
            const int l = (int)sp.p[0];
            v = (l >= 0 && l <= max_pacf_lag) ? pac[l] : TSFA_NAN;
        } break;
        case TSFA_C_AUGMENTED_DICKEY_FULLER: {                           // fc.py:499
            const int attr = (int)sp.p[0];
            v = (attr == TSFA_ADF_TESTSTAT) ? adf_stat : (attr == TSFA_ADF_PVALUE ? adf_p : (attr == TSFA_ADF_USEDLAG ? adf_lag : TSFA_NAN));
        } break;
        case TSFA_C_AR_COEFFICIENT: {                                    // fc.py:1459
            const int coeff = (int)sp.p[0], k = (int)sp.p[1];
            if (coeff > k) { v = TSFA_NAN; break; }
            // AutoReg(x, lags=k, trend="c") can only be estimated with n >= 2k + 2; on failure the reference
            // substitutes [nan]*k, so coeff < k -> NaN and coeff == k -> IndexError -> 0
            if (n < 2 * k + 2 || k < 1) {
                v = (coeff < k) ? TSFA_NAN : 0.0;
                break;
            }
            if (k > TSFA_AR_TABLE_K || k + 2 > P) {   // beyond the float64 pass's matrices: every such fit in the second pass (fam_ar_dd.h)
                degenerate |= 1;
                v = TSFA_NAN;
                break;
            }
            if (ar_cached_k != k) {
                // regressors: const, xc[t-1..t-k]; target xc[t]; rows t in [k, n)
                auto sx = [=](int u) { return xcc[u]; };
                auto y_blocks = [=](int i0, double (&B)[8]) { blk.y8(i0, B); };
                double sy[2];
                blk_tile_lagdots(b, n, k + 1,
                                 [=](int tb, double (&A)[8]) { blk.y8(tb, A); tile_mask_rows(A, tb, k, n); }, y_blocks,
                                 [=](int j, double v) { T[j] = v; }, sy, G, P * P);
                if (b.tid == 0) C[0] = sy[0];
                blk_lag_products_rest(b, sx, k, k, n, T, P, C);
                const int p = k + 1;
                for (int e = b.tid; e < p * p; e += b.nt) {
                    const int a = e % p, c = e / p;
                    if (c > a) continue;
                    G[a + c * P] = (a == 0) ? (double)(n - k) : (c == 0 ? C[a] : T[a + c * P]);
                }
                for (int a = b.tid; a < p; a += b.nt) g[a] = (a == 0) ? C[0] : T[a];
                blk_sync();
                double dmin_ar = 0.0;
                ar_ok = blk_chol_factor(b, G, p, P, diag0, &dmin_ar);
                if (!ar_ok) degenerate |= 1;
                else {
                    const double mu = xcc.mean, rows = (double)(n - k);
                    double tr = rows;
                    for (int a = 1; a < p; ++a) tr += diag0[a] + 2.0 * mu * C[a] + rows * mu * mu;
                    if (ar_raw_design_suspect(dmin_ar, sqrt((double)k) * fabs(mu), tr)) degenerate |= 1;
                }
                if (ar_ok && k > TSFA_AR_COND_CHECK_K) {
                    // The pivots of a Cholesky in the natural column order do not reveal a design that is singular to working
                    // precision WITHOUT an exact dependency among leading columns (k + 1 noisy samples, then a stuck sensor:
                    // smallest singular value 1e-21 s_max, every pivot above 1e-3 of its diagonal): the float64 factor then
                    // "solves" a perturbed problem, the refinement step agrees with it, and the coefficients are wrong in the
                    // first digit (a random-parameter fuzz find: ~10 % of such series from AR(16) on; none seen at AR(10) /
                    // AR(12), which is why the orders of ComprehensiveFCParameters do not pay for this).  Two steps of inverse
                    // iteration through the factor estimate the smallest eigenvalue; below 1e-12 of the trace (singular value
                    // ratio 1e-6) the series goes to the double-double pass, which orders the columns by pivoting.
                    double tr0 = 0.0, s0 = 0.0;
                    for (int a = 0; a < p; ++a) { tr0 += diag0[a]; s0 += 1.0 / diag0[a]; }   // uniform
                    for (int a = b.tid; a < p; a += b.nt) tmp1[a] = 1.0 / sqrt(diag0[a] * s0);
                    blk_sync();
                    double lmin = tr0;
                    for (int it = 0; it < 2; ++it) {
                        blk_chol_solve(b, G, p, P, tmp1, tmp2);
                        blk_sync();
                        double nrm2 = 0.0;
                        for (int a = 0; a < p; ++a) nrm2 += tmp2[a] * tmp2[a];   // uniform: every thread reads the same LDS values
                        const double nrm = sqrt(nrm2);
                        if (!(nrm > 0.0) || isinf(nrm) || nrm != nrm) { lmin = 0.0; break; }
                        lmin = 1.0 / nrm;
                        blk_sync();
                        for (int a = b.tid; a < p; a += b.nt) tmp1[a] = tmp2[a] / nrm;
                        blk_sync();
                    }
                    if (!(lmin >= 1e-12 * tr0)) degenerate |= 1;
                }
                if (ar_ok) blk_chol_solve(b, G, p, P, g, beta);
                blk_sync();
                if (ar_ok) {
                    // X^T r of the refinement step: r(t) = y(t) - beta_0 - sum_c beta_c y(t - c) on the tile, then the same
                    // lagged sums with A = r  (lag 0 is the constant's entry: sum_t r(t))
                    double sr[2];
                    blk_tile_lagdots(b, n, k + 1,
                                     [=](int tb, double (&A)[8]) {
                                         blk.y8(tb, A);
                                         const double b0 = beta[0];
#pragma unroll
                                         for (int u = 0; u < 8; ++u) A[u] -= b0;
                                         double lo[8], hi[8], cf[8];
                                         blk.y8(tb, hi);
                                         for (int c0 = 0; c0 <= k; c0 += 8) {
                                             blk.y8(tb - c0 - 8, lo);
#pragma unroll
                                             for (int j = 0; j < 8; ++j) {
                                                 const int c = c0 + j;
                                                 cf[j] = (c >= 1 && c <= k) ? beta[(c <= k) ? c : 0] : 0.0;
                                             }
                                             tile_axpy8(cf, lo, hi, A);
#pragma unroll
                                             for (int q = 0; q < 8; ++q) hi[q] = lo[q];
                                         }
                                         tile_mask_rows(A, tb, k, n);
                                     },
                                     y_blocks, [=](int a, double v) { if (a > 0) tmp1[a] = v; }, sr, T, P * P);
                    if (b.tid == 0) tmp1[0] = sr[0];
                    blk_sync();
                    blk_chol_solve(b, G, p, P, tmp1, tmp2);
                    blk_sync();
                    if (ar_refinement_suspect(b, beta, tmp2, diag0, p)) degenerate |= 1;
                    blk_sync();
                    if (b.tid == 0) {
                        double sphi = 0.0;
                        for (int a = 0; a < p; ++a) {
                            beta[a] += tmp2[a];
                            if (a > 0) sphi += beta[a];
                        }
                        arres[0] = beta[0] + mean * (1.0 - sphi);  // undo the centring of the constant
                        for (int a = 1; a < p; ++a) arres[a] = beta[a];
                    }
                    blk_sync();
                }
                ar_cached_k = k;
            }
            v = ar_ok ? arres[coeff] : TSFA_NAN;
        } break;
        default: break;
        }
        if (b.tid == 0) out_row[sp.col] = v;
        TSFA_TICK(tk, b, sp.calc);
    }
    if (nloop < nspecs) {
        // lane = column: reads of pac[], the ADF outputs and the cached AR fit (all resident in LDS / uniform)
        blk_sync();
        for (int s = nloop + b.tid; s < nspecs; s += b.nt) {
            const TsfaSpec sp = specs[s];
            double v = TSFA_NAN;
            if (sp.calc == TSFA_C_PARTIAL_AUTOCORRELATION) {
                const int l = (int)sp.p[0];
                v = (l >= 0 && l <= max_pacf_lag) ? pac[l] : TSFA_NAN;
            } else if (sp.calc == TSFA_C_AUGMENTED_DICKEY_FULLER) {
                const int attr = (int)sp.p[0];
                v = (attr == TSFA_ADF_TESTSTAT) ? adf_stat : (attr == TSFA_ADF_PVALUE ? adf_p : (attr == TSFA_ADF_USEDLAG ? adf_lag : TSFA_NAN));
            } else if (sp.calc == TSFA_C_AR_COEFFICIENT) {
                const int coeff = (int)sp.p[0], k = (int)sp.p[1];
                if (coeff > k) v = TSFA_NAN;
                else if (n < 2 * k + 2 || k < 1) v = (coeff < k) ? TSFA_NAN : 0.0;
                else v = (ar_cached_k == k && ar_ok && k <= TSFA_AR_TABLE_K) ? arres[coeff] : TSFA_NAN;
            }
            out_row[sp.col] = v;
        }
    }
    return degenerate;
}

#endif
