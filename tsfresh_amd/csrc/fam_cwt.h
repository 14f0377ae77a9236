// Family CWT (per-series part): number_cwt_peaks (fc.py:1320) =
//   len(scipy.signal.find_peaks_cwt(x, widths=1..n, wavelet=_ricker))
// restated from scipy/signal/_peak_finding.py (find_peaks_cwt, _identify_ridge_lines, _filter_ridge_lines,
// _boolrelextrema), scipy/signal/_wavelets.py (_cwt) and fc.py:1307 (_ricker).
//   phase A : CWT rows by direct convolution with the Ricker taps (one column per lane), relative-maximum bit mask
//   phase B : ridge-line linking, one row at a time, every step parallel over columns / lines:
//               snapshot of the live lines' last columns (column -> line map), each maximum looks up its nearest
//               line in the snapshot, new lines get indices in ascending column order (ballot prefix), then every
//               line collects the maxima that chose it.  The reference's algorithm is sequential and O(lines^2).
//   phase C : length / signal-to-noise filter, lane = line (10th percentile of the width-1 row in a window)
// The cwt_coefficients contraction is a separate MFMA kernel (tsfa_kernels.hip: k_cwt_gemm).
#ifndef TSFA_FAM_CWT_H
#define TSFA_FAM_CWT_H

#include "tsfa_common.h"

#define TSFA_CWTP_MAXW 16
#define TSFA_CWTP_MAXTAPS (10 * TSFA_CWTP_MAXW)

// fc.py:1307 _ricker(points, a)[k]
TSFA_DEV double ricker_tap(int points, double a, int k) {
    const double A = 2.0 / (sqrt(3.0 * a) * pow(M_PI, 0.25));
    const double wsq = a * a;
    const double vec = (double)k - ((double)points - 1.0) / 2.0;
    const double xsq = vec * vec;
    const double mod = 1.0 - xsq / wsq;
    const double gauss = exp(-xsq / (2.0 * wsq));
    return A * mod * gauss;
}

// the same tap from the plan's table (rk: Ricker part of tsfa_build_consts, or null) when the width has its full 10 w
// points -- every width of a series longer than 10 W; shorter series truncate the wavelet and evaluate it
TSFA_DEV double ricker_tap_t(const double *rk, int points, int w, int k) {
    if (rk != nullptr && points == 10 * w && w <= 16) return rk[5 * w * (w - 1) + k];
    return ricker_tap(points, (double)w, k);
}

// scipy.signal.convolve(x, h, mode="same")[c] for len(h) = nw <= n: full[c + (nw-1)/2], full[m] = sum_k h[k] x[m-k]
template <class X>
TSFA_DEV double conv_same_at(X xv, int n, const double *h, int nw, int c) {
    const int m = c + (nw - 1) / 2;
    int k0 = m - (n - 1);
    if (k0 < 0) k0 = 0;
    const int k1 = (m < nw - 1) ? m : (nw - 1);
    double acc = 0.0;
    for (int k = k1; k >= k0; --k) acc += xv(m - k) * h[k];
    return acc;
}

#if TSFA_GPU
#define TSFA_ENT_WAVE_C 64
#else
#define TSFA_ENT_WAVE_C 1
#endif
#define TSFA_CWTP_HALO ((TSFA_CWTP_MAXTAPS + 1) / 2 + 2)   // zero padding of the staged series on either side

// column / ridge-line indices: 16 bits where a series fits a CU's LDS, 32 bits in the long-series build (series beyond
// 65 535 samples; the mask stays 16 bits per column: one bit per width + the two marks)
#if defined(TSFA_LONG)
typedef unsigned int cwt_idx_t;
#define TSFA_CWT_IDX_NONE 0xFFFFFFFFu
#else
typedef unsigned short cwt_idx_t;
#define TSFA_CWT_IDX_NONE 0xFFFFu
#endif

struct CwtPeaksLds {
    double *red; double *row0; double *rowv; double *taps; void *xpad; unsigned short *mask; cwt_idx_t *lcol;
    cwt_idx_t *linf; cwt_idx_t *colmap; cwt_idx_t *mline; int *misc;
    const double *rk = nullptr;   // the plan's Ricker tap table (tsfa_build_consts + TSFA_CONSTS_RICKER; global memory), or null
};

// linf packing: length (6 bits, saturating), gap (2 bits), retired flag, last row (4 bits)
#define TSFA_LI_LEN(v) ((v) & 63)
#define TSFA_LI_GAP(v) (((v) >> 6) & 3)
#define TSFA_LI_DEAD(v) (((v) >> 8) & 1)
#define TSFA_LI_ROW(v) (((v) >> 9) & 15)
#define TSFA_LI_PACK(len, gap, dead, row) ((cwt_idx_t)(((len) & 63) | (((gap) & 3) << 6) | (((dead) & 1) << 8) | (((row) & 15) << 9)))

// ascending sort of eight values: the 19-comparator network
TSFA_DEV void sort8_f64(double (&v)[8]) {
    ce_f64(v[0], v[1]); ce_f64(v[2], v[3]); ce_f64(v[4], v[5]); ce_f64(v[6], v[7]);
    ce_f64(v[0], v[2]); ce_f64(v[1], v[3]); ce_f64(v[4], v[6]); ce_f64(v[5], v[7]);
    ce_f64(v[1], v[2]); ce_f64(v[5], v[6]); ce_f64(v[0], v[4]); ce_f64(v[3], v[7]);
    ce_f64(v[1], v[5]); ce_f64(v[2], v[6]);
    ce_f64(v[1], v[4]); ce_f64(v[3], v[6]);
    ce_f64(v[2], v[4]); ce_f64(v[3], v[5]);
    ce_f64(v[3], v[4]);
}
// r <- the eight smallest of r and g, ascending (both ascending on entry): min(r[i], g[7 - i]) is a bitonic sequence
// holding exactly those eight; three half-cleaner stages sort it
TSFA_DEV void merge_low8_f64(double (&r)[8], const double (&g)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = min_f64(r[i], g[7 - i]);
    ce_f64(r[0], r[4]); ce_f64(r[1], r[5]); ce_f64(r[2], r[6]); ce_f64(r[3], r[7]);
    ce_f64(r[0], r[2]); ce_f64(r[1], r[3]); ce_f64(r[4], r[6]); ce_f64(r[5], r[7]);
    ce_f64(r[0], r[1]); ce_f64(r[2], r[3]); ce_f64(r[4], r[5]); ce_f64(r[6], r[7]);
}

// value of rank i0 and i0 + 1 (0-based, ascending) among r0[0 .. m-1], i0 + 1 < 8 (nmax: readable elements of r0).
// The window is taken in groups of eight (the last one padded with +inf), each sorted by the network and merged into
// the running eight smallest: 70 minimum / maximum instructions per group, against ~25 per ELEMENT for an insertion
// list -- this selection (one per ridge-line end point) is the largest part of the kernel after the convolutions.
TSFA_DEV void lowest8_select(const double *r0, int m, int nmax, int i0, double *s0, double *s1) {
    double r[8], g[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const double e = r0[(q < nmax) ? q : (nmax - 1)];
        r[q] = (q < m) ? e : TSFA_INF;
    }
    sort8_f64(r);
    for (int a = 8; a < m; a += 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const double e = r0[(a + q < nmax) ? a + q : (nmax - 1)];
            g[q] = (a + q < m) ? e : TSFA_INF;
        }
        sort8_f64(g);
        merge_low8_f64(r, g);
    }
    double a0 = r[0], a1 = r[1];
#pragma unroll
    for (int k = 1; k < 7; ++k)
        if (k == i0) { a0 = r[k]; a1 = r[k + 1]; }
    *s0 = a0;
    *s1 = a1;
}

// Four consecutive outputs of one width's convolution in registers: out[c0 + j] = sum_t hh[t] x[u0 + j + t], t ascending
// (the accumulation order of conv_same_at: ascending sample).  The seven-sample window slides by four taps per trip, so a
// trip is 16 fused multiply-adds, four conversions and three register moves -- the kernel is VALU-issue bound and the
// multiply-adds are the only instructions that have to be there.
template <class XA>
TSFA_DEV void cwt_tile4(XA xat, const double *hh, int nw, int u0, double &r0, double &r1, double &r2, double &r3) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    double x0 = xat(u0), x1 = xat(u0 + 1), x2 = xat(u0 + 2);
    int t = 0;
    for (; t + 4 <= nw; t += 4) {
        const double h0 = hh[t], h1 = hh[t + 1], h2 = hh[t + 2], h3 = hh[t + 3];
        const double x3 = xat(u0 + t + 3), x4 = xat(u0 + t + 4), x5 = xat(u0 + t + 5), x6 = xat(u0 + t + 6);
        // explicit fused multiply-adds: the fused form is the more accurate one, and the reference's own convolution
        // (np.convolve) leaves the contraction to the CPU's BLAS-style inner loop anyway
        a0 = fma(x0, h0, a0); a1 = fma(x1, h0, a1); a2 = fma(x2, h0, a2); a3 = fma(x3, h0, a3);
        a0 = fma(x1, h1, a0); a1 = fma(x2, h1, a1); a2 = fma(x3, h1, a2); a3 = fma(x4, h1, a3);
        a0 = fma(x2, h2, a0); a1 = fma(x3, h2, a1); a2 = fma(x4, h2, a2); a3 = fma(x5, h2, a3);
        a0 = fma(x3, h3, a0); a1 = fma(x4, h3, a1); a2 = fma(x5, h3, a2); a3 = fma(x6, h3, a3);
        x0 = x4; x1 = x5; x2 = x6;
    }
    for (; t < nw; ++t) {
        const double h = hh[t];
        const double x3 = xat(u0 + t + 3);
        a0 = fma(x0, h, a0); a1 = fma(x1, h, a1); a2 = fma(x2, h, a2); a3 = fma(x3, h, a3);
        x0 = x1; x1 = x2; x2 = x3;
    }
    r0 = a0; r1 = a1; r2 = a2; r3 = a3;
}

// CWT rows for widths 1..W by direct convolution with the Ricker taps, relative-maximum bit mask, row 0 kept.
// xat(i) = sample i, 0 outside [0, n): every output runs the full tap range without bounds (the padded products are
// exact zeros).  A thread owns four consecutive columns (cwt_tile4); the relative-maximum test of its first and last
// column needs the neighbouring threads' outer values, which travel through `edge` (two values per four columns,
// L.lcol / L.linf are idle until phase B): the bit is set on the thread's own evidence and withdrawn after the barrier if
// the neighbour is not smaller.  No output is computed twice and no CWT row is stored.
template <class XA>
TSFA_DEV void cwt_rows_tiled(const Blk &b, XA xat, int n, int W, const CwtPeaksLds &L) {
    double *edge = (double *)L.lcol;   // edge[2 g] / edge[2 g + 1]: first / last output of columns 4 g .. 4 g + 3
    for (int w = 1; w <= W; ++w) {
        const int nw = (10 * w < n) ? 10 * w : n;
        for (int k = b.tid; k < nw; k += b.nt) L.taps[k] = ricker_tap_t(L.rk, nw, w, k);  // tap of sample offset k
        blk_sync();
        const int lead = (nw - 1) - (nw - 1) / 2;  // out[c] = sum_t taps[t] x[c - lead + t]
        const unsigned short bit = (unsigned short)(1u << (w - 1));
        for (int c0 = 4 * b.tid; c0 < n; c0 += 4 * b.nt) {
            double a0, a1, a2, a3;
            cwt_tile4(xat, L.taps, nw, c0 - lead, a0, a1, a2, a3);
            edge[c0 >> 1] = a0;
            edge[(c0 >> 1) + 1] = a3;
            // _boolrelextrema(order=1, mode="clip"): strict, never at the ends
            if (c0 >= 1 && c0 < n - 1 && a0 > a1) L.mask[c0] |= bit;                 // pending: the left neighbour
            if (c0 + 1 < n - 1 && a1 > a0 && a1 > a2) L.mask[c0 + 1] |= bit;
            if (c0 + 2 < n - 1 && a2 > a1 && a2 > a3) L.mask[c0 + 2] |= bit;
            if (c0 + 3 < n - 1 && a3 > a2) L.mask[c0 + 3] |= bit;                    // pending: the right neighbour
            if (w == 1) {
                L.row0[c0] = a0;
                if (c0 + 1 < n) L.row0[c0 + 1] = a1;
                if (c0 + 2 < n) L.row0[c0 + 2] = a2;
                if (c0 + 3 < n) L.row0[c0 + 3] = a3;
            }
        }
        blk_sync();
        for (int c0 = 4 * b.tid; c0 < n; c0 += 4 * b.nt) {
            const int g2 = c0 >> 1;
            if (c0 >= 1 && c0 < n - 1 && !(edge[g2] > edge[g2 - 1])) L.mask[c0] &= (unsigned short)~bit;
            if (c0 + 3 < n - 1 && !(edge[g2 + 1] > edge[g2 + 2])) L.mask[c0 + 3] &= (unsigned short)~bit;
        }
        blk_sync();  // the next width's taps and edges
    }
}

#if TSFA_GPU && !defined(TSFA_LONG)
// ---- phase A on the matrix cores (v_mfma_f64_16x16x4_f64): the contraction BASELINE.json's north_star names ----
// The Ricker convolution of one width, out[c] = sum_t taps_w[t] x[c - 5 w + t] (t < 10 w), is a GEMM once the output
// OFFSET inside a block of 16 goes on N:  c = 16 i + j,
//     out[16 i + j] = sum_k' A[i][k'] B_w[k'][j],   A[i][k'] = x[16 i + k' - 5 w]   (overlapping rows of the padded LDS copy)
//                                                   B_w[k'][j] = taps_w[k' - j]     (banded Toeplitz, zero outside the taps)
// k' < 10 w + 15, in steps of 4: (10 w + 18) / 4 MFMAs per tile of 16 x 16 = 256 consecutive outputs -- 59 per tile for widths
// 1..5, 64 % of the multiply-adds useful (padding the five widths to N = 16 would waste 3 x).  A wavefront owns whole
// (tile, width) items; the accumulator layout D[i = 4 v + (lane >> 4)][j = lane & 15] puts output 256 tile + 64 v + lane into
// register v of a lane, so the strict-maximum test reads its neighbours with wave-shift DPP moves and only the first / last
// output of a tile travel through LDS (`edge`, as in the register-tiled form).  The float64 matrix pipe has the VALU's own
// multiply-add rate on gfx950 (64 cycles per 16x16x4 = 16 MAC / cycle / SIMD: SQ_VALU_MFMA_BUSY_CYCLES of k_cwt_gemm); what
// it buys is ISSUE: one instruction per 1024 multiply-adds in a kernel that is VALU-issue bound, on a pipe that runs beside
// the other workgroups' ridge-line phases.  The summation order differs from np.convolve's (as the FMA tiles' did): a pair
// of neighbours equal to 1e-12 is R8's business (tests/parity.py).
typedef double cwt_d4 __attribute__((ext_vector_type(4)));
#define TSFA_CWTM_TBL(w) (10 * (w) + 33)   /* band table of a width: [15 zeros][10 w taps][18 zeros] */

// can phase A of a series of n samples, widths 1..W, run here?  (full 10 w taps for every width, the widest band table in
// the colmap | mline block, two or more K steps)
TSFA_DEV bool cwt_mfma_fits(int n, int W) { return W <= TSFA_CWTP_MAXW && 10 * W < n && 2 * TSFA_CWTM_TBL(W) <= n && 16 * ((n + 255) >> 8) <= (n >> 1); }

template <int CTRL>
TSFA_DEV double dpp_mov_old_f64(double old, double v) {   // lanes the DPP pattern gives no source keep `old`
    union { double d; int i[2]; } a, o, r;
    a.d = v;
    o.d = old;
    r.i[0] = __builtin_amdgcn_update_dpp(o.i[0], a.i[0], CTRL, 0xf, 0xf, false);
    r.i[1] = __builtin_amdgcn_update_dpp(o.i[1], a.i[1], CTRL, 0xf, 0xf, false);
    return r.d;
}

// One (tile, width) item: ksteps MFMAs over two accumulator chains (even / odd K steps: a dependent MFMA would wait out the
// pipe's latency).  KS > 0: the step count as a constant -- the loop unrolls, the LDS operands sit at immediate offsets and
// the scheduler hoists the reads over the matrix instructions; KS == 0: any width, operands of the NEXT pair fetched before
// the current pair issues.
template <class ST, int KS>
TSFA_DEV void cwt_mfma_item(const ST *xa, const double *tb, int ksteps, cwt_d4 &acc0, cwt_d4 &acc1) {
    acc0 = (cwt_d4){0.0, 0.0, 0.0, 0.0};
    acc1 = (cwt_d4){0.0, 0.0, 0.0, 0.0};
    if (KS > 0) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const double a = (double)xa[4 * ks], bv = tb[4 * ks];
            if (ks & 1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc0, 0, 0, 0);
        }
    } else {
        double a0 = (double)xa[0], a1 = (double)xa[4], b0 = tb[0], b1 = tb[4];
        int ks = 0;
        for (; ks + 4 <= ksteps; ks += 2) {
            const double na0 = (double)xa[4 * ks + 8], na1 = (double)xa[4 * ks + 12], nb0 = tb[4 * ks + 8], nb1 = tb[4 * ks + 12];
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc1, 0, 0, 0);
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        }
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc1, 0, 0, 0);
        ks += 2;
        if (ks < ksteps) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)xa[4 * ks], tb[4 * ks], acc0, 0, 0, 0);
    }
}

template <class ST>
TSFA_DEV void cwt_rows_mfma(const Blk &b, const ST *xpad, int n, int W, const CwtPeaksLds &L) {
    const int lane = b.tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(b.tid >> 6), nwaves = (b.nt + 63) >> 6;   // scalar: the item loop is SALU
    const int r = lane & 15, kq = lane >> 4;
    const int ntiles = (n + 255) >> 8, nchunks = 4 * ntiles;   // a chunk = 64 consecutive outputs = one accumulator register
    double *tbl = (double *)L.colmap;          // colmap | mline: 4 n bytes, re-zeroed below
    const int tbl_cap = n >> 1;
    double *edge = (double *)L.lcol;           // lcol | linf: edge[2 e] / edge[2 e + 1] = first / last output of chunk e
    unsigned *maskw = (unsigned *)L.mask;
    const int xlast = n + 2 * TSFA_CWTP_HALO + 7;   // last initialised (zero) entry of the padded copy
    const double ninf = -TSFA_INF;
    int w0 = 1;
    while (w0 <= W) {
        int w1 = w0, used = 0;
        while (w1 <= W && used + TSFA_CWTM_TBL(w1) <= tbl_cap && 2 * (w1 - w0 + 1) * nchunks <= tbl_cap) { used += TSFA_CWTM_TBL(w1); ++w1; }
        const int gw = w1 - w0;   // >= 1 by cwt_mfma_fits
        {   // the band tables of the group, one pass over their concatenation
            int w = w0, base = 0;
            for (int q0 = 0; q0 < used; q0 += b.nt) {
                const int q = q0 + b.tid;
                while (w < w1 - 1 && q0 >= base + TSFA_CWTM_TBL(w)) { base += TSFA_CWTM_TBL(w); ++w; }   // uniform lower bound
                int wq = w, bq = base;
                while (wq < w1 - 1 && q >= bq + TSFA_CWTM_TBL(wq)) { bq += TSFA_CWTM_TBL(wq); ++wq; }
                const int t = q - bq - 15;
                if (q < used) tbl[q] = (t >= 0 && t < 10 * wq) ? L.rk[5 * wq * (wq - 1) + t] : 0.0;
            }
        }
        blk_sync();
        int k = 0, pass = 0;   // item p = (wi, tile) -> wave (pass odd ? nwaves - 1 - k : k), k = p mod nwaves: snake order
        for (int wi = 0; wi < gw; ++wi) {
            const int w = w1 - 1 - wi;   // the widest widths go out first
            const int toff = 5 * (w * (w - 1) - w0 * (w0 - 1)) + 33 * (w - w0);
            const int ksteps = (10 * w + 18) >> 2;
            const unsigned bit = 1u << (w - 1);
            for (int tile = 0; tile < ntiles; ++tile) {
                const int owner = pass ? nwaves - 1 - k : k;
                if (++k == nwaves) { k = 0; pass ^= 1; }
                if (owner != wave) continue;
                const double *tb = tbl + toff + (kq - r + 15);
                // rows of a tile hanging over the end of the series would read past the padded copy: every output of such a
                // row lies beyond n (16 i > n + 72), so the row reads the head of the buffer instead -- finite, never used
                int xi0 = TSFA_CWTP_HALO + 16 * (16 * tile + r) + kq - 5 * w;
                xi0 = (xi0 + 4 * (ksteps - 1) <= xlast) ? xi0 : 0;
                const ST *xa = xpad + xi0;
                cwt_d4 acc0, acc1;
                switch (ksteps) {   // widths 1 .. 5 (the settings objects): K loop unrolled, operands at immediate offsets
                case 7: cwt_mfma_item<ST, 7>(xa, tb, 7, acc0, acc1); break;
                case 9: cwt_mfma_item<ST, 9>(xa, tb, 9, acc0, acc1); break;
                case 12: cwt_mfma_item<ST, 12>(xa, tb, 12, acc0, acc1); break;
                case 14: cwt_mfma_item<ST, 14>(xa, tb, 14, acc0, acc1); break;
                case 17: cwt_mfma_item<ST, 17>(xa, tb, 17, acc0, acc1); break;
                default: cwt_mfma_item<ST, 0>(xa, tb, ksteps, acc0, acc1); break;
                }
                const int e0 = 2 * (wi * nchunks + 4 * tile);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const double o = acc0[v] + acc1[v];
                    const int c = 256 * tile + 64 * v + lane;
                    // the neighbours inside the chunk by wave shifts; lane 0 / lane 63 see -inf (the bit is set on the lane's
                    // own evidence and withdrawn below against the neighbouring chunk's edge value)
                    const double left = dpp_mov_old_f64<0x138 /* wave_shr:1 */>(ninf, o);
                    const double right = dpp_mov_old_f64<0x130 /* wave_shl:1 */>(ninf, o);
                    if (lane == 0) edge[e0 + 2 * v] = o;
                    if (lane == 63) edge[e0 + 2 * v + 1] = o;
                    // _boolrelextrema(order=1, mode="clip"): strict, never at the ends
                    if (c >= 1 && c < n - 1 && o > left && o > right) atomicOr(&maskw[c >> 1], bit << ((c & 1) * 16));
                    if (w == 1 && c < n) L.row0[c] = o;
                }
            }
        }
        blk_sync();
        // the chunk boundaries: column 64 q - 1 (last of chunk q - 1) against column 64 q (first of chunk q)
        for (int q = b.tid; q < gw * nchunks; q += b.nt) {
            const int wi = q / nchunks, ch = q - wi * nchunks;
            if (ch == 0) continue;
            const unsigned bit = 1u << (w1 - 2 - wi);   // width w1 - 1 - wi
            const double lastp = edge[2 * (q - 1) + 1], first = edge[2 * q];
            const int c = 64 * ch;
            if (!(first > lastp)) atomicAnd(&maskw[c >> 1], ~(bit << ((c & 1) * 16)));
            if (!(lastp > first)) atomicAnd(&maskw[(c - 1) >> 1], ~(bit << (((c - 1) & 1) * 16)));
        }
        blk_sync();
        w0 = w1;
    }
    for (int c = b.tid; c < n; c += b.nt) { L.colmap[c] = 0; L.mline[c] = 0; }
    blk_sync();
}
#endif

// cwt[row, col], re-evaluated for a ridge line that ended above row 0 (row 0 itself is resident: L.row0)
template <class X>
TSFA_DEV double cwt_signal_at(X xv, int n, const CwtPeaksLds &L, int col, int row, bool taps_cached) {
    if (row == 0) return L.row0[col];
    const int w = row + 1;
    const int nw = (10 * w < n) ? 10 * w : n;
    const int m2 = col + (nw - 1) / 2;
    int kk0 = m2 - (n - 1);
    if (kk0 < 0) kk0 = 0;
    const int kk1 = (m2 < nw - 1) ? m2 : (nw - 1);
    double acc = 0.0;
    if (taps_cached) {
        const double *tw = L.taps + 5 * (w - 2) * (w + 1);  // sum_{v=2}^{w-1} 10 v
        for (int kk = kk1; kk >= kk0; --kk) acc += xv(m2 - kk) * tw[kk];
    } else {
        for (int kk = kk1; kk >= kk0; --kk) acc += xv(m2 - kk) * ricker_tap_t(L.rk, nw, w, nw - 1 - kk);
    }
    return acc;
}

// scipy.stats.scoreatpercentile(window, 10) from the two order statistics it reads (ranks i0, i0 + 1; idx = 0.1 (m - 1))
TSFA_DEV double cwt_noise_of(double s0, double s1, double idx, int i0) {
    if ((double)i0 == idx) return s0;
    const double j = (double)(i0 + 1);
    const double w0 = j - idx, w1 = idx - (double)i0;
    return (s0 * w0 + s1 * w1) / (w0 + w1);
}

// Length / signal-to-noise filter of scipy.signal._peak_finding._filter_ridge_lines over a list of ridge-line end points:
// entry k = (cols[k], rows ? rows[k] : 0).  signal = cwt[row, col], noise = the 10th percentile
// (scipy.stats.scoreatpercentile) of row 0 in the window [col - hf, col + hf + odd) clipped to the series.
// Returns the number of entries with |signal / noise| >= 1; *extra receives how many of those carry `extra_bit` in
// mask[col] (0: none asked).  SHORT windows (the percentile is among the eight smallest, or a rank count over <= 70 values),
// lane = entry; long windows: cwt_filter_counts below.
template <class X>
TSFA_DEV double cwt_filter_list(const Blk &b, X xv, int n, const CwtPeaksLds &L, const cwt_idx_t *cols,
                                const cwt_idx_t *rows, int cnt, int hf, int odd, bool taps_cached,
                                unsigned short extra_bit, double *extra) {
    double kept = 0.0, ext = 0.0;
    for (int k0 = 0; k0 < cnt; k0 += b.nt) {  // uniform over the workgroup
        const int k = k0 + b.tid;
        const bool live = k < cnt;
        const int col = live ? (int)cols[k] : 0, row = (live && rows) ? (int)rows[k] : 0;
        const int ws = (col - hf > 0) ? col - hf : 0;
        const int we = (col + hf + odd < n) ? col + hf + odd : n;
        const int m = we - ws;
        const double idx = 10.0 / 100.0 * (double)(m - 1);
        const int i0 = (int)idx;
        double s0 = 0.0, s1 = 0.0;
        if (live) {
            const double *r0 = L.row0 + ws;
            if (i0 + 1 < 8) {
                lowest8_select(r0, m, n - ws, i0, &s0, &s1);
            } else {
                for (int a = 0; a < m; ++a) {
                    const double ea = r0[a];
                    int rank = 0;
                    for (int c = 0; c < m; ++c) {
                        const double ec = r0[c];
                        rank += (ec < ea || (ec == ea && c < a)) ? 1 : 0;
                    }
                    if (rank == i0) s0 = ea;
                    if (rank == i0 + 1) s1 = ea;
                }
            }
        }
        if (!live) continue;
        const double noise = cwt_noise_of(s0, s1, idx, i0);
        const double sig = cwt_signal_at(xv, n, L, col, row, taps_cached);
        const double snr = fabs(sig / noise);
        if (!(snr < 1.0)) {
            kept += 1.0;
            if (extra_bit && (L.mask[col] & extra_bit)) ext += 1.0;
        }
    }
    kept = blk_sum(b, kept);
    if (extra_bit) *extra = blk_sum(b, ext);
    return kept;
}

// The same filter for LONG windows (series beyond ~1400 samples: hundreds of columns per window, the percentile is the
// 8th .. 40th smallest).  Rounds 2-5 argsorted row 0 once and let every entry walk that global order until it had met
// i0 + 2 columns of its own window (~0.1 n order entries per 64 end points, 12 instructions each): at 8192 samples the sort and
// the walks were three quarters of the kernel.  The filter only needs a DECISION, |signal| against |noise|, and
//     |x / y| >= 1  <=>  |x| >= |y|     (IEEE division: a quotient below 1 cannot round up to 1)
// so an entry is decided by COUNTING its window against thresholds at +-|signal| -- no order at all.  With a = |signal|,
// d = 2^-40 and the window counts
//     A = #{x < -a (1 + d)},  B = #{x < -a (1 - d)},  C = #{x <= a (1 - d)},  D = #{x <= a (1 + d)}
// and r1 = the larger rank the percentile reads (i0 + 1, or i0 when 0.1 (m - 1) is whole):
//     r1 < A              both order statistics lie below -a (1 + d): |noise| > a              -> dropped
//     i0 >= B, r1 < C     both lie in [-a (1 - d), a (1 - d)]:       |noise| < a              -> kept
//     i0 >= D             both lie above a (1 + d)                                              -> dropped
// (the interpolation of two values moves them by a few ulps, far inside the 2^-40 margins).  Eight instructions per window
// column.  What is left -- a threshold falls between the two ranks, or the window holds values equal (or closer than 2^-40) to
// +-a: ~3 % of the entries of a noisy series, every entry of a periodic one -- is compacted into a second list and decided from
// the two order statistics themselves: in those cases both ranks lie next to a threshold, where one more sweep gives them
// exactly -- rank A - 1 = max{x < -a (1 + d)}, ranks [A, B) = -a if that band holds one value, rank B = min{x >= -a (1 - d)},
// and the same around +a -- then scipy's interpolation and division as written.  A band of DIFFERENT values within 2^-40 of
// a threshold falls back to ranking the window (m^2, never seen outside constructed inputs).  amb_col / amb_row: scratch for
// the second list, cnt entries each.
template <class X>
TSFA_DEV double cwt_filter_counts(const Blk &b, X xv, int n, const CwtPeaksLds &L, const cwt_idx_t *cols,
                                  const cwt_idx_t *rows, int cnt, int hf, int odd, bool taps_cached,
                                  unsigned short extra_bit, double *extra, cwt_idx_t *amb_col, cwt_idx_t *amb_row) {
    double kept = 0.0, ext = 0.0;
    const double dl = 1.0 + 0x1p-40, ds = 1.0 - 0x1p-40;
    int namb = 0;
    for (int k0 = 0; k0 < cnt; k0 += b.nt) {  // uniform over the workgroup
        const int k = k0 + b.tid;
        const bool live = k < cnt;
        const int col = live ? (int)cols[k] : 0, row = (live && rows) ? (int)rows[k] : 0;
        const int ws = (col - hf > 0) ? col - hf : 0;
        const int we = (col + hf + odd < n) ? col + hf + odd : n;
        const int m = we - ws;
        const double idx = 10.0 / 100.0 * (double)(m - 1);
        const int i0 = (int)idx;
        const int r1 = ((double)i0 == idx) ? i0 : i0 + 1;
        bool amb = false;
        if (live) {
            const double sig = cwt_signal_at(xv, n, L, col, row, taps_cached);
            const double a = fabs(sig);
            if (!(a >= 1e-290 && a <= 1e300)) {
                amb = true;   // 0, subnormal, huge, NaN: no margins to speak of -- the order statistics decide
            } else {
                const double tA = -a * dl, tB = -a * ds, tC = a * ds, tD = a * dl;
                int cA = 0, cB = 0, cC = 0, cD = 0;
                const double *r0 = L.row0 + ws;
                for (int e = 0; e < m; ++e) {
                    const double x = r0[e];
                    cA += (x < tA) ? 1 : 0;
                    cB += (x < tB) ? 1 : 0;
                    cC += (x <= tC) ? 1 : 0;
                    cD += (x <= tD) ? 1 : 0;
                }
                bool keep = false;
                if (r1 < cA) keep = false;
                else if (i0 >= cB && r1 < cC) keep = true;
                else if (i0 >= cD) keep = false;
                else amb = true;
                if (keep) {
                    kept += 1.0;
                    if (extra_bit && (L.mask[col] & extra_bit)) ext += 1.0;
                }
            }
        }
        int tot;
        const int at = namb + blk_excl_count(b, amb, &tot);
        if (amb) { amb_col[at] = (cwt_idx_t)col; amb_row[at] = (cwt_idx_t)row; }
        namb += tot;
    }
    blk_sync();
    for (int k0 = 0; k0 < namb; k0 += b.nt) {
        const int k = k0 + b.tid;
        if (k >= namb) continue;
        const int col = (int)amb_col[k], row = (int)amb_row[k];
        const int ws = (col - hf > 0) ? col - hf : 0;
        const int we = (col + hf + odd < n) ? col + hf + odd : n;
        const int m = we - ws;
        const double idx = 10.0 / 100.0 * (double)(m - 1);
        const int i0 = (int)idx;
        const bool two = ((double)i0 != idx);
        const double sig = cwt_signal_at(xv, n, L, col, row, taps_cached);
        const double a = fabs(sig);
        const double *r0 = L.row0 + ws;
        double s0 = 0.0, s1 = 0.0;
        bool known = false;
        if (a >= 1e-290 && a <= 1e300) {
            const double tA = -a * dl, tB = -a * ds, tC = a * ds, tD = a * dl;
            int cA = 0, cB = 0, cC = 0, cD = 0;
            double lb = -TSFA_INF, la = TSFA_INF, ub = -TSFA_INF, ua = TSFA_INF;       // ranks A - 1, B, C - 1, D
            double minL = TSFA_INF, maxL = -TSFA_INF, minU = TSFA_INF, maxU = -TSFA_INF;   // the bands [A, B), [C, D)
            for (int e = 0; e < m; ++e) {
                const double x = r0[e];
                const bool bA = x < tA, bB = x < tB, bC = x <= tC, bD = x <= tD;
                cA += bA ? 1 : 0; cB += bB ? 1 : 0; cC += bC ? 1 : 0; cD += bD ? 1 : 0;
                lb = fmax(lb, bA ? x : -TSFA_INF);
                la = fmin(la, bB ? TSFA_INF : x);
                ub = fmax(ub, bC ? x : -TSFA_INF);
                ua = fmin(ua, bD ? TSFA_INF : x);
                const bool inL = bB && !bA, inU = bD && !bC;
                minL = fmin(minL, inL ? x : TSFA_INF); maxL = fmax(maxL, inL ? x : -TSFA_INF);
                minU = fmin(minU, inU ? x : TSFA_INF); maxU = fmax(maxU, inU ? x : -TSFA_INF);
            }
            // value of rank r, where this sweep determines it
            auto at_rank = [&](int r, double *v) {
                if (r == cA - 1) { *v = lb; return true; }
                if (r >= cA && r < cB) { *v = minL; return minL == maxL; }
                if (r == cB && cB < m) { *v = la; return true; }
                if (r == cC - 1 && cC >= 1) { *v = ub; return true; }
                if (r >= cC && r < cD) { *v = minU; return minU == maxU; }
                if (r == cD && cD < m) { *v = ua; return true; }
                return false;
            };
            known = at_rank(i0, &s0);
            if (known && two) known = at_rank(i0 + 1, &s1);
        }
        if (!known) {   // rank the window
            for (int p = 0; p < m; ++p) {
                const double ea = r0[p];
                int rank = 0;
                for (int c = 0; c < m; ++c) {
                    const double ec = r0[c];
                    rank += (ec < ea || (ec == ea && c < p)) ? 1 : 0;
                }
                if (rank == i0) s0 = ea;
                if (rank == i0 + 1) s1 = ea;
            }
        }
        const double noise = cwt_noise_of(s0, s1, idx, i0);
        const double snr = fabs(sig / noise);
        if (!(snr < 1.0)) {
            kept += 1.0;
            if (extra_bit && (L.mask[col] & extra_bit)) ext += 1.0;
        }
    }
    kept = blk_sum(b, kept);
    if (extra_bit) *extra = blk_sum(b, ext);
    return kept;
}

// ST: element type of the padded LDS copy of the series (the input precision: float32 samples stay float32)
// derive_w1 / kept_w1: also return number_cwt_peaks(n = 1) of the same series (phase C), W <= 14
template <class ST, bool MFMA = false, class X>
TSFA_DEV double number_cwt_peaks_one(const Blk &b, X xv, int n, int W, const CwtPeaksLds &L, bool derive_w1 = false,
                                     double *kept_w1 = nullptr) {
    const int cap = n;  // line capacity
    TSFA_TICKER(tk, 0);
    blk_sync();
    for (int c = b.tid; c < n; c += b.nt) { L.mask[c] = 0; L.colmap[c] = 0; L.mline[c] = 0; }
    blk_sync();
    // ---- phase A: the series goes to LDS between two zero halos (xpad[TSFA_CWTP_HALO + i] = x[i]) when it fits ----
    if (L.xpad != nullptr) {
        blk_sync();
        ST *xpad = (ST *)L.xpad;
        bool finite = true;
        for (int i = b.tid; i < n + 2 * TSFA_CWTP_HALO + 8; i += b.nt) {
            const int j = i - TSFA_CWTP_HALO;
            const ST v = (j >= 0 && j < n) ? (ST)xv(j) : (ST)0;
            xpad[i] = v;
            if (MFMA) finite = finite && (fabs((double)v) < TSFA_INF);
        }
        const ST *xp0 = xpad + TSFA_CWTP_HALO;
        (void)finite;
#if TSFA_GPU && !defined(TSFA_LONG)
        // the opt-in instantiation (TSFA_CWT_MFMA=1; measured slower): a non-finite sample would poison
        // the 15 other outputs of its MFMA row (inf x 0), those series keep the tiles
        if (MFMA && L.rk != nullptr && cwt_mfma_fits(n, W) && !__syncthreads_or(finite ? 0 : 1))
            cwt_rows_mfma<ST>(b, xpad, n, W, L);
        else
#endif
        cwt_rows_tiled(b, [=](int i) { return (double)xp0[i]; }, n, W, L);
    } else {  // no room for the padded copy (very long series): the same tiles, samples straight from HBM / L2
        cwt_rows_tiled(b, [=](int i) { return (i >= 0 && i < n) ? xv(i) : 0.0; }, n, W, L);
    }
    blk_sync();
    TSFA_TICK(tk, b, 151);
    // ---- phase B ----
    // A thread owns CT consecutive columns: the maxima of a row that start new lines get their indices from ONE
    // workgroup scan of per-thread counts (ascending column order = the reference's order of creation).
    const int CT = (n + b.nt - 1) / b.nt;
    int ctbits = 1;
    while ((1 << ctbits) <= CT) ++ctbits;
    const int cbeg = CT * b.tid, cend = (cbeg + CT < n) ? cbeg + CT : n;
    unsigned anyrow = 0;
    for (int c = cbeg; c < cend; ++c) anyrow |= L.mask[c];
    anyrow = blk_or16(b, anyrow) & ((1u << W) - 1u);
    int start_row = -1;
    for (int r = 0; r < W; ++r)
        if ((anyrow >> r) & 1u) start_row = r;
    int nlines = 0, overflow = 0;
    if (start_row >= 0) {
        const unsigned sbit = 1u << start_row;
        {   // initial lines, ascending column order
            int mine = 0;
            for (int c = cbeg; c < cend; ++c) mine += (L.mask[c] & sbit) ? 1 : 0;
            int tot;
            int idx = blk_excl_sum_small(b, mine, ctbits, &tot);
            for (int c = cbeg; c < cend; ++c) {
                if (!(L.mask[c] & sbit)) continue;
                if (idx < cap) {
                    L.lcol[idx] = (cwt_idx_t)c;
                    L.linf[idx] = TSFA_LI_PACK(1, 0, 0, start_row);
                }
                ++idx;
            }
            nlines = tot;
        }
        if (nlines > cap) { overflow = 1; nlines = cap; }
        const int gap_thresh = 1;  // ceil(widths[0])
        for (int row = start_row - 1; row >= 0; --row) {
            const int nprev = nlines;
            const int D = (row + 1) / 4;  // max_distances[row] = widths[row] / 4 -> integer distance <= floor
            const unsigned bit = 1u << row;
            blk_sync();
            // snapshot: column -> live line (no two live lines share a last column)
            for (int l = b.tid; l < nprev; l += b.nt)
                if (!TSFA_LI_DEAD(L.linf[l])) L.colmap[L.lcol[l]] = (cwt_idx_t)(l + 1);
            blk_sync();
            // every maximum of this row picks the nearest live line (earliest line wins a distance tie)
            int fresh = 0;
            for (int c = cbeg; c < cend; ++c) {
                if (!(L.mask[c] & bit)) continue;
                int line = 0;
                for (int d = 0; d <= D && line == 0; ++d) {
                    int best = 0;
                    if (c - d >= 0 && L.colmap[c - d]) best = L.colmap[c - d];
                    if (d > 0 && c + d < n && L.colmap[c + d]) {
                        const int o = L.colmap[c + d];
                        if (best == 0 || o < best) best = o;
                    }
                    line = best;
                }
                L.mline[c] = (cwt_idx_t)line;
                fresh += (line == 0) ? 1 : 0;
            }
            int tot;
            int idx = nlines + blk_excl_sum_small(b, fresh, ctbits, &tot);
            for (int c = cbeg; c < cend; ++c) {
                if (!(L.mask[c] & bit) || L.mline[c] != 0) continue;
                if (idx < cap) {
                    L.lcol[idx] = (cwt_idx_t)c;
                    L.linf[idx] = TSFA_LI_PACK(1, 0, 0, row);
                }
                ++idx;
            }
            nlines += tot;
            if (nlines > cap) { overflow = 1; nlines = cap; }
            blk_sync();
            // every previously live line collects the maxima that chose it (ascending: the last one is its new column)
            for (int l = b.tid; l < nprev; l += b.nt) {
                const cwt_idx_t v = L.linf[l];
                if (TSFA_LI_DEAD(v)) continue;
                const int prev = L.lcol[l];
                L.colmap[prev] = 0;
                int cnt = 0, last = prev;
                for (int c = prev - D; c <= prev + D; ++c) {
                    if (c < 0 || c >= n) continue;
                    if ((L.mask[c] & bit) && L.mline[c] == l + 1) { ++cnt; last = c; }
                }
                if (cnt > 0) {
                    int len = TSFA_LI_LEN(v) + cnt;
                    if (len > 63) len = 63;
                    L.linf[l] = TSFA_LI_PACK(len, 0, 0, row);
                    L.lcol[l] = (cwt_idx_t)last;
                } else {
                    const int g = TSFA_LI_GAP(v) + 1;
                    L.linf[l] = TSFA_LI_PACK(TSFA_LI_LEN(v), g > 3 ? 3 : g, g > gap_thresh ? 1 : 0, TSFA_LI_ROW(v));
                }
            }
        }
    }
    blk_sync();
    TSFA_TICK(tk, b, 152);
    // ---- phase C ----
    const int min_length = (W + 3) / 4;             // ceil(rows / 4)
    const int window = (n + 19) / 20;               // ceil(num_points / 20)
    const int hf = window / 2, odd = window % 2;
    // cwt[row, col] of a line that ended above row 0 is re-evaluated here; its Ricker taps (exp / pow / sqrt each) are
    // tabulated once per width instead of once per tap per line:  taps[5 (w-2)(w+1) + k] = reversed tap k of width w
    const bool taps_cached = (W >= 2) && (10 * W < n) && (5 * (W - 1) * (W + 2) <= TSFA_CWTP_MAXTAPS + 16);
    if (taps_cached) {
        blk_sync();
        for (int w = 2; w <= W; ++w) {
            const int nw = 10 * w;
            double *tw = L.taps + 5 * (w - 2) * (w + 1);
            for (int k = b.tid; k < nw; k += b.nt) tw[k] = ricker_tap_t(L.rk, nw, w, nw - 1 - k);
        }
        blk_sync();
    }
    // Long series: the noise window holds hundreds of samples and its 10th percentile is no longer among the eight
    // smallest: every entry is decided by counting its window against thresholds at +-|signal| (cwt_filter_counts).
    const bool long_windows = ((int)(0.1 * (double)(window - 1)) + 1 >= 8);
    // The filter needs (column, row) of a line's end point.  The qualifying lines are compacted to the front of
    // (lcol, linf) -- entry k = (lcol[k], row in linf[k]) -- so every lane of the evaluation below has work.
    // derive_w1 (the plan also asks for n = 1, whose ridge lines are exactly the maxima of row 0, each of length 1): the
    // lines of THIS width that end on row 0 are not evaluated here but marked (bit W + 1 of mask[col]); the second list --
    // all maxima of row 0 -- then yields the n = 1 count and, through the marks, this width's row-0 lines: one noise
    // percentile per row-0 maximum for both columns of the plan, one argsort, one CWT.
    const unsigned short bit_a = (unsigned short)(1u << (W + 1));
    int cnt_a = 0;
    blk_sync();
    {   // a thread owns LT consecutive lines; the marks first
        const int LT = (nlines + b.nt - 1) / b.nt;
        int ltbits = 1;
        while ((1 << ltbits) <= LT) ++ltbits;
        const int lbeg = LT * b.tid, lend = (lbeg + LT < nlines) ? lbeg + LT : nlines;
        int mine = 0;
        for (int l = lbeg; l < lend; ++l) {
            const cwt_idx_t v = L.linf[l];
            const bool qual = TSFA_LI_LEN(v) >= min_length;
            const bool defer = derive_w1 && qual && TSFA_LI_ROW(v) == 0;
            if (defer) L.mask[L.lcol[l]] |= bit_a;
            mine += (qual && !defer) ? 1 : 0;
        }
        {
            // one scan; the entries are staged in (mline, colmap) -- dead by now -- because entry k may land on a line
            // another thread has not read yet, then copied to the front of (lcol, linf)
            int idx = blk_excl_sum_small(b, mine, ltbits, &cnt_a);
            for (int l = lbeg; l < lend; ++l) {
                const cwt_idx_t v = L.linf[l];
                const int row = TSFA_LI_ROW(v);
                if (TSFA_LI_LEN(v) < min_length || (derive_w1 && row == 0)) continue;
                L.mline[idx] = L.lcol[l];
                L.colmap[idx] = (cwt_idx_t)row;
                ++idx;
            }
            blk_sync();
            for (int k = b.tid; k < cnt_a; k += b.nt) {
                L.lcol[k] = L.mline[k];
                L.linf[k] = L.colmap[k];
            }
        }
        blk_sync();
    }
    double unused = 0.0;
    double kept;
    const ST *xp0 = (L.xpad != nullptr) ? (const ST *)L.xpad + TSFA_CWTP_HALO : nullptr;
    // the re-evaluated signal reads the staged copy of the series (LDS) where there is one
    // (mline, colmap): free again -- the second list of cwt_filter_counts
    if (long_windows) {
        blk_sync();
        if (xp0 != nullptr)
            kept = cwt_filter_counts(b, [=](int i) { return (double)xp0[i]; }, n, L, L.lcol, L.linf, cnt_a, hf, odd, taps_cached, 0, &unused, L.mline, L.colmap);
        else
            kept = cwt_filter_counts(b, xv, n, L, L.lcol, L.linf, cnt_a, hf, odd, taps_cached, 0, &unused, L.mline, L.colmap);
    } else if (xp0 != nullptr)
        kept = cwt_filter_list(b, [=](int i) { return (double)xp0[i]; }, n, L, L.lcol, L.linf, cnt_a, hf, odd, taps_cached, 0, &unused);
    else
        kept = cwt_filter_list(b, xv, n, L, L.lcol, L.linf, cnt_a, hf, odd, taps_cached, 0, &unused);
    TSFA_TICK(tk, b, 153);
    if (derive_w1) {
        int cnt_b = 0;
        blk_sync();
        {
            int mine = 0;
            for (int c = cbeg; c < cend; ++c) mine += (L.mask[c] & 1u) ? 1 : 0;
            int idx = blk_excl_sum_small(b, mine, ctbits, &cnt_b);
            for (int c = cbeg; c < cend; ++c)
                if (L.mask[c] & 1u) L.lcol[idx++] = (cwt_idx_t)c;
            blk_sync();
        }
        double marked = 0.0;
        if (long_windows) *kept_w1 = cwt_filter_counts(b, xv, n, L, L.lcol, nullptr, cnt_b, hf, odd, taps_cached, bit_a, &marked, L.mline, L.colmap);
        else *kept_w1 = cwt_filter_list(b, xv, n, L, L.lcol, nullptr, cnt_b, hf, odd, taps_cached, bit_a, &marked);
        kept += marked;
        TSFA_TICK(tk, b, 156);
    }
    return overflow ? TSFA_NAN : kept;
}

template <class ST, bool MFMA = false, class X>
TSFA_DEV void fam_cwtpeaks_series(const Blk &b, X xv, int n, const TsfaSpec *specs, int nspecs, double *out_row,
                                  const CwtPeaksLds &L) {
    // n = 1 rides along with the widest other width of the plan (<= 14): its ridge lines are the maxima of row 0, which
    // that pass computes anyway (number_cwt_peaks_one, phase C)
    int s_w1 = -1, s_host = -1, w_host = 0;
    for (int s = 0; s < nspecs; ++s) {
        const TsfaSpec sp = specs[s];
        if (sp.calc != TSFA_C_NUMBER_CWT_PEAKS) continue;
        const int W = (int)sp.p[0];
        if (W == 1 && s_w1 < 0) s_w1 = s;
        else if (W > w_host && W <= 14) { w_host = W; s_host = s; }
    }
    const bool fuse = (s_w1 >= 0 && s_host >= 0);
    for (int s = 0; s < nspecs; ++s) {
        const TsfaSpec sp = specs[s];
        if (sp.calc != TSFA_C_NUMBER_CWT_PEAKS) continue;
        if (fuse && s == s_w1) continue;
        if (fuse && s == s_host) {
            double k1 = 0.0;
            const double v = number_cwt_peaks_one<ST, MFMA>(b, xv, n, (int)sp.p[0], L, true, &k1);
            if (b.tid == 0) {
                out_row[sp.col] = v;
                out_row[specs[s_w1].col] = k1;
            }
            continue;
        }
        const double v = number_cwt_peaks_one<ST, MFMA>(b, xv, n, (int)sp.p[0], L);
        if (b.tid == 0) out_row[sp.col] = v;
    }
}

#endif
