// Family ENTROPY: sample_entropy (fc.py:1701) and approximate_entropy (fc.py:1759).
//
// Both count, for every template i of length m (and m+1), the templates j whose Chebyshev distance is within a
// tolerance r = c * np.std(x).  Distances and comparisons are done in float64 on the exact sample values so the
// counts are bit-identical to the reference's float64 arithmetic (a float32 subtraction would mis-round
// |x_i - x_j| next to a threshold; see DESIGN.md "entropy exactness").
//
// Two sweeps are implemented for m = 2:
//   entropy_sweep_sym  : every unordered pair once, per-template counters in LDS (the fast path)
//   entropy_sweep_m2   : every ordered pair, counters in registers (no LDS counters; very long series)
#ifndef TSFA_FAM_ENTROPY_H
#define TSFA_FAM_ENTROPY_H

#include "tsfa_common.h"

#define TSFA_ENT_MAXK 8
#define TSFA_ENT_GROUP 3   // thresholds per symmetric sweep (LDS holds one packed word per template per threshold)

// The tolerance r * std of a series whose variance overflowed (|x| beyond 1e154) is +inf in the reference too, and there
// every pair of templates matches (finite differences <= inf).  The sweeps pad their sorted orders with +inf sentinels,
// which an infinite tolerance would match as well: the largest finite float64 keeps the sentinels out and every real pair in.
TSFA_DEV double ent_tolerance(double t) { return (t > 1.7976931348623157e308) ? 1.7976931348623157e308 : t; }

struct EntAcc {          // per-threshold accumulators of one sweep (template length m)
    double sum_log_m;    // sum_i log(C_m[i] / (N - m + 1))
    double sum_log_m1;   // sum_i log(C_{m+1}[i] / (N - m))
    double sum_cnt_m;    // sum_i C_m[i]
    double sum_cnt_m1;   // sum_i C_{m+1}[i]
};

#if TSFA_GPU
#define TSFA_ENT_WAVE 64
#define TSFA_ENT_G 4
#else
#define TSFA_ENT_WAVE 1
#define TSFA_ENT_G 1
#endif

// number of lanes of the calling wavefront for which p holds (scalar ALU: s_bcnt1 of the compare mask)
TSFA_DEV int wave_count(bool p) {
#if TSFA_GPU
    return __popcll(__ballot(p));
#else
    return p ? 1 : 0;
#endif
}

TSFA_DEV void ent_lds_add(unsigned int *p, unsigned int v) {
#if TSFA_GPU
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // ds_add_u32, no return
#else
    *p += v;
#endif
}

struct EntCol { double x0, x1, x2; };

// The four totals of the group's k-th threshold go straight to the batch table in LDS (racc[4 * batch position]):
// no per-thread result arrays, hence no scratch memory.
TSFA_DEV void ent_store_acc(const Blk &b, double *racc, const int *gidx, int gn, int k, double slm, double slm1,
                            double scm, double scm1) {
    if (b.tid == 0 && k < gn) {
        double *d = racc + 4 * gidx[k];
        d[0] = slm;
        d[1] = slm1;
        d[2] = scm;
        d[3] = scm1;
    }
}

// Reference to the first sample of a template: on the GPU the absolute LDS byte address of xs[idx] (a ds_read
// address register as it stands: no scaling, no base add), in the emulation the index itself.
typedef unsigned int ent_ref;
#if TSFA_GPU
typedef __attribute__((address_space(3))) const double *ent_lds_cdp;
TSFA_DEV ent_ref ent_make_ref(const double *xs, int idx) {
    return (ent_ref)(uintptr_t)(__attribute__((address_space(3))) const void *)(xs + idx);
}
TSFA_DEV void ent_load_group(const double *, const ent_ref *refs, int q, EntCol *c) {
    const uint4 a = *reinterpret_cast<const uint4 *>(refs + q);  // ds_read_b128: q is a multiple of 4, refs 16-B aligned
    const ent_lds_cdp p0 = (ent_lds_cdp)a.x, p1 = (ent_lds_cdp)a.y, p2 = (ent_lds_cdp)a.z, p3 = (ent_lds_cdp)a.w;
    c[0].x0 = p0[0]; c[0].x1 = p0[1]; c[0].x2 = p0[2];
    c[1].x0 = p1[0]; c[1].x1 = p1[1]; c[1].x2 = p1[2];
    c[2].x0 = p2[0]; c[2].x1 = p2[1]; c[2].x2 = p2[2];
    c[3].x0 = p3[0]; c[3].x1 = p3[1]; c[3].x2 = p3[2];
}
#else
TSFA_DEV ent_ref ent_make_ref(const double *, int idx) { return (ent_ref)idx; }
TSFA_DEV void ent_load_group(const double *xs, const ent_ref *refs, int q, EntCol *c) {
    for (int g = 0; g < TSFA_ENT_G; ++g) { c[g].x0 = xs[refs[q + g]]; c[g].x1 = xs[refs[q + g] + 1]; c[g].x2 = xs[refs[q + g] + 2]; }
}
#endif

// One group of TSFA_ENT_G columns against the lane's row.  COL: also hand the per-column match counts of the
// wavefront (popcount of the compare masks) to lanes 0 .. G*NK-1 of colv (c2 | c3 << 16).
template <int NK, bool COL>
TSFA_DEV void ent_eval_group(double xi0, double xi1, double xi2, const EntCol *c, const double *r, int *c2, int *c3,
                             unsigned int &colv, unsigned int *cnt_emul) {
#pragma unroll
    for (int g = 0; g < TSFA_ENT_G; ++g) {
        const double d0 = fabs(xi0 - c[g].x0), d1 = fabs(xi1 - c[g].x1), d2 = fabs(xi2 - c[g].x2);
        const double m2 = fmax(d0, d1);
        const double m3 = fmax(m2, d2);
#pragma unroll
        for (int k = 0; k < NK; ++k) {
#if TSFA_GPU
            // v_cmp -> SGPR-pair mask; the row counter takes the mask as carry-in (one VALU op per predicate)
            const unsigned long long k2 = __ballot(m2 <= r[k]), k3 = __ballot(m3 <= r[k]);
            asm("v_addc_co_u32_e64 %0, vcc, 0, %0, %1" : "+v"(c2[k]) : "s"(k2) : "vcc");
            asm("v_addc_co_u32_e64 %0, vcc, 0, %0, %1" : "+v"(c3[k]) : "s"(k3) : "vcc");
            if (COL) {
                const unsigned int pk = (unsigned int)__popcll(k2) | ((unsigned int)__popcll(k3) << 16);
                asm("v_writelane_b32 %0, %1, %2" : "+v"(colv) : "s"(pk), "n"(g * NK + k));
            }
#else
            const bool p2 = (m2 <= r[k]), p3 = (m3 <= r[k]);
            c2[k] += p2 ? 1 : 0;
            c3[k] += p3 ? 1 : 0;
            if (COL) cnt_emul[g * NK + k] += (p2 ? 1u : 0u) | ((p3 ? 1u : 0u) << 16);
#endif
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Symmetric sweep.  The match relation is symmetric, so every unordered pair of templates is evaluated once.  Rows
// and columns are both taken in the sorted order perm[] (templates sorted by their first sample).  A wavefront owns
// a row block [q0, q0 + 64) of that order (blocks are dealt round-robin to the wavefronts of the workgroup):
//   * diagonal block  -- columns [q0, q0 + 64): every lane counts its own row (all 64 columns, self included);
//   * later columns   -- q in [q0 + 64, jhi), jhi = end of the r_max window of the block: lane i adds the pair to its
//                        row counter, and the number of matching LANES (popcount of the compare mask, scalar ALU) is
//                        what the COLUMN's template gains; the packed counts (c2 | c3 << 16) of four columns x NK
//                        thresholds are placed in lanes 0..4NK-1 (v_writelane) and added to the LDS counters with
//                        one ds_add_u32.
// Earlier columns are never visited (their blocks visit us).  At the end of a pass the lanes add their row
// counters to the same LDS counters; after a workgroup barrier cnt[q] holds the complete C_m / C_{m+1} of template
// perm[q].  Column groups are double-buffered in registers (the loads of the next group are issued before the
// current one is evaluated) and addressed through refs[] = absolute LDS addresses, so the inner loop carries no
// address arithmetic and no LDS latency.
// Requirements: xs[n .. n+3] = +inf; perm[] / refs[] padded with the template n (the sentinels) up to
// roundup(n-1, 64) + 32 entries; cnt[] holds (n + 16) * NK words.
// ---------------------------------------------------------------------------------------------------------------
template <int NK>
TSFA_DEV void entropy_sweep_sym(const Blk &b, const double *xs, int n, const double *thr, const unsigned short *perm,
                                const ent_ref *refs, unsigned int *cnt, double *racc, const int *gidx, int gn) {
    const int W = TSFA_ENT_WAVE, G = TSFA_ENT_G;
    const int nrow_m = n - 1;   // templates of length 2
    const int nrow_m1 = n - 2;  // templates of length 3
    const int lane = b.tid % W, wave = b.tid / W, nwave = b.nt / W;
    double r[NK];
    double rmax = 0.0;
#pragma unroll
    for (int k = 0; k < NK; ++k) { r[k] = thr[k]; rmax = fmax(rmax, r[k]); }
    blk_sync();
    for (int i = b.tid; i < (nrow_m + 16) * NK; i += b.nt) cnt[i] = 0u;
    blk_sync();

    TSFA_TICKER(tks, 0);
    const int npass = (nrow_m + W - 1) / W;
    for (int pass = wave; pass < npass; pass += nwave) {
        const int q0 = pass * W;
        const int qi = q0 + lane;
        const bool row_m = (qi < nrow_m);
        const int ri = row_m ? (int)perm[qi] : n;  // xs[n ..] = +inf: an absent row never matches
        const double xi0 = xs[ri], xi1 = xs[ri + 1], xi2 = xs[ri + 2];
        const int qlast = ((q0 + W < nrow_m) ? (q0 + W) : nrow_m) - 1;
        const double band_hi = xs[perm[qlast]];
        const double key_hi = band_hi + rmax + 1e-9 * (fabs(band_hi) + rmax);
        int jhi;
        {
            int lo = qlast + 1, hi = nrow_m;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (xs[perm[mid]] <= key_hi) lo = mid + 1; else hi = mid; }
            jhi = lo;
        }
        const int qdiag_end = q0 + W;
        int c2[NK], c3[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) { c2[k] = 0; c3[k] = 0; }
        unsigned int colv = 0u;

        // ---- diagonal block: row counters only (W / G groups: an even number on the GPU) ----
        EntCol ca[TSFA_ENT_G], cb[TSFA_ENT_G];
        int q = q0;
        ent_load_group(xs, refs, q, ca);
#if TSFA_GPU
        for (; q < qdiag_end; q += 2 * G) {
            ent_load_group(xs, refs, q + G, cb);
            ent_eval_group<NK, false>(xi0, xi1, xi2, ca, r, c2, c3, colv, nullptr);
            ent_load_group(xs, refs, q + 2 * G, ca);
            ent_eval_group<NK, false>(xi0, xi1, xi2, cb, r, c2, c3, colv, nullptr);
        }
        // ---- later columns: row counters + column counters; a trailing odd group lies beyond the window and
        //      matches nothing (it adds zeros) ----
        for (; q < jhi; q += 2 * G) {
            ent_load_group(xs, refs, q + G, cb);
            colv = 0u;
            ent_eval_group<NK, true>(xi0, xi1, xi2, ca, r, c2, c3, colv, nullptr);
            if (lane < G * NK) ent_lds_add(&cnt[q * NK + lane], colv);
            ent_load_group(xs, refs, q + 2 * G, ca);
            colv = 0u;
            ent_eval_group<NK, true>(xi0, xi1, xi2, cb, r, c2, c3, colv, nullptr);
            if (lane < G * NK) ent_lds_add(&cnt[(q + G) * NK + lane], colv);
        }
#else
        (void)cb;
        for (; q < qdiag_end; q += G) {
            ent_eval_group<NK, false>(xi0, xi1, xi2, ca, r, c2, c3, colv, nullptr);
            ent_load_group(xs, refs, q + G, ca);
        }
        for (; q < jhi; q += G) {
            ent_eval_group<NK, true>(xi0, xi1, xi2, ca, r, c2, c3, colv, &cnt[q * NK]);
            ent_load_group(xs, refs, q + G, ca);
        }
#endif
        if (row_m) {
#pragma unroll
            for (int k = 0; k < NK; ++k) ent_lds_add(&cnt[qi * NK + k], (unsigned int)c2[k] | ((unsigned int)c3[k] << 16));
        }
    }
    TSFA_TICK(tks, b, 136);
    blk_sync();
    TSFA_TICK(tks, b, 137);
    // ---- totals: cnt[q] is now complete for every template ----
    // sum_i log(C_i / N) = log(prod_i C_i) - (#rows) * log(N): the counts are integers <= 2^16, so a thread multiplies
    // up to 16 of them into one double (< 2^1024, relative error 1e-16 per factor) and takes ONE logarithm.  Rows
    // with C_i == N contribute exactly 0, as in the reference (log(1.0)).
    double slm[NK], slm1[NK], scm[NK], scm1[NK], pm[NK], pm1[NK];
    int nm[NK], nm1[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) { slm[k] = 0.0; slm1[k] = 0.0; scm[k] = 0.0; scm1[k] = 0.0; pm[k] = 1.0; pm1[k] = 1.0; nm[k] = 0; nm1[k] = 0; }
    int it = 0;
    for (int q0 = 0; q0 < nrow_m; q0 += b.nt, ++it) {
        const int q = q0 + b.tid;
        if (q < nrow_m) {
            const bool row_m1 = ((int)perm[q] < nrow_m1);
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const unsigned int cc = cnt[q * NK + k];
                const int t2 = (int)(cc & 0xFFFFu), t3 = (int)(cc >> 16);
                scm[k] += (double)t2;
                if (t2 != nrow_m) { pm[k] *= (double)t2; ++nm[k]; }
                if (row_m1) {
                    scm1[k] += (double)t3;
                    if (t3 != nrow_m1) { pm1[k] *= (double)t3; ++nm1[k]; }
                }
            }
        }
        if ((it & 15) == 15) {
#pragma unroll
            for (int k = 0; k < NK; ++k) { slm[k] += log(pm[k]); slm1[k] += log(pm1[k]); pm[k] = 1.0; pm1[k] = 1.0; }
        }
    }
    const double ldm = log((double)nrow_m), ldm1 = log((double)nrow_m1);
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        slm[k] += log(pm[k]) - (double)nm[k] * ldm;
        slm1[k] += log(pm1[k]) - (double)nm1[k] * ldm1;
        ent_store_acc(b, racc, gidx, gn, k, blk_sum(b, slm[k]), blk_sum(b, slm1[k]), blk_sum(b, scm[k]), blk_sum(b, scm1[k]));
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Staged symmetric sweep (n <= TSFA_ENT_STAGED_MAXN, 4..6 thresholds): ALL thresholds in one sweep.  The distance
// evaluation of a pair (3 subtractions, 2 maxima) is shared by every threshold, and a threshold is only compared
// inside ITS OWN window: thresholds ascend, thr[k] < 0 for the unused leading slots, and the columns after the
// diagonal block are swept in six segments -- segment S ends where the window of threshold S ends and evaluates the
// thresholds S..5 (the smaller ones cannot match any more: columns are sorted by first sample).  Compared with two
// grouped sweeps this shares the 5 distance operations and drops the compares of the small tolerances over most
// of the large window (about -25 % vector instructions at L = 1024, six thresholds).
// Counters: C <= n - 1 <= 1023 fits 10 bits, so a template owns four LDS words: C_m of thresholds 0..2, C_m of 3..5,
// C_{m+1} of 0..2, C_{m+1} of 3..5 (three 10-bit fields each; fields never overflow into their neighbours).
// ---------------------------------------------------------------------------------------------------------------
#define TSFA_ENT_STAGED_K 6
#define TSFA_ENT_STAGED_MAXN 1024

template <int S, bool COL>
TSFA_DEV void ent_eval_staged(double xi0, double xi1, double xi2, const EntCol *c, const double *r, int *c2, int *c3,
                              unsigned int &colv, unsigned int *cnt_emul) {
#pragma unroll
    for (int g = 0; g < TSFA_ENT_G; ++g) {
        const double d0 = fabs(xi0 - c[g].x0), d1 = fabs(xi1 - c[g].x1), d2 = fabs(xi2 - c[g].x2);
        const double m2 = fmax(d0, d1);
        const double m3 = fmax(m2, d2);
        unsigned int w0 = 0u, w1 = 0u, w2 = 0u, w3 = 0u;
#pragma unroll
        for (int k = S; k < TSFA_ENT_STAGED_K; ++k) {
#if TSFA_GPU
            const unsigned long long k2 = __ballot(m2 <= r[k]), k3 = __ballot(m3 <= r[k]);
            asm("v_addc_co_u32_e64 %0, vcc, 0, %0, %1" : "+v"(c2[k]) : "s"(k2) : "vcc");
            asm("v_addc_co_u32_e64 %0, vcc, 0, %0, %1" : "+v"(c3[k]) : "s"(k3) : "vcc");
            const unsigned int n2 = (unsigned int)__popcll(k2), n3 = (unsigned int)__popcll(k3);
#else
            const unsigned int n2 = (m2 <= r[k]) ? 1u : 0u, n3 = (m3 <= r[k]) ? 1u : 0u;
            c2[k] += (int)n2;
            c3[k] += (int)n3;
#endif
            if (COL) {
                if (k < 3) { w0 |= n2 << (10 * k); w2 |= n3 << (10 * k); }
                else { w1 |= n2 << (10 * (k - 3)); w3 |= n3 << (10 * (k - 3)); }
            }
        }
        if (COL) {
#if TSFA_GPU
            if (S < 3) {
                asm("v_writelane_b32 %0, %1, %2" : "+v"(colv) : "s"(w0), "n"(g * 4 + 0));
                asm("v_writelane_b32 %0, %1, %2" : "+v"(colv) : "s"(w2), "n"(g * 4 + 2));
            }
            asm("v_writelane_b32 %0, %1, %2" : "+v"(colv) : "s"(w1), "n"(g * 4 + 1));
            asm("v_writelane_b32 %0, %1, %2" : "+v"(colv) : "s"(w3), "n"(g * 4 + 3));
#else
            cnt_emul[g * 4 + 0] += w0; cnt_emul[g * 4 + 1] += w1; cnt_emul[g * 4 + 2] += w2; cnt_emul[g * 4 + 3] += w3;
#endif
        }
    }
}

// one segment of the staged sweep: columns [q, qend) with the thresholds S..5 (qend - q is a multiple of 2 G)
template <int S>
TSFA_DEV void ent_staged_segment(const double *xs, const ent_ref *refs, unsigned int *cnt, int lane, int &q, int qend,
                                 double xi0, double xi1, double xi2, EntCol *ca, EntCol *cb, const double *r, int *c2,
                                 int *c3) {
    const int G = TSFA_ENT_G;
    unsigned int colv;
#if TSFA_GPU
    for (; q < qend; q += 2 * G) {
        ent_load_group(xs, refs, q + G, cb);
        colv = 0u;
        ent_eval_staged<S, true>(xi0, xi1, xi2, ca, r, c2, c3, colv, nullptr);
        if (lane < G * 4) ent_lds_add(&cnt[q * 4 + lane], colv);
        ent_load_group(xs, refs, q + 2 * G, ca);
        colv = 0u;
        ent_eval_staged<S, true>(xi0, xi1, xi2, cb, r, c2, c3, colv, nullptr);
        if (lane < G * 4) ent_lds_add(&cnt[(q + G) * 4 + lane], colv);
    }
#else
    (void)cb; (void)lane;
    for (; q < qend; q += G) {
        colv = 0u;
        ent_eval_staged<S, true>(xi0, xi1, xi2, ca, r, c2, c3, colv, &cnt[q * 4]);
        ent_load_group(xs, refs, q + G, ca);
    }
#endif
}

// thr[0..5] ascending (unused leading slots < 0); gidx[k] = batch position of thr[k] or -1; cnt holds (n + 16) * 4 words
TSFA_DEV void entropy_sweep_staged(const Blk &b, const double *xs, int n, const double *thr, const unsigned short *perm,
                                   const ent_ref *refs, unsigned int *cnt, double *racc, const int *gidx) {
    const int W = TSFA_ENT_WAVE, G = TSFA_ENT_G, K = TSFA_ENT_STAGED_K;
    const int nrow_m = n - 1;   // templates of length 2
    const int nrow_m1 = n - 2;  // templates of length 3
    const int lane = b.tid % W, wave = b.tid / W, nwave = b.nt / W;
    double r[K];
#pragma unroll
    for (int k = 0; k < K; ++k) r[k] = thr[k];
    blk_sync();
    for (int i = b.tid; i < (nrow_m + 16) * 4; i += b.nt) cnt[i] = 0u;
    blk_sync();

    TSFA_TICKER(tks, 0);
    const int npass = (nrow_m + W - 1) / W;
    for (int pass = wave; pass < npass; pass += nwave) {
        const int q0 = pass * W;
        const int qi = q0 + lane;
        const bool row_m = (qi < nrow_m);
        const int ri = row_m ? (int)perm[qi] : n;  // xs[n ..] = +inf: an absent row never matches
        const double xi0 = xs[ri], xi1 = xs[ri + 1], xi2 = xs[ri + 2];
        const int qlast = ((q0 + W < nrow_m) ? (q0 + W) : nrow_m) - 1;
        const int qdiag_end = q0 + W;
        const double band_hi = xs[perm[qlast]];
        // end of every threshold's window, rounded up to the double-buffered step (extra columns match nothing)
        int qend[K];
#if TSFA_GPU
        {
            const double rk = thr[(lane < K) ? lane : (K - 1)];  // lane k searches the window of threshold k
            const double key_hi = band_hi + rk + 1e-9 * (fabs(band_hi) + rk);
            int lo = qlast + 1, hi = nrow_m;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (xs[perm[mid]] <= key_hi) lo = mid + 1; else hi = mid; }
            const int e = (((lo > qdiag_end) ? lo : qdiag_end) + 2 * G - 1) & ~(2 * G - 1);
#pragma unroll
            for (int k = 0; k < K; ++k) qend[k] = __builtin_amdgcn_readlane(e, k);
        }
#else
        for (int k = 0; k < K; ++k) {
            const double key_hi = band_hi + thr[k] + 1e-9 * (fabs(band_hi) + thr[k]);
            int lo = qlast + 1, hi = nrow_m;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (xs[perm[mid]] <= key_hi) lo = mid + 1; else hi = mid; }
            qend[k] = (((lo > qdiag_end) ? lo : qdiag_end) + 7) & ~7;
        }
#endif
        int c2[K], c3[K];
#pragma unroll
        for (int k = 0; k < K; ++k) { c2[k] = 0; c3[k] = 0; }
        unsigned int colv = 0u;

        // ---- diagonal block: row counters only ----
        EntCol ca[TSFA_ENT_G], cb[TSFA_ENT_G];
        int q = q0;
        ent_load_group(xs, refs, q, ca);
#if TSFA_GPU
        for (; q < qdiag_end; q += 2 * G) {
            ent_load_group(xs, refs, q + G, cb);
            ent_eval_staged<0, false>(xi0, xi1, xi2, ca, r, c2, c3, colv, nullptr);
            ent_load_group(xs, refs, q + 2 * G, ca);
            ent_eval_staged<0, false>(xi0, xi1, xi2, cb, r, c2, c3, colv, nullptr);
        }
#else
        for (; q < qdiag_end; q += G) {
            ent_eval_staged<0, false>(xi0, xi1, xi2, ca, r, c2, c3, colv, nullptr);
            ent_load_group(xs, refs, q + G, ca);
        }
#endif
        // ---- later columns: segment S = thresholds S..5 up to the end of window S ----
        ent_staged_segment<0>(xs, refs, cnt, lane, q, qend[0], xi0, xi1, xi2, ca, cb, r, c2, c3);
        ent_staged_segment<1>(xs, refs, cnt, lane, q, qend[1], xi0, xi1, xi2, ca, cb, r, c2, c3);
        ent_staged_segment<2>(xs, refs, cnt, lane, q, qend[2], xi0, xi1, xi2, ca, cb, r, c2, c3);
        ent_staged_segment<3>(xs, refs, cnt, lane, q, qend[3], xi0, xi1, xi2, ca, cb, r, c2, c3);
        ent_staged_segment<4>(xs, refs, cnt, lane, q, qend[4], xi0, xi1, xi2, ca, cb, r, c2, c3);
        ent_staged_segment<5>(xs, refs, cnt, lane, q, qend[5], xi0, xi1, xi2, ca, cb, r, c2, c3);
        if (row_m) {
            ent_lds_add(&cnt[qi * 4 + 0], (unsigned int)c2[0] | ((unsigned int)c2[1] << 10) | ((unsigned int)c2[2] << 20));
            ent_lds_add(&cnt[qi * 4 + 1], (unsigned int)c2[3] | ((unsigned int)c2[4] << 10) | ((unsigned int)c2[5] << 20));
            ent_lds_add(&cnt[qi * 4 + 2], (unsigned int)c3[0] | ((unsigned int)c3[1] << 10) | ((unsigned int)c3[2] << 20));
            ent_lds_add(&cnt[qi * 4 + 3], (unsigned int)c3[3] | ((unsigned int)c3[4] << 10) | ((unsigned int)c3[5] << 20));
        }
    }
    TSFA_TICK(tks, b, 136);
    blk_sync();
    TSFA_TICK(tks, b, 137);
    // ---- totals (as in entropy_sweep_sym), three thresholds at a time to bound the live registers ----
    const double ldm = log((double)nrow_m), ldm1 = log((double)nrow_m1);
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        double slm[3], slm1[3], scm[3], scm1[3], pm[3], pm1[3];
        int nm[3], nm1[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { slm[k] = 0.0; slm1[k] = 0.0; scm[k] = 0.0; scm1[k] = 0.0; pm[k] = 1.0; pm1[k] = 1.0; nm[k] = 0; nm1[k] = 0; }
        int it = 0;
        for (int qb = 0; qb < nrow_m; qb += b.nt, ++it) {
            const int q = qb + b.tid;
            if (q < nrow_m) {
                const bool row_m1 = ((int)perm[q] < nrow_m1);
                const unsigned int cw2 = cnt[q * 4 + h], cw3 = cnt[q * 4 + 2 + h];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int t2 = (int)((cw2 >> (10 * k)) & 1023u), t3 = (int)((cw3 >> (10 * k)) & 1023u);
                    scm[k] += (double)t2;
                    if (t2 != nrow_m) { pm[k] *= (double)t2; ++nm[k]; }
                    if (row_m1) {
                        scm1[k] += (double)t3;
                        if (t3 != nrow_m1) { pm1[k] *= (double)t3; ++nm1[k]; }
                    }
                }
            }
            if ((it & 15) == 15) {
#pragma unroll
                for (int k = 0; k < 3; ++k) { slm[k] += log(pm[k]); slm1[k] += log(pm1[k]); pm[k] = 1.0; pm1[k] = 1.0; }
            }
        }
        double tot[12];  // the twelve totals of this half reduced together: one barrier pair instead of twelve
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            tot[4 * k + 0] = slm[k] + (log(pm[k]) - (double)nm[k] * ldm);
            tot[4 * k + 1] = slm1[k] + (log(pm1[k]) - (double)nm1[k] * ldm1);
            tot[4 * k + 2] = scm[k];
            tot[4 * k + 3] = scm1[k];
        }
        blk_sum_multi<12>(b, tot);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int pos = gidx[3 * h + k];
            if (b.tid == 0 && pos >= 0) {
                double *d = racc + 4 * pos;
                d[0] = tot[4 * k + 0]; d[1] = tot[4 * k + 1]; d[2] = tot[4 * k + 2]; d[3] = tot[4 * k + 3];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Ordered-pair sweep (no LDS counters): thread = row i, the loop runs over the columns j of the row block's r_max
// window, x[j..j+2] is a wave-uniform broadcast read, per-row counters live in registers.
// xs[n], xs[n+1] must hold +inf (the length-3 extension of the last templates then never matches).
// ---------------------------------------------------------------------------------------------------------------
template <int NK, typename XT, typename IDX>
TSFA_DEV void entropy_sweep_m2(const Blk &b, const XT *xs, int n, const double *thr, const IDX *perm,
                               double *racc, const int *gidx, int gn) {
    const int nrow_m = n - 1;   // templates of length 2: i in [0, n-2]
    const int nrow_m1 = n - 2;  // templates of length 3: i in [0, n-3]
    double r[NK];
    double rmax = 0.0;
#pragma unroll
    for (int k = 0; k < NK; ++k) { r[k] = thr[k]; rmax = fmax(rmax, r[k]); }
    double slm[NK], slm1[NK], scm[NK], scm1[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) { slm[k] = 0.0; slm1[k] = 0.0; scm[k] = 0.0; scm1[k] = 0.0; }
    const double dm = (double)nrow_m, dm1 = (double)nrow_m1;

    const int npass = (nrow_m + b.nt - 1) / b.nt;
    for (int pass = 0; pass < npass; ++pass) {
        const int q0 = pass * b.nt;
        const int q1 = (q0 + b.nt < nrow_m) ? (q0 + b.nt) : nrow_m;  // rows [q0, q1) of the sorted order
        const int qi = q0 + b.tid;
        const bool row_m = (qi < nrow_m);
        const int ri = row_m ? (int)perm[qi] : 0;
        const bool row_m1 = row_m && (ri < nrow_m1);
        const double xi0 = (double)xs[ri], xi1 = (double)xs[ri + 1], xi2 = (double)xs[ri + 2];
        // column window: first samples within r_max of [band_lo, band_hi], widened by a rounding margin
        const double band_lo = (double)xs[perm[q0]], band_hi = (double)xs[perm[q1 - 1]];
        const double margin = 1e-9 * (fabs(band_lo) + fabs(band_hi) + rmax);
        const double key_lo = band_lo - rmax - margin, key_hi = band_hi + rmax + margin;
        int jlo, jhi;
        {
            int lo = 0, hi = nrow_m;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if ((double)xs[perm[mid]] < key_lo) lo = mid + 1; else hi = mid; }
            jlo = lo;
            lo = jlo; hi = nrow_m;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if ((double)xs[perm[mid]] <= key_hi) lo = mid + 1; else hi = mid; }
            jhi = lo;
        }
        int c2[NK], c3[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) { c2[k] = 0; c3[k] = 0; }
        for (int q = jlo; q < jhi; ++q) {
            const int c = (int)perm[q];
            const double xj0 = (double)xs[c], xj1 = (double)xs[c + 1], xj2 = (double)xs[c + 2];
            const double d0 = fabs(xi0 - xj0), d1 = fabs(xi1 - xj1), d2 = fabs(xi2 - xj2);
            const double m2 = fmax(d0, d1);
            const double m3 = fmax(m2, d2);
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                c2[k] += (m2 <= r[k]) ? 1 : 0;
                c3[k] += (m3 <= r[k]) ? 1 : 0;
            }
        }
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            if (row_m) {
                scm[k] += (double)c2[k];
                slm[k] += log((double)c2[k] / dm);
            }
            if (row_m1) {
                scm1[k] += (double)c3[k];
                slm1[k] += log((double)c3[k] / dm1);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NK; ++k)
        ent_store_acc(b, racc, gidx, gn, k, blk_sum(b, slm[k]), blk_sum(b, slm1[k]), blk_sum(b, scm[k]), blk_sum(b, scm1[k]));
}

// perm[0 .. n-2] = indices of the length-2 templates sorted by their first sample (ties by index); np2 = padded size
#if TSFA_GPU
// F32: the samples are exact float32 values (float32 input staged as float64): one packed 64-bit sort key
template <int E, typename XT>
TSFA_DEVN void entropy_sort_templates_packed(const Blk b, const XT *xs, int n, unsigned short *perm) {
    const int nrow_m = n - 1;
    unsigned long long pk[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int g = b.tid * E + e;
        pk[e] = (g < nrow_m) ? sort_pack_f32((float)xs[g], g) : sort_pack_f32((float)TSFA_INF, 0xFFFF);
    }
    blk_sort_packed_regs<E>(b, pk, perm, [=](int i) { return (i == 0xFFFF) ? (float)TSFA_INF : (float)xs[i]; });
    blk_sync();
#pragma unroll
    for (int e = 0; e < E; ++e) perm[b.tid * E + e] = (unsigned short)(pk[e] & 0xFFFFull);
    blk_sync();
}

template <int E, typename XT>
TSFA_DEVN void entropy_sort_templates_regs(const Blk b, const XT *xs, int n, unsigned short *perm) {
    const int nrow_m = n - 1;
    double key[E];
    int idx[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int g = b.tid * E + e;
        idx[e] = (g < nrow_m) ? g : 0xFFFF;
        key[e] = (g < nrow_m) ? (double)xs[g] : TSFA_INF;
    }
    blk_sort_pairs_regs<E>(b, key, idx, perm, [=](int i) { return (i == 0xFFFF) ? TSFA_INF : (double)xs[i]; });
    blk_sync();
#pragma unroll
    for (int e = 0; e < E; ++e) perm[b.tid * E + e] = (unsigned short)idx[e];
    blk_sync();
}
#endif

// IDX: unsigned short (series up to 65 535 samples: the LDS build, the bit-matrix sweeps) or unsigned int (the pair sweep of the
// long-series build: any length)
template <typename XT, typename IDX>
TSFA_DEV void entropy_sort_templates(const Blk &b, const XT *xs, int n, IDX *perm, int np2, bool f32 = false) {
    const int nrow_m = n - 1;
    const IDX none = (IDX)~(IDX)0;
#if TSFA_GPU
    if constexpr (sizeof(IDX) == 2) {
    if (f32) {
        if (np2 == b.nt) { entropy_sort_templates_packed<1>(b, xs, n, perm); return; }
        if (np2 == 2 * b.nt) { entropy_sort_templates_packed<2>(b, xs, n, perm); return; }
        if (np2 == 4 * b.nt) { entropy_sort_templates_packed<4>(b, xs, n, perm); return; }
    }
    // register-blocked sort when every thread gets 1, 2 or 4 templates (np2 = E * nt)
    if (np2 == b.nt) { entropy_sort_templates_regs<1>(b, xs, n, perm); return; }
    if (np2 == 2 * b.nt) { entropy_sort_templates_regs<2>(b, xs, n, perm); return; }
    if (np2 == 4 * b.nt) { entropy_sort_templates_regs<4>(b, xs, n, perm); return; }
    }
#endif
    blk_sync();
    for (int i = b.tid; i < np2; i += b.nt) perm[i] = (i < nrow_m) ? (IDX)i : none;
    for (int k = 2; k <= np2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            blk_sync();
            for (int t = b.tid; t < (np2 >> 1); t += b.nt) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const bool up = ((i & k) == 0);
                const IDX a = perm[i], c = perm[l];
                const double ka = (a == none) ? TSFA_INF : (double)xs[a], kc = (c == none) ? TSFA_INF : (double)xs[c];
                const bool gt = (ka > kc) || (ka == kc && a > c);
                if (gt == up) {
                    perm[i] = c;
                    perm[l] = a;
                }
            }
        }
    }
    blk_sync();
}

// generic template length m (slow path; settings.py only uses m = 2)
template <typename XT>
TSFA_DEV void entropy_sweep_generic(const Blk &b, const XT *xs, int n, int m, double thr, EntAcc *acc) {
    const int nrow_m = n - m + 1, nrow_m1 = n - m;
    double slm = 0.0, slm1 = 0.0, scm = 0.0, scm1 = 0.0;
    for (int i = b.tid; i < nrow_m; i += b.nt) {
        int cm = 0, cm1 = 0;
        for (int j = 0; j < nrow_m; ++j) {
            double d = 0.0;
            for (int t = 0; t < m; ++t) d = fmax(d, fabs((double)xs[i + t] - (double)xs[j + t]));
            cm += (d <= thr) ? 1 : 0;
            if (i < nrow_m1 && j < nrow_m1) {
                d = fmax(d, fabs((double)xs[i + m] - (double)xs[j + m]));
                cm1 += (d <= thr) ? 1 : 0;
            }
        }
        scm += (double)cm;
        slm += log((double)cm / (double)nrow_m);
        if (i < nrow_m1) {
            scm1 += (double)cm1;
            slm1 += log((double)cm1 / (double)nrow_m1);
        }
    }
    acc->sum_log_m = blk_sum(b, slm);
    acc->sum_log_m1 = blk_sum(b, slm1);
    acc->sum_cnt_m = blk_sum(b, scm);
    acc->sum_cnt_m1 = blk_sum(b, scm1);
}

TSFA_DEV double apen_from_acc(const EntAcc &a, int n, int m) {
    // fc.py:1797-1805: |phi(m) - phi(m+1)|, phi(m) = sum(log(C)) / (N - m + 1.0)
    const double phi_m = a.sum_log_m / ((double)(n - m) + 1.0);
    const double phi_m1 = a.sum_log_m1 / ((double)(n - (m + 1)) + 1.0);
    return fabs(phi_m - phi_m1);
}
TSFA_DEV double sampen_from_acc(const EntAcc &a, int n, int m) {
    // fc.py:1729-1754: B = sum_i (C_m[i] - 1), A = sum_i (C_{m+1}[i] - 1); -log(A / B)
    const double B = a.sum_cnt_m - (double)(n - m + 1);
    const double A = a.sum_cnt_m1 - (double)(n - m);
    return -log(A / B);
}

// Evaluate the ENTROPY specs of one series.
//   xs   : LDS, n + 4 elements in the input precision (xs[n .. n+3] are overwritten with +inf sentinels)
//   thr  : LDS scratch >= 7 * TSFA_ENT_MAXK doubles (batch thresholds, group thresholds, totals, group index)
//   perm : LDS, max(next_pow2(n), 64) + 32 unsigned shorts
//   refs : LDS, as many ent_ref as perm (symmetric sweep only)
//   cnt  : LDS, (n + 16) * TSFA_ENT_GROUP words, or null (-> ordered-pair sweep).  May alias b.np: the numpy-order
//          sums are finished before the first sweep.  The symmetric sweep needs XT = double.
//   FAST : the plan holds only m = 2 specs and cnt != null (decided on the host): the ordered-pair and generic
//          sweeps are compiled out, which keeps the register allocation of the hot kernel free of spills.
template <typename XT, bool FAST = false, bool F32 = false, typename IDX = unsigned short>
TSFA_DEV void fam_entropy_series(const Blk &b0, XT *xs, int n, const TsfaSpec *specs, int nspecs, double *out_row,
                                 double *thr, IDX *perm, ent_ref *refs, unsigned int *cnt,
                                 int staged_ok = 1, const double *stats = nullptr) {
    const Blk &b = b0;
    // np.std(x), numpy summation order (the tolerances are c * np.std(x)); stats: the record k_basic left (TSFA_STATS_*)
    TSFA_TICKER(tk, 0);
    const double dn = (double)n;
    const double mean = stats ? stats[TSFA_STATS_MEAN] : np_sum(b, n, [=](int i) { return (double)xs[i]; }) / dn;
    const double var = stats ? stats[TSFA_STATS_VAR] : np_sum(b, n, [=](int i) { const double d = (double)xs[i] - mean; return d * d; }) / dn;
    const double sd = sqrt(var);
    blk_sync();
    if (b.tid == 0) { xs[n] = (XT)TSFA_INF; xs[n + 1] = (XT)TSFA_INF; xs[n + 2] = (XT)TSFA_INF; xs[n + 3] = (XT)TSFA_INF; }
    bool sorted = false;
    TSFA_TICK(tk, b, 130);

    // m = 2 specs are batched TSFA_ENT_MAXK at a time
    int done = 0;
    while (done < nspecs) {
#if TSFA_GPU
        // the thread index is made opaque per batch: everything derived from it inside the sweeps (addresses, lane
        // predicates) is then computed where it is used instead of being hoisted to the kernel's top, kept live over
        // the whole loop and spilled (scratch traffic was 10x the algorithmic HBM bytes of this kernel)
        int tid_opaque = b0.tid;
        asm volatile("" : "+v"(tid_opaque));
        const Blk b{tid_opaque, b0.nt, b0.red, b0.np};  // shadows the function-level alias of b0
#endif
        int nk = 0;
        int s = done;
        for (; s < nspecs && nk < TSFA_ENT_MAXK; ++s) {
            const TsfaSpec sp = specs[s];
            const bool is_m2 = (sp.calc == TSFA_C_SAMPLE_ENTROPY) ||
                               (sp.calc == TSFA_C_APPROXIMATE_ENTROPY && (int)sp.p[0] == 2);
            if (!is_m2) continue;
            blk_sync();
            if (b.tid == 0) thr[nk] = ent_tolerance((sp.calc == TSFA_C_SAMPLE_ENTROPY) ? 0.2 * sd : sp.p[1] * sd);
            ++nk;
        }
        const int first = done;
        done = s;
        if (nk == 0) break;
        blk_sync();
        if (b.tid == 0)
            for (int k = nk; k < TSFA_ENT_MAXK; ++k) thr[k] = -1.0;  // never matches
        blk_sync();
        if (n >= 3 && !sorted) {
            entropy_sort_templates(b, xs, n, perm, next_pow2(n - 1), F32);
            // pad: absent templates point at the +inf sentinels
            const int padded = ((n - 1 + 63) / 64) * 64 + 32;
            for (int i = n - 1 + b.tid; i < padded; i += b.nt) perm[i] = (IDX)n;
            blk_sync();
            if ((FAST || cnt != nullptr) && sizeof(IDX) == 2) {
                for (int i = b.tid; i < padded; i += b.nt) refs[i] = ent_make_ref((const double *)(const void *)xs, (int)perm[i]);
                blk_sync();
            }
            sorted = true;
            TSFA_TICK(tk, b, 131);
        }
        // Thresholds are swept in ascending order in groups of <= TSFA_ENT_GROUP neighbours (a group of small
        // tolerances only visits the narrow window of ITS largest one).  gthr = the group's thresholds; racc = the
        // four totals of every threshold of the batch, indexed by its position in the batch (LDS: no dynamically
        // indexed register arrays).
        double *gthr = thr + TSFA_ENT_MAXK;
        double *racc = thr + 2 * TSFA_ENT_MAXK;  // [TSFA_ENT_MAXK][4]
        if ((FAST || cnt != nullptr) && sizeof(IDX) == 2 && staged_ok && nk >= 4 && nk <= TSFA_ENT_STAGED_K && n >= 3 &&
            n <= TSFA_ENT_STAGED_MAXN) {
            // all thresholds of the batch in one staged sweep: ascending, right-aligned in gthr[0..5]
            int *gidx = (int *)(racc + 4 * TSFA_ENT_MAXK);
            blk_sync();
            if (b.tid == 0) {
                for (int k = 0; k < TSFA_ENT_MAXK; ++k) { gthr[k] = -1.0; gidx[k] = -1; }
                for (int k = 0; k < nk; ++k) {
                    int rk = 0;
                    for (int o = 0; o < nk; ++o) rk += (thr[o] < thr[k] || (thr[o] == thr[k] && o < k)) ? 1 : 0;
                    gthr[TSFA_ENT_STAGED_K - nk + rk] = thr[k];
                    gidx[TSFA_ENT_STAGED_K - nk + rk] = k;
                }
            }
            blk_sync();
            TSFA_TICK(tk, b, 132);
            if constexpr (sizeof(IDX) == 2) entropy_sweep_staged(b, (const double *)(const void *)xs, n, gthr, perm, refs, cnt, racc, gidx);
            TSFA_TICK(tk, b, 133);
        } else {
        const bool sym = (FAST || cnt != nullptr) && sizeof(IDX) == 2;   // (the counter sweeps hold 16-bit template references)
        const int gcap = sym ? TSFA_ENT_GROUP : TSFA_ENT_MAXK;
        const int ngroups = (nk + gcap - 1) / gcap;
        const int gsize = (nk + ngroups - 1) / ngroups;
        for (int g0 = 0; n >= 3 && g0 < nk; g0 += gsize) {
            const int gn = (nk - g0 < gsize) ? (nk - g0) : gsize;
            int *gidx = (int *)(racc + 4 * TSFA_ENT_MAXK);  // batch position of the group's k-th threshold
            blk_sync();
            if (b.tid == 0) {
                for (int k = 0; k < TSFA_ENT_MAXK; ++k) { gthr[k] = -1.0; gidx[k] = -1; }
                for (int k = 0; k < nk; ++k) {  // rank of thr[k] among the batch (ties by index)
                    int rk = 0;
                    for (int o = 0; o < nk; ++o) rk += (thr[o] < thr[k] || (thr[o] == thr[k] && o < k)) ? 1 : 0;
                    if (rk >= g0 && rk < g0 + gn) { gthr[rk - g0] = thr[k]; gidx[rk - g0] = k; }
                }
            }
            blk_sync();
            TSFA_TICK(tk, b, 132);
            if (sym) {
                if constexpr (sizeof(IDX) == 2) {
                const double *xd = (const double *)(const void *)xs;  // cnt != null implies XT = double
                if (gn <= 1) entropy_sweep_sym<1>(b, xd, n, gthr, perm, refs, cnt, racc, gidx, gn);
                else if (gn == 2) entropy_sweep_sym<2>(b, xd, n, gthr, perm, refs, cnt, racc, gidx, gn);
                else entropy_sweep_sym<3>(b, xd, n, gthr, perm, refs, cnt, racc, gidx, gn);
                }
            } else if (!FAST) {
                if (gn <= 1) entropy_sweep_m2<1>(b, xs, n, gthr, perm, racc, gidx, gn);
                else if (gn <= 2) entropy_sweep_m2<2>(b, xs, n, gthr, perm, racc, gidx, gn);
                else if (gn <= 4) entropy_sweep_m2<4>(b, xs, n, gthr, perm, racc, gidx, gn);
                else if (gn <= 6) entropy_sweep_m2<6>(b, xs, n, gthr, perm, racc, gidx, gn);
                else entropy_sweep_m2<8>(b, xs, n, gthr, perm, racc, gidx, gn);
            }
            TSFA_TICK(tk, b, 133 + (g0 > 0 ? 1 : 0));
        }
        }
        blk_sync();
        {
            int k = 0;
            for (int t = first; t < done; ++t) {
                const TsfaSpec sp = specs[t];
                const bool is_m2 = (sp.calc == TSFA_C_SAMPLE_ENTROPY) ||
                                   (sp.calc == TSFA_C_APPROXIMATE_ENTROPY && (int)sp.p[0] == 2);
                if (!is_m2) continue;
                EntAcc a;
                a.sum_log_m = racc[4 * k + 0];
                a.sum_log_m1 = racc[4 * k + 1];
                a.sum_cnt_m = racc[4 * k + 2];
                a.sum_cnt_m1 = racc[4 * k + 3];
                ++k;
                double v;
                if (sp.calc == TSFA_C_APPROXIMATE_ENTROPY) {
                    if (n <= 3) v = 0.0;  // N <= m + 1
                    else v = apen_from_acc(a, n, 2);
                } else {
                    if (n < 3) v = TSFA_NAN;  // no length-3 template: A = 0, B = 0 -> -log(0/0)
                    else v = sampen_from_acc(a, n, 2);
                }
                if (b.tid == 0) out_row[sp.col] = v;
            }
        }
        blk_sync();
    }
    // generic m
    for (int s = 0; !FAST && s < nspecs; ++s) {
        const TsfaSpec sp = specs[s];
        if (sp.calc != TSFA_C_APPROXIMATE_ENTROPY || (int)sp.p[0] == 2) continue;
        const int m = (int)sp.p[0];
        double v;
        if (n <= m + 1 || m < 1) {
            v = 0.0;
        } else {
            EntAcc a;
            entropy_sweep_generic(b, xs, n, m, ent_tolerance(sp.p[1] * sd), &a);
            v = apen_from_acc(a, n, m);
        }
        if (b.tid == 0) out_row[sp.col] = v;
    }
}

#include "fam_entropy_bits.h"
#include "fam_entropy_hbits.h"

#endif
