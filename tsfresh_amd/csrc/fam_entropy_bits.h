// Family ENTROPY, bit-matrix sweep (series of 3 .. TSFA_ENTB_MAXN samples, template length m = 2).
//
// sample_entropy (fc.py:1701) and approximate_entropy (fc.py:1759) need, for every template i, the number of
// templates j with max_t |x[i+t] - x[j+t]| <= r (t < m, and t < m + 1).  The pair sweeps of fam_entropy.h evaluate that
// Chebyshev distance per PAIR: three float64 subtractions, two maxima and two compares + two counter updates per
// tolerance.  This sweep never forms a distance:
//
//   * Per SAMPLE a and tolerance r, the samples within r of x[a] form one contiguous range of the sorted order,
//     [lo, hi) -- fl(|x_a - x_b|) <= r is monotone in x_b on either side of x_a (rounding is monotone), so both ends
//     are found by bisection with the reference's own float64 predicate on a sorted copy padded with +inf:
//     lo = #{p : !(x_a - x_p <= r)}, hi = #{p : x_p - x_a <= r}.  O(n log n) float64 operations per tolerance instead
//     of O(n^2); ties, infinities and NaN tolerances fall out of the same compares.
//   * With Pref(t) = the bit set {b : rank(x_b) < t} over the NATURAL sample index b, the row of the 0/1 matrix
//     A[a, b] = (|x_a - x_b| <= r) is Pref(hi) xor Pref(lo): two table reads and one xor per 32 pairs.
//   * Templates i, j of length 2 match iff A[i, j] & A[i+1, j+1]; of length 3 iff additionally A[i+2, j+2]: the
//     partners sit one step down the DIAGONAL.  Lane l of a strip (template i = strip start + l) therefore holds its row
//     rotated left by l bits, E_l[d] = A[i, d + l] -- one funnel shift per word -- and then
//         M2_l = E_l & E_{l+1},   M3_l = M2_l & M2_{l+1},   C_2[i] = popcount(M2_l),   C_3[i] = popcount(M3_l)
//     with the neighbour lane's word as a DPP operand of the AND itself: per 32 pairs one xor, one funnel shift, two
//     ANDs and two popcount-accumulates.  The counts are exact integers: the result is bit-identical to the pair
//     sweeps'.  (Columns are taken modulo 32 * NW with NW = ceil((n + 1) / 32): at least one zero guard column, so a
//     wrapped partner never matches.)
//
// Pref is (n + 1) x n bits = 128 KB at n = 1024, so it is built in column parts of TSFA_ENTB_QW words (+ one halo
// word for the rotation) by a workgroup prefix-OR over the sorted order.  A task = one tolerance x two half-strips of
// 30 templates: lanes 0-29 and 32-61 hold templates, lanes 30, 31, 62, 63 only supply their neighbours; the rotation
// is the lane index mod 32, so every lane reads the same 48 aligned bytes of its two entries (three ds_read_b128 each
// -- the table gathers of random entries are what the LDS spends its cycles on).  Ranges and counters of a wavefront's
// tasks stay in registers across the parts.
#ifndef TSFA_FAM_ENTROPY_BITS_H
#define TSFA_FAM_ENTROPY_BITS_H

#include "tsfa_entb_params.h"

#if TSFA_GPU
typedef __attribute__((address_space(3))) const unsigned int *entb_lds_cup;
TSFA_DEV unsigned int entb_lds_addr(const void *p) {
    return (unsigned int)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
// v & (v of lane + 1); lane 63: 0.  (v_and_b32_dpp wave_shl:1 -- the neighbour's word is an operand of the AND)
TSFA_DEV unsigned int entb_and_next(unsigned int v) {
    return v & (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
}
// inclusive OR-scan over the 64 lanes (the sequence LLVM's atomic optimizer emits for wave64 on gfx9)
TSFA_DEV unsigned int entb_wave_or_scan(unsigned int v) {
    v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
    v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x112 /* row_shr:2 */, 0xf, 0xf, false);
    v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x114 /* row_shr:4 */, 0xf, 0xf, false);
    v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x118 /* row_shr:8 */, 0xf, 0xf, false);
    v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
    v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);
    return v;
}
TSFA_DEV unsigned int entb_from_prev(unsigned int v) {  // value of lane - 1 (lane 0: 0)
    return (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
}
#endif

// The thread index made opaque to the optimiser: addresses and predicates derived from it are then computed inside the
// phase that uses them instead of being hoisted to the kernel's top, kept live across the sweep and spilled.
TSFA_DEV Blk entb_opaque(const Blk &b0) {
#if TSFA_GPU
    int t = b0.tid;
    asm volatile("" : "+v"(t));
    return Blk{t, b0.nt, b0.red, b0.np};
#else
    return b0;
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Ranges.  perm[0 .. n) = sample indices sorted by value (ties by index).  For tolerance k and sorted position q
// (sample a = perm[q]) the samples within thr[k] of x_a are the sorted positions [lo, hi).  Written per NATURAL sample
// index: rng[k * n + a] = (lo * S) | (hi * S) << 16 -- word offsets of the two table entries (lo == hi: empty).
// lo16: scratch, nk * n unsigned shorts.
// ---------------------------------------------------------------------------------------------------------------
template <int S_>
TSFA_DEV void entb_ranges(const Blk &b0, const double *xs, int n, const double *thr, int nk, const unsigned short *perm,
                          double *xsrt, unsigned int *rng) {
    const Blk b = entb_opaque(b0);
    const int K = TSFA_ENTB_MAXK;
    // xsrt: the sorted copy, padded with +inf to a power of two (the bisections then need no bound checks: a sample
    // beyond the order matches nothing).  The 2 K bisections of a sample run side by side as independent chains.
    const int P = next_pow2(n);
    for (int q = b.tid; q < P; q += b.nt) xsrt[q] = (q < n) ? xs[perm[q]] : TSFA_INF;
    double r[TSFA_ENTB_MAXK];
#pragma unroll
    for (int k = 0; k < K; ++k) r[k] = (k < nk) ? thr[k] : -1.0;
    blk_sync();
    const char *xb = (const char *)xsrt;
    for (int q = b.tid; q < n; q += b.nt) {
        const double xq = xsrt[q];
        // lo = #{p : !(x_q - x_p <= r)} (leading samples too far below), hi = #{p : x_p - x_q <= r}: both predicates are
        // monotone along the sorted order (rounding is monotone) and equal the reference's |x_q - x_p| <= r on their side
        int pl[TSFA_ENTB_MAXK], ph[TSFA_ENTB_MAXK];  // byte offsets into xsrt
#pragma unroll
        for (int k = 0; k < K; ++k) { pl[k] = 0; ph[k] = 0; }
        for (int step = (P >> 1) * 8; step >= 8; step >>= 1) {
            const char *at = xb + (step - 8);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const double vl = *(const double *)(at + pl[k]), vh = *(const double *)(at + ph[k]);
                pl[k] += (xq - vl <= r[k]) ? 0 : step;
                ph[k] += (vh - xq <= r[k]) ? step : 0;
            }
        }
        const int a = (int)perm[q];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            ph[k] += (*(const double *)(xb + ph[k]) - xq <= r[k]) ? 8 : 0;  // the count may reach P (P - 1 after the steps)
            // infinite sample / negative or NaN tolerance: matches nothing (not even itself)
            const bool any = (fabs(xq - xq) <= r[k]) && pl[k] < ph[k];
            const unsigned int lo = any ? (unsigned int)pl[k] : 0u, hi = any ? (unsigned int)ph[k] : 0u;  // 8 * rank
            if (k < nk) rng[k * n + a] = ((lo >> 3) * (unsigned int)S_) | (((hi >> 3) * (unsigned int)S_) << 16);
        }
    }
    blk_sync();
}

// ---------------------------------------------------------------------------------------------------------------
// Table of one column part: entry t (0 .. n), word u (0 .. S-1) = bits of the columns of word (w0 + u) mod NW whose
// rank is below t.  wtot: TSFA_ENTB_MAXWAVES * S words of cross-wavefront scratch.
// ---------------------------------------------------------------------------------------------------------------
template <int S_>
TSFA_DEV void entb_build_table(const Blk &b0, int n, const unsigned short *perm, int w0, int NW, unsigned int *table,
                               unsigned int *wtot) {
    const Blk b = entb_opaque(b0);
    const int S = S_;
    int target[S_];  // wave-uniform: the row word each entry word mirrors
#pragma unroll
    for (int u = 0; u < S; ++u) target[u] = (w0 + u) % NW;
#if TSFA_GPU
    const int E = (n + b.nt - 1) / b.nt;
    const int p0 = b.tid * E;
    unsigned int tot[S_];
#pragma unroll
    for (int u = 0; u < S; ++u) tot[u] = 0u;
    for (int e = 0; e < E; ++e) {
        const int p = p0 + e;
        if (p < n) {
            const int j = (int)perm[p];
            const int jw = j >> 5;
            const unsigned int bit = 1u << (j & 31);
#pragma unroll
            for (int u = 0; u < S; ++u) tot[u] |= (jw == target[u]) ? bit : 0u;
        }
    }
    unsigned int run[S_];
    const int lane = b.tid & 63, wave = b.tid >> 6;
#pragma unroll
    for (int u = 0; u < S; ++u) {
        const unsigned int inc = entb_wave_or_scan(tot[u]);
        if (lane == 63) wtot[wave * S + u] = inc;
        run[u] = entb_from_prev(inc);
    }
    blk_sync();
    for (int v = 0; v < wave; ++v) {
#pragma unroll
        for (int u = 0; u < S; ++u) run[u] |= wtot[v * S + u];
    }
    for (int e = 0; e < E; ++e) {
        const int p = p0 + e;
        if (p < n) {
#pragma unroll
            for (int u = 0; u < S; ++u) table[p * S + u] = run[u];
            const int j = (int)perm[p];
            const int jw = j >> 5;
            const unsigned int bit = 1u << (j & 31);
#pragma unroll
            for (int u = 0; u < S; ++u) run[u] |= (jw == target[u]) ? bit : 0u;
        }
    }
    if (b.tid == b.nt - 1) {  // entry n: every column (the last thread's running value is the total: its range ends the order)
#pragma unroll
        for (int u = 0; u < S; ++u) table[n * S + u] = run[u];
    }
#else
    (void)b; (void)wtot;
    unsigned int run[S_];
    for (int u = 0; u < S; ++u) run[u] = 0u;
    for (int p = 0; p <= n; ++p) {
        for (int u = 0; u < S; ++u) table[p * S + u] = run[u];
        if (p < n) {
            const int j = (int)perm[p];
            for (int u = 0; u < S; ++u)
                if ((j >> 5) == target[u]) run[u] |= 1u << (j & 31);
        }
    }
#endif
    blk_sync();
}

// One task (strip, tolerance) over one column part.  pl / ph: LDS byte addresses of the lane's two table entries, already
// advanced by the lane's word offset (lane >> 5); sh = lane & 31; nq = diagonal words of this part (<= QW).
#if TSFA_GPU
#define TSFA_ENTB_DPP " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
// FULL: all QW diagonal words of the part exist (every part of a series with NW % QW == 0)
template <int QW_, bool FULL>
TSFA_DEV unsigned int entb_task_part(unsigned int pl_addr, unsigned int ph_addr, unsigned int sh, int nq) {
    static_assert((QW_ + 1) % 4 == 0, "entries are read as 16-byte vectors");
    typedef unsigned int entb_u4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) const entb_u4 *entb_lds_cv4;
    const entb_lds_cv4 pl = (entb_lds_cv4)pl_addr, ph = (entb_lds_cv4)ph_addr;
    unsigned int A[QW_ + 1];
#pragma unroll
    for (int v = 0; v < (QW_ + 1) / 4; ++v) {
        const entb_u4 l4 = pl[v], h4 = ph[v];
        A[4 * v + 0] = h4.x ^ l4.x; A[4 * v + 1] = h4.y ^ l4.y; A[4 * v + 2] = h4.z ^ l4.z; A[4 * v + 3] = h4.w ^ l4.w;
    }
    unsigned int e[QW_], m[QW_];
#pragma unroll
    for (int t = 0; t < QW_; ++t) {
        e[t] = __builtin_amdgcn_alignbit(A[t + 1], A[t], sh);  // the row rotated left by the lane index
        if (!FULL) e[t] = (t < nq) ? e[t] : 0u;
    }
    // M2 = E & E(lane + 1), M3 = M2 & M2(lane + 1), the neighbour's word as the DPP operand of the AND; the words are
    // interleaved so that no DPP read follows the write of its source by less than the two required wait states
    unsigned int c2 = 0u, c3 = 0u;
    if constexpr (QW_ == 11) {
    asm("s_nop 1\n\t"
        "v_and_b32_dpp %2, %13, %13" TSFA_ENTB_DPP
        "v_and_b32_dpp %3, %14, %14" TSFA_ENTB_DPP
        "v_and_b32_dpp %4, %15, %15" TSFA_ENTB_DPP
        "v_and_b32_dpp %5, %16, %16" TSFA_ENTB_DPP
        "v_and_b32_dpp %6, %17, %17" TSFA_ENTB_DPP
        "v_and_b32_dpp %7, %18, %18" TSFA_ENTB_DPP
        "v_and_b32_dpp %8, %19, %19" TSFA_ENTB_DPP
        "v_and_b32_dpp %9, %20, %20" TSFA_ENTB_DPP
        "v_and_b32_dpp %10, %21, %21" TSFA_ENTB_DPP
        "v_and_b32_dpp %11, %22, %22" TSFA_ENTB_DPP
        "v_and_b32_dpp %12, %23, %23" TSFA_ENTB_DPP
        "v_bcnt_u32_b32 %0, %2, %0\n\t"
        "v_bcnt_u32_b32 %0, %3, %0\n\t"
        "v_bcnt_u32_b32 %0, %4, %0\n\t"
        "v_bcnt_u32_b32 %0, %5, %0\n\t"
        "v_bcnt_u32_b32 %0, %6, %0\n\t"
        "v_bcnt_u32_b32 %0, %7, %0\n\t"
        "v_bcnt_u32_b32 %0, %8, %0\n\t"
        "v_bcnt_u32_b32 %0, %9, %0\n\t"
        "v_bcnt_u32_b32 %0, %10, %0\n\t"
        "v_bcnt_u32_b32 %0, %11, %0\n\t"
        "v_bcnt_u32_b32 %0, %12, %0\n\t"
        "v_and_b32_dpp %2, %2, %2" TSFA_ENTB_DPP
        "v_and_b32_dpp %3, %3, %3" TSFA_ENTB_DPP
        "v_and_b32_dpp %4, %4, %4" TSFA_ENTB_DPP
        "v_and_b32_dpp %5, %5, %5" TSFA_ENTB_DPP
        "v_and_b32_dpp %6, %6, %6" TSFA_ENTB_DPP
        "v_and_b32_dpp %7, %7, %7" TSFA_ENTB_DPP
        "v_and_b32_dpp %8, %8, %8" TSFA_ENTB_DPP
        "v_and_b32_dpp %9, %9, %9" TSFA_ENTB_DPP
        "v_and_b32_dpp %10, %10, %10" TSFA_ENTB_DPP
        "v_and_b32_dpp %11, %11, %11" TSFA_ENTB_DPP
        "v_and_b32_dpp %12, %12, %12" TSFA_ENTB_DPP
        "v_bcnt_u32_b32 %1, %2, %1\n\t"
        "v_bcnt_u32_b32 %1, %3, %1\n\t"
        "v_bcnt_u32_b32 %1, %4, %1\n\t"
        "v_bcnt_u32_b32 %1, %5, %1\n\t"
        "v_bcnt_u32_b32 %1, %6, %1\n\t"
        "v_bcnt_u32_b32 %1, %7, %1\n\t"
        "v_bcnt_u32_b32 %1, %8, %1\n\t"
        "v_bcnt_u32_b32 %1, %9, %1\n\t"
        "v_bcnt_u32_b32 %1, %10, %1\n\t"
        "v_bcnt_u32_b32 %1, %11, %1\n\t"
        "v_bcnt_u32_b32 %1, %12, %1"
        : "+v"(c2), "+v"(c3), "=&v"(m[0]), "=&v"(m[1]), "=&v"(m[2]), "=&v"(m[3]), "=&v"(m[4]), "=&v"(m[5]), "=&v"(m[6]), "=&v"(m[7]), "=&v"(m[8]), "=&v"(m[9]), "=&v"(m[10])
        : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(e[4]), "v"(e[5]), "v"(e[6]), "v"(e[7]), "v"(e[8]), "v"(e[9]), "v"(e[10]));
    } else if constexpr (QW_ == 7) {
    asm("s_nop 1\n\t"
        "v_and_b32_dpp %2, %9, %9" TSFA_ENTB_DPP
        "v_and_b32_dpp %3, %10, %10" TSFA_ENTB_DPP
        "v_and_b32_dpp %4, %11, %11" TSFA_ENTB_DPP
        "v_and_b32_dpp %5, %12, %12" TSFA_ENTB_DPP
        "v_and_b32_dpp %6, %13, %13" TSFA_ENTB_DPP
        "v_and_b32_dpp %7, %14, %14" TSFA_ENTB_DPP
        "v_and_b32_dpp %8, %15, %15" TSFA_ENTB_DPP
        "v_bcnt_u32_b32 %0, %2, %0\n\t"
        "v_bcnt_u32_b32 %0, %3, %0\n\t"
        "v_bcnt_u32_b32 %0, %4, %0\n\t"
        "v_bcnt_u32_b32 %0, %5, %0\n\t"
        "v_bcnt_u32_b32 %0, %6, %0\n\t"
        "v_bcnt_u32_b32 %0, %7, %0\n\t"
        "v_bcnt_u32_b32 %0, %8, %0\n\t"
        "v_and_b32_dpp %2, %2, %2" TSFA_ENTB_DPP
        "v_and_b32_dpp %3, %3, %3" TSFA_ENTB_DPP
        "v_and_b32_dpp %4, %4, %4" TSFA_ENTB_DPP
        "v_and_b32_dpp %5, %5, %5" TSFA_ENTB_DPP
        "v_and_b32_dpp %6, %6, %6" TSFA_ENTB_DPP
        "v_and_b32_dpp %7, %7, %7" TSFA_ENTB_DPP
        "v_and_b32_dpp %8, %8, %8" TSFA_ENTB_DPP
        "v_bcnt_u32_b32 %1, %2, %1\n\t"
        "v_bcnt_u32_b32 %1, %3, %1\n\t"
        "v_bcnt_u32_b32 %1, %4, %1\n\t"
        "v_bcnt_u32_b32 %1, %5, %1\n\t"
        "v_bcnt_u32_b32 %1, %6, %1\n\t"
        "v_bcnt_u32_b32 %1, %7, %1\n\t"
        "v_bcnt_u32_b32 %1, %8, %1"
        : "+v"(c2), "+v"(c3), "=&v"(m[0]), "=&v"(m[1]), "=&v"(m[2]), "=&v"(m[3]), "=&v"(m[4]), "=&v"(m[5]), "=&v"(m[6])
        : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(e[4]), "v"(e[5]), "v"(e[6]));
    } else {
    static_assert(QW_ == 11 || QW_ == 7 || QW_ == 3, "write the DPP block for this part width");
    asm("s_nop 1\n\t"
        "v_and_b32_dpp %2, %5, %5" TSFA_ENTB_DPP
        "v_and_b32_dpp %3, %6, %6" TSFA_ENTB_DPP
        "v_and_b32_dpp %4, %7, %7" TSFA_ENTB_DPP
        "v_bcnt_u32_b32 %0, %2, %0\n\t"
        "v_bcnt_u32_b32 %0, %3, %0\n\t"
        "v_bcnt_u32_b32 %0, %4, %0\n\t"
        "v_and_b32_dpp %2, %2, %2" TSFA_ENTB_DPP
        "v_and_b32_dpp %3, %3, %3" TSFA_ENTB_DPP
        "v_and_b32_dpp %4, %4, %4" TSFA_ENTB_DPP
        "v_bcnt_u32_b32 %1, %2, %1\n\t"
        "v_bcnt_u32_b32 %1, %3, %1\n\t"
        "v_bcnt_u32_b32 %1, %4, %1"
        : "+v"(c2), "+v"(c3), "=&v"(m[0]), "=&v"(m[1]), "=&v"(m[2])
        : "v"(e[0]), "v"(e[1]), "v"(e[2]));
    }
    return c2 | (c3 << 16);
}
#endif

TSFA_DEV int entb_wave_sum_i32(int v) {
#if TSFA_GPU
    v += __builtin_amdgcn_update_dpp(0, v, TSFA_DPP_QUAD_XOR1, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, TSFA_DPP_QUAD_XOR2, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, TSFA_DPP_ROW_HALF_MIRROR, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, TSFA_DPP_ROW_MIRROR, 0xf, 0xf, false);
    return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) +
           (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
#else
    return v;
#endif
}

// Reduction of the per-thread products / sums of one round (see entropy_bits_batch): part = LDS scratch of
// 2 K * (nt / 16) doubles + 2 K * (nt / 64) words.
TSFA_DEV void entb_totals(const Blk &b, int kn, int k0, double *pm, double *pm1, int *sc, int *sc1, int *nm, int *nm1,
                          int nrow_m, int nrow_m1, double *part, double *racc, bool accumulate) {
    const int K = TSFA_ENTB_MAXK;
#if !TSFA_GPU
    const double ldm = log((double)nrow_m), ldm1 = log((double)nrow_m1);
#endif
#if TSFA_GPU
    const int lane = b.tid & 63, wave = b.tid >> 6, nrows = b.nt >> 4;  // DPP rows of 16 lanes
    // products over the 16 lanes of a row (xor butterflies inside the row), sums over the wavefront
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (k < kn) {
            pm[k] *= dpp_mov_f64<TSFA_DPP_QUAD_XOR1>(pm[k]);   pm1[k] *= dpp_mov_f64<TSFA_DPP_QUAD_XOR1>(pm1[k]);
            pm[k] *= dpp_mov_f64<TSFA_DPP_QUAD_XOR2>(pm[k]);   pm1[k] *= dpp_mov_f64<TSFA_DPP_QUAD_XOR2>(pm1[k]);
            pm[k] *= dpp_mov_f64<TSFA_DPP_ROW_HALF_MIRROR>(pm[k]); pm1[k] *= dpp_mov_f64<TSFA_DPP_ROW_HALF_MIRROR>(pm1[k]);
            pm[k] *= dpp_mov_f64<TSFA_DPP_ROW_MIRROR>(pm[k]);  pm1[k] *= dpp_mov_f64<TSFA_DPP_ROW_MIRROR>(pm1[k]);
            if ((lane & 15) == 0) {
                part[(2 * k) * nrows + (b.tid >> 4)] = pm[k];
                part[(2 * k + 1) * nrows + (b.tid >> 4)] = pm1[k];
            }
        }
    }
    // the integer totals of a tolerance packed into one word each (sum C < 2^21, #rows < 2^11): wavefront sums by DPP,
    // one slot per wavefront
    unsigned int *islot = (unsigned int *)(void *)(part + 2 * K * nrows);  // [nwaves][2 K]
    const int nw = b.nt >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (k < kn) {
            const int s0 = entb_wave_sum_i32((int)((unsigned int)sc[k] | ((unsigned int)nm[k] << 21)));
            const int s1 = entb_wave_sum_i32((int)((unsigned int)sc1[k] | ((unsigned int)nm1[k] << 21)));
            if (lane == 0) { islot[wave * 2 * K + 2 * k] = (unsigned int)s0; islot[wave * 2 * K + 2 * k + 1] = (unsigned int)s1; }
        }
    }
    blk_sync();
    // one logarithm per partial product -- by as many lanes as there are products (a float64 log is ~150 instructions) -- and,
    // in the same instruction stream, log N of the two template counts by the two lanes behind them (round 6: they used to
    // be evaluated by the closing lanes, 300 instructions on the one wavefront everybody waits for)
    double *lslot = (double *)(void *)(islot + (((size_t)nw * 2 * K + 1) & ~(size_t)1));
    {
        const int nlog = 2 * kn * nrows;
        if (b.tid < nlog + 2) {
            const double arg = (b.tid < nlog) ? part[b.tid] : (double)((b.tid == nlog) ? nrow_m : nrow_m1);
            const double lg = log(arg);
            if (b.tid < nlog) part[b.tid] = lg; else lslot[b.tid - nlog] = lg;
        }
    }
    blk_sync();
    for (int j0 = 0; j0 < 2 * kn; j0 += nrows) {  // value j = 2 k + (0: m, 1: m + 1) on the 16 lanes of a DPP row (nrows rows)
        // (round 6: the row's lanes add the partial logarithms -- one lane used to walk all nt / 16 of them, a chain of
        // dependent LDS reads at the tail of every series)
        const int j = j0 + (b.tid >> 4), l16 = b.tid & 15;
        if (j >= 2 * kn) continue;   // (whole rows: the DPP partners of a live lane are live)
        const int k = j >> 1, odd = j & 1;
        double a = 0.0;
        for (int r = l16; r < nrows; r += 16) a += part[j * nrows + r];
        a += dpp_mov_f64<TSFA_DPP_QUAD_XOR1>(a);
        a += dpp_mov_f64<TSFA_DPP_QUAD_XOR2>(a);
        a += dpp_mov_f64<TSFA_DPP_ROW_HALF_MIRROR>(a);
        a += dpp_mov_f64<TSFA_DPP_ROW_MIRROR>(a);
        if (l16 == 0) {
            const double ldm = lslot[0], ldm1 = lslot[1];
            unsigned int tot_c = 0u, tot_n = 0u;   // a wavefront's packed word holds sum C < 2^21 and #rows < 2^11; the block's may not
            for (int w = 0; w < nw; ++w) { const unsigned int t = islot[w * 2 * K + j]; tot_c += t & 0x1FFFFFu; tot_n += t >> 21; }
            double *d = racc + 4 * (k0 + k);
            const double v = a - (double)tot_n * (odd ? ldm1 : ldm), c = (double)tot_c;
            if (accumulate) { d[odd] += v; d[2 + odd] += c; }
            else { d[odd] = v; d[2 + odd] = c; }
        }
    }
    blk_sync();
#else
    (void)b; (void)part;
    for (int k = 0; k < kn && k < K; ++k) {
        double *d = racc + 4 * (k0 + k);
        if (!accumulate) { d[0] = 0.0; d[1] = 0.0; d[2] = 0.0; d[3] = 0.0; }
        d[0] += log(pm[k]) - (double)nm[k] * ldm;
        d[1] += log(pm1[k]) - (double)nm1[k] * ldm1;
        d[2] += (double)sc[k];
        d[3] += (double)sc1[k];
    }
#endif
}

// Totals of one round's kn tolerances from the counters cnt[k * n + i] = C_2 | C_3 << 16 (see entropy_bits_batch): products of
// the counts per thread and DPP row, one logarithm per row of lanes.  part: LDS behind the counters.
TSFA_DEV void entb_round_totals(const Blk &b_in, const unsigned int *cnt, double *part, int n, int kn, int k0, int nrow_m,
                                int nrow_m1, double *racc) {
    const Blk b = entb_opaque(b_in);
    const int K = TSFA_ENTB_MAXK;
    double pm[TSFA_ENTB_MAXK], pm1[TSFA_ENTB_MAXK];
    int sc[TSFA_ENTB_MAXK], sc1[TSFA_ENTB_MAXK], nm[TSFA_ENTB_MAXK], nm1[TSFA_ENTB_MAXK];
    // (a thread multiplies at most four counts before the lanes combine theirs: longer rows go in chunks)
    for (int c0 = 0; c0 < nrow_m; c0 += 4 * b.nt) {
#pragma unroll
        for (int k = 0; k < K; ++k) { pm[k] = 1.0; pm1[k] = 1.0; sc[k] = 0; sc1[k] = 0; nm[k] = 0; nm1[k] = 0; }
        for (int ib = c0; ib < nrow_m && ib < c0 + 4 * b.nt; ib += b.nt) {
            const int i = ib + b.tid;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (k < kn && i < nrow_m) {
                    const unsigned int cc = cnt[k * n + i];
                    const int t2 = (int)(cc & 0xFFFFu), t3 = (int)(cc >> 16);
                    sc[k] += t2;
                    if (t2 != nrow_m) { pm[k] *= (double)t2; ++nm[k]; }
                    if (i < nrow_m1) {
                        sc1[k] += t3;
                        if (t3 != nrow_m1) { pm1[k] *= (double)t3; ++nm1[k]; }
                    }
                }
            }
        }
        entb_totals(b, kn, k0, pm, pm1, sc, sc1, nm, nm1, nrow_m, nrow_m1, part, racc, c0 > 0);
    }
}

#if TSFA_GPU
// Series of 2049 .. 4096 samples (16-byte table entries, TSFA_ENTB_QW_LONG): the tasks of a round of three tolerances fill
// the 14 register slots of a wavefront, so six tolerances used to take two rounds -- and every round rebuilds all 43 column
// parts (a table build and three barriers each; one workgroup of sixteen wavefronts per CU: the barriers are what the
// kernel waits for).  Here the ranges of TWO rounds are found one after the other (the work region holds one round's) and kept
// in registers together -- a range as ONE packed word (two 16-bit word offsets, unpacked per part: three instructions
// beside a sweep of ~25) -- and the parts are built and swept once for both.  Counters and totals again round by round.
template <int QW_>
TSFA_DEV void entropy_bits_batch_paired(const Blk &b, const double *xs, int n, const double *thr, int nk,
                                        const unsigned short *perm, unsigned int *work, double *racc, int kround) {
    constexpr int S = QW_ + 1, QW = QW_, MT = TSFA_ENTB_MAXT;
    const int nrow_m = n - 1, nrow_m1 = n - 2;
    const int NW = (n + 32) >> 5;
    const int nparts = (NW + QW - 1) / QW;
    const int nstrips = ((nrow_m + TSFA_ENTB_STRIP - 1) / TSFA_ENTB_STRIP + 1) / 2;
    unsigned int *rng = work + 2 * (size_t)next_pow2(n);
    unsigned int *table = work;
    unsigned int *wtot = work + (size_t)(n + 1) * S;
    unsigned int *cnt = work;
    const int kcap = kround;
    const int lane = b.tid & 63, wave = __builtin_amdgcn_readfirstlane(b.tid >> 6), nw = b.nt >> 6;
    const unsigned int tbase = entb_lds_addr(table), sh = (unsigned int)(lane & 31);
    const int lane_row = (lane >> 5) * TSFA_ENTB_STRIP + (lane & 31);
    for (int k0 = 0; k0 < nk; k0 += 2 * kround) {
        unsigned int rr[2 * MT], ct[2 * MT];
        int kn_g[2], ntask_g[2];
        unsigned int kmagic_g[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int kb = k0 + g * kround;
            int kn = nk - kb;
            kn = (kn > kround) ? kround : kn;
            kn = (kn < 0) ? 0 : kn;
            kn_g[g] = kn;
            ntask_g[g] = nstrips * kn;
            kmagic_g[g] = 65536u / (unsigned int)(kn > 0 ? kn : 1) + 1u;
            if (kn > 0) {
                blk_sync();   // the work region held the other round's ranges / the previous pair's counters
                entb_ranges<QW_ + 1>(b, xs, n, thr + kb, kn, perm, (double *)(void *)work, rng);
            }
#pragma unroll
            for (int tt = 0; tt < MT; ++tt) {
                const int id = wave + tt * nw;
                unsigned int r = 0u;
                if (id < ntask_g[g]) {
                    const int s = (int)(((unsigned int)id * kmagic_g[g]) >> 16), k = id - s * kn;
                    const int i = s * (2 * TSFA_ENTB_STRIP) + lane_row;
                    r = (i < n) ? rng[k * n + i] : 0u;
                }
                rr[g * MT + tt] = r;
                ct[g * MT + tt] = 0u;
            }
        }
        blk_sync();
        for (int part = 0; part < nparts; ++part) {
            entb_build_table<QW_ + 1>(b, n, perm, part * QW, NW, table, wtot);
            const int nq = (NW - part * QW < QW) ? (NW - part * QW) : QW;
            if (nq == QW) {
#pragma unroll
                for (int t2 = 0; t2 < 2 * MT; ++t2) {
                    if (wave + (t2 % MT) * nw < ntask_g[t2 / MT]) {
                        const unsigned int r = rr[t2];
                        ct[t2] += entb_task_part<QW_, true>(tbase + 4u * (r & 0xFFFFu), tbase + 4u * (r >> 16), sh, nq);
                    }
                }
            } else {
#pragma unroll
                for (int t2 = 0; t2 < 2 * MT; ++t2) {
                    if (wave + (t2 % MT) * nw < ntask_g[t2 / MT]) {
                        const unsigned int r = rr[t2];
                        ct[t2] += entb_task_part<QW_, false>(tbase + 4u * (r & 0xFFFFu), tbase + 4u * (r >> 16), sh, nq);
                    }
                }
            }
            blk_sync();
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            if (kn_g[g] > 0) {
                const int kn = kn_g[g];
#pragma unroll
                for (int tt = 0; tt < MT; ++tt) {
                    const int id = wave + tt * nw;
                    if (id < ntask_g[g]) {
                        const int s = (int)(((unsigned int)id * kmagic_g[g]) >> 16), k = id - s * kn;
                        const int i = s * (2 * TSFA_ENTB_STRIP) + lane_row;
                        if ((lane & 31) < TSFA_ENTB_STRIP && i < nrow_m) cnt[k * n + i] = ct[g * MT + tt];
                    }
                }
                blk_sync();
                entb_round_totals(b, cnt, (double *)(void *)(cnt + (((size_t)kcap * n + 1) & ~(size_t)1)), n, kn, k0 + g * kround,
                                  nrow_m, nrow_m1, racc);
                blk_sync();
            }
        }
    }
}
#endif

// ---------------------------------------------------------------------------------------------------------------
// One batch of nk <= TSFA_ENTB_MAXK tolerances (thr[0 .. nk), any order): racc[4 k .. 4 k + 3] = sum log(C_2 / (n-1)),
// sum log(C_3 / (n-2)), sum C_2, sum C_3 of tolerance k -- the totals the pair sweeps deliver.
// perm: all n samples sorted; work: entb_work_words(n) words of LDS.
// ---------------------------------------------------------------------------------------------------------------
template <int QW_>
TSFA_DEV void entropy_bits_batch(const Blk &b_in, const double *xs, int n, const double *thr, int nk,
                                 const unsigned short *perm, unsigned int *work, double *racc, int kcap_max = TSFA_ENTB_MAXK) {
    const Blk b = entb_opaque(b_in);
    const int S = QW_ + 1, QW = QW_;
    const int nrow_m = n - 1, nrow_m1 = n - 2;
    const int NW = (n + 32) >> 5;  // row words: at least one zero guard column
    const int nparts = (NW + QW - 1) / QW;
    const int nstrips = ((nrow_m + TSFA_ENTB_STRIP - 1) / TSFA_ENTB_STRIP + 1) / 2;  // pairs of half-strips
    unsigned int *rng = work + 2 * (size_t)next_pow2(n);  // behind the sorted copy
    unsigned int *table = work;
    unsigned int *wtot = work + (size_t)(n + 1) * S;
    unsigned int *cnt = work;
    TSFA_TICKER(tk, 0);
    blk_sync();
#if TSFA_GPU
    const int lane = b.tid & 63, wave = __builtin_amdgcn_readfirstlane(b.tid >> 6), nw = b.nt >> 6;
    const unsigned int tbase = entb_lds_addr(table);
    // tolerances per round: all of them when the tasks fit the wavefronts' registers
    int kround = (TSFA_ENTB_MAXT * nw) / nstrips;
    if (kround > nk) kround = nk;
    // ... and no more than the work region was sized for: the layout takes entb_kround of the LONGEST series of the
    // launch (5 at 2500 samples), a shorter series of the same launch could fit 6 in its registers -- and wrote its sixth
    // row of ranges past the region (found by the fuzz on a ragged batch: NaN for the last tolerance)
    if (kround > kcap_max) kround = kcap_max;
    if (kround < 1) kround = 1;  // (the host never selects this sweep for such a shape; the counts below stay correct
                                 //  only for nstrips <= MAXT * nw)
    const int kcap = kround;     // tolerances per round: the ranges / counters of ONE round live in the work region
    if (QW_ != TSFA_ENTB_QW && nk > kround) {   // two rounds' tasks in registers, the column parts built once for both
        entropy_bits_batch_paired<QW_>(b, xs, n, thr, nk, perm, work, racc, kround);
        return;
    }
    for (int k0 = 0; k0 < nk; k0 += kround) {
        const int kn = (nk - k0 < kround) ? (nk - k0) : kround;
        const int ntask = nstrips * kn;
        unsigned int rl[TSFA_ENTB_MAXT], rh[TSFA_ENTB_MAXT], ct[TSFA_ENTB_MAXT];
        const unsigned int lane_off = tbase, sh = (unsigned int)(lane & 31);
        const int lane_row = (lane >> 5) * TSFA_ENTB_STRIP + (lane & 31);  // template of the lane within its pair of half-strips
        const unsigned int kmagic = 65536u / (unsigned int)kn + 1u;  // id / kn == (id * kmagic) >> 16 for id < 10 000
        blk_sync();   // the work region held the previous round's counters
        entb_ranges<QW_ + 1>(b, xs, n, thr + k0, kn, perm, (double *)(void *)work, rng);
        TSFA_TICK(tk, b, 132);
#pragma unroll
        for (int tt = 0; tt < TSFA_ENTB_MAXT; ++tt) {
            const int id = wave + tt * nw;
            rl[tt] = lane_off;
            rh[tt] = lane_off;
            ct[tt] = 0u;
            if (id < ntask) {
                const int s = (int)(((unsigned int)id * kmagic) >> 16), k = id - s * kn;   // tolerance within the round
                const int i = s * (2 * TSFA_ENTB_STRIP) + lane_row;
                const unsigned int r = (i < n) ? rng[k * n + i] : 0u;
                rl[tt] = lane_off + 4u * (r & 0xFFFFu);
                rh[tt] = lane_off + 4u * (r >> 16);
            }
        }
        blk_sync();
        TSFA_TICK(tk, b, 138);
        for (int part = 0; part < nparts; ++part) {
            entb_build_table<QW_ + 1>(b, n, perm, part * QW, NW, table, wtot);
            TSFA_TICK(tk, b, 139);
            const int nq = (NW - part * QW < QW) ? (NW - part * QW) : QW;
            if (nq == QW) {
#pragma unroll
                for (int tt = 0; tt < TSFA_ENTB_MAXT; ++tt) {
                    if (wave + tt * nw < ntask) ct[tt] += entb_task_part<QW_, true>(rl[tt], rh[tt], sh, nq);
                }
            } else {
#pragma unroll
                for (int tt = 0; tt < TSFA_ENTB_MAXT; ++tt) {
                    if (wave + tt * nw < ntask) ct[tt] += entb_task_part<QW_, false>(rl[tt], rh[tt], sh, nq);
                }
            }
            TSFA_TICK(tk, b, 136);
            blk_sync();
            TSFA_TICK(tk, b, 137);
        }
        // counts to LDS: cnt[k * n + i] = C_2 | C_3 << 16
#pragma unroll
        for (int tt = 0; tt < TSFA_ENTB_MAXT; ++tt) {
            const int id = wave + tt * nw;
            if (id < ntask) {
                const int s = (int)(((unsigned int)id * kmagic) >> 16), k = id - s * kn;
                const int i = s * (2 * TSFA_ENTB_STRIP) + lane_row;
                if ((lane & 31) < TSFA_ENTB_STRIP && i < nrow_m) cnt[k * n + i] = ct[tt];
            }
        }
        blk_sync();
        TSFA_TICK(tk, b, 133);
#else
    // the emulation runs the rounds a 16-wavefront workgroup would (entb_kround): ranges and counters of ONE round at a time
    int kround = entb_kround(n, nk, TSFA_ENTB_MAXWAVES);
    if (kround > kcap_max) kround = kcap_max;
    const int kcap = kround;
    for (int k0 = 0; k0 < nk; k0 += kround) {
        const int kn = (nk - k0 < kround) ? (nk - k0) : kround;
        entb_ranges<QW_ + 1>(b, xs, n, thr + k0, kn, perm, (double *)(void *)work, rng);
        // single-thread emulation of the same strips: 64 "lanes", lanes 62 / 63 only supply their neighbours
        const int RS = TSFA_ENTB_MAXN_LONG + 64;
        static thread_local unsigned int rg[TSFA_ENTB_MAXK * (TSFA_ENTB_MAXN_LONG + 64)], ct[TSFA_ENTB_MAXK * (TSFA_ENTB_MAXN_LONG + 64)];
        for (int k = 0; k < kn; ++k)
            for (int i = 0; i < n + 64; ++i) { rg[k * RS + i] = (i < n) ? rng[k * n + i] : 0u; ct[k * RS + i] = 0u; }
        for (int part = 0; part < nparts; ++part) {
            entb_build_table<QW_ + 1>(b, n, perm, part * QW, NW, table, wtot);
            const int nq = (NW - part * QW < QW) ? (NW - part * QW) : QW;
            for (int k = 0; k < kn; ++k) {
                for (int s = 0; s < nstrips; ++s) {
                    unsigned int e[66][QW_], m2[66][QW_];
                    for (int l = 0; l < 64; ++l) {
                        const int row = s * (2 * TSFA_ENTB_STRIP) + (l >> 5) * TSFA_ENTB_STRIP + (l & 31);
                        const unsigned int r = rg[k * RS + row];
                        const unsigned int *pl = table + (r & 0xFFFFu), *ph = table + (r >> 16);
                        const unsigned int sh = (unsigned int)(l & 31);
                        for (int t = 0; t < QW; ++t) {
                            const unsigned long long two = ((unsigned long long)(ph[t + 1] ^ pl[t + 1]) << 32) | (ph[t] ^ pl[t]);
                            e[l][t] = (unsigned int)(two >> sh);
                        }
                    }
                    for (int t = 0; t < QW; ++t) { e[64][t] = 0u; m2[64][t] = m2[65][t] = 0u; }
                    for (int l = 0; l < 64; ++l)
                        for (int t = 0; t < QW; ++t) m2[l][t] = e[l][t] & e[l + 1][t];
                    for (int l = 0; l < 64; ++l) {
                        if ((l & 31) >= TSFA_ENTB_STRIP) continue;
                        const int row = s * (2 * TSFA_ENTB_STRIP) + (l >> 5) * TSFA_ENTB_STRIP + (l & 31);
                        unsigned int c2 = 0u, c3 = 0u;
                        for (int t = 0; t < nq; ++t) {
                            c2 += (unsigned int)__builtin_popcount(m2[l][t]);
                            c3 += (unsigned int)__builtin_popcount(m2[l][t] & m2[l + 1][t]);
                        }
                        ct[k * RS + row] += c2 | (c3 << 16);
                    }
                }
            }
        }
        for (int k = 0; k < kn; ++k)
            for (int i = 0; i < nrow_m; ++i) cnt[k * n + i] = ct[k * RS + i];
#endif
        // ---- totals of the round's tolerances: sum_i log(C_i / N) = log(prod_i C_i) - (#rows) log N.  The counts are
        //      integers <= 2^11, so a thread multiplies its rows' counts, the 16 lanes of a DPP row multiply theirs
        //      (< 2^1024 up to 92 factors), and ONE logarithm is taken per row of lanes -- in a second step, by as many
        //      lanes as there are partial products (a float64 log is ~150 instructions: per thread it would cost more
        //      than the sweep of a column part).  Rows with C_i == N contribute exactly 0, as in the reference.
        entb_round_totals(b_in, cnt, (double *)(void *)(cnt + (((size_t)kcap * n + 1) & ~(size_t)1)), n, kn, k0, nrow_m, nrow_m1, racc);
        blk_sync();
    }
    TSFA_TICK(tk, b, 134);
}

#if TSFA_GPU
// ---------------------------------------------------------------------------------------------------------------
// Sort of the n samples (float32 values: one packed 48-bit key = ordered value bits | index), E = np2 / nt keys per
// thread.  Every wavefront sorts its 64 E keys with the bitonic network in registers / across lanes (no barrier); the
// sorted runs are then merged pairwise by RANKING: a key's position in the merged run = its offset in its own run +
// the number of smaller keys in the sibling run (bisection in LDS; "smaller or equal" for the right-hand run, so equal
// padding keys keep distinct places).  log2(nt / 64) barriers instead of two per cross-wavefront bitonic stage, and a
// merge level costs log2(run) dependent LDS reads instead of log2(run) + 1 compare-exchange stages.
// buf: 2 * np2 64-bit words of LDS.  Result: perm[0 .. np2) = indices in sorted order (0xFFFF beyond n).
// ---------------------------------------------------------------------------------------------------------------
template <int E>
TSFA_DEVN void entb_sort_merge(const Blk b, const double *xs, int n, unsigned short *perm, unsigned long long *buf) {
    const int np2 = E * b.nt;
    const int g0 = b.tid * E;
    unsigned long long pk[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int g = g0 + e;
        pk[e] = (g < n) ? sort_pack_f32((float)xs[g], g) : sort_pack_f32((float)TSFA_INF, 0xFFFF);
    }
    const int W = 64 * E;  // keys per wavefront
    for (int k = 2; k <= W; k <<= 1) {
        const int kdir = (k == W) ? (1 << 30) : k;  // the last in-wavefront merge sorts every run ascending
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j < E) {
                switch (j) {
                case 1: if (E > 1) sortp_stage_regs<E, (E > 1 ? 1 : 0)>(pk, g0, kdir); break;
                case 2: if (E > 2) sortp_stage_regs<E, (E > 2 ? 2 : 0)>(pk, g0, kdir); break;
                default: break;
                }
            } else {
                switch (j / E) {
                case 1: sortp_stage_lanes<E, 1>(pk, g0, kdir, j); break;
                case 2: sortp_stage_lanes<E, 2>(pk, g0, kdir, j); break;
                case 4: sortp_stage_lanes<E, 4>(pk, g0, kdir, j); break;
                case 8: sortp_stage_lanes<E, 8>(pk, g0, kdir, j); break;
                case 16: sortp_stage_lanes<E, 16>(pk, g0, kdir, j); break;
                default: sortp_stage_lanes<E, 32>(pk, g0, kdir, j); break;
                }
            }
        }
    }
    unsigned long long *src = buf, *dst = buf + np2;
    if (np2 > W) {
        blk_sync();
#pragma unroll
        for (int e = 0; e < E; ++e) src[g0 + e] = pk[e];
        blk_sync();
        int lr = 6;
        while ((1 << lr) < W) ++lr;
        for (int R = W; R < np2; R <<= 1, ++lr) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int pos = g0 + e;
                const int run = pos >> lr, o = pos & (R - 1);
                const unsigned long long *sib = src + ((run ^ 1) << lr);
                const unsigned long long key = pk[e] + (unsigned long long)(run & 1);  // right-hand run: count <=
                int cnt = 0;
                for (int step = R >> 1; step >= 1; step >>= 1) cnt += (sib[cnt + step - 1] < key) ? step : 0;
                cnt += (sib[cnt] < key) ? 1 : 0;
                dst[((run >> 1) << (lr + 1)) + o + cnt] = pk[e];
            }
            blk_sync();
#pragma unroll
            for (int e = 0; e < E; ++e) pk[e] = dst[g0 + e];
            unsigned long long *t = src; src = dst; dst = t;
        }
    }
    blk_sync();
#pragma unroll
    for (int e = 0; e < E; ++e) perm[g0 + e] = (unsigned short)(pk[e] & 0xFFFFull);
    blk_sync();
}
#endif

// The ENTROPY specs of one series by the bit-matrix sweep (every spec has m = 2; 3 <= n <= TSFA_ENTB_MAXN is decided
// on the host, shorter series take the closed forms below).  xs: n + 4 doubles; thr: >= 56 doubles; perm:
// next_pow2(n) + 32 entries; work: entb_work_words(maxn) words (may alias b.np: the numpy-order sums finish first).
template <bool F32, int QW_ = TSFA_ENTB_QW>
TSFA_DEV void fam_entropy_series_bits(const Blk &b, double *xs, int n, const TsfaSpec *specs, int nspecs, double *out_row,
                                      double *thr, unsigned short *perm, unsigned int *work,
                                      unsigned short *perm_out = nullptr, int kcap_max = TSFA_ENTB_MAXK,
                                      const double *stats = nullptr) {
    TSFA_TICKER(tk, 0);
    const double dn = (double)n;
    // (stats: the record k_basic left for this series -- the same numpy-order mean and variance, TSFA_STATS_*)
    const double mean = stats ? stats[TSFA_STATS_MEAN] : np_sum(b, n, [=](int i) { return xs[i]; }) / dn;
    const double var = stats ? stats[TSFA_STATS_VAR] : np_sum(b, n, [=](int i) { const double d = xs[i] - mean; return d * d; }) / dn;
    const double sd = sqrt(var);
    blk_sync();
    TSFA_TICK(tk, b, 130);
    if (n >= 3) {
        const int np2 = next_pow2(n);
        bool sorted = false;
#if TSFA_GPU
        if (!sorted && F32 && b.nt >= 64) {  // wavefront-local bitonic sort + merge by ranking (work: 2 * np2 64-bit words)
            unsigned long long *buf = (unsigned long long *)(void *)work;
            if (np2 == b.nt) { entb_sort_merge<1>(b, xs, n, perm, buf); sorted = true; }
            else if (np2 == 2 * b.nt) { entb_sort_merge<2>(b, xs, n, perm, buf); sorted = true; }
            else if (np2 == 4 * b.nt) { entb_sort_merge<4>(b, xs, n, perm, buf); sorted = true; }
        }
#endif
        if (!sorted) entropy_sort_templates(b, xs, n + 1, perm, np2, F32);  // all n samples (a "template" per sample)
        if (perm_out != nullptr) {  // the sample order, for the SORT family of the same plan (fam_sort.h: perm_in)
            for (int i = b.tid; i < n; i += b.nt) perm_out[i] = perm[i];
        }
        TSFA_TICK(tk, b, 131);
    }
    double *racc = thr + TSFA_ENTB_MAXK;
    for (int first = 0; first < nspecs; first += TSFA_ENTB_MAXK) {
        const int nk = (nspecs - first < TSFA_ENTB_MAXK) ? (nspecs - first) : TSFA_ENTB_MAXK;
        blk_sync();
        for (int k = b.tid; k < nk; k += b.nt) {
            const TsfaSpec sp = specs[first + k];
            thr[k] = ent_tolerance((sp.calc == TSFA_C_SAMPLE_ENTROPY) ? 0.2 * sd : sp.p[1] * sd);
        }
        blk_sync();
        if (n >= 3) entropy_bits_batch<QW_>(b, xs, n, thr, nk, perm, work, racc, kcap_max);
        // lane = column: the closing divisions and sample_entropy's logarithm once per column, not once per column AND
        // wavefront (every thread used to evaluate all six tails for thread 0 to store: 3.8 k wave-instructions per series)
        for (int k = b.tid; k < nk; k += b.nt) {
            const TsfaSpec sp = specs[first + k];
            EntAcc a;
            a.sum_log_m = racc[4 * k + 0];
            a.sum_log_m1 = racc[4 * k + 1];
            a.sum_cnt_m = racc[4 * k + 2];
            a.sum_cnt_m1 = racc[4 * k + 3];
            double v;
            if (sp.calc == TSFA_C_APPROXIMATE_ENTROPY) v = (n <= 3) ? 0.0 : apen_from_acc(a, n, 2);
            else v = (n < 3) ? TSFA_NAN : sampen_from_acc(a, n, 2);
            out_row[sp.col] = v;
        }
        blk_sync();
    }
}

#endif
