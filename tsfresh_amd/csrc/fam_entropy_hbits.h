// Family ENTROPY, bit-matrix sweep for series BEYOND a CU's LDS (TSFA_ENTB_MAXN_LONG < n <= TSFA_ENTH_MAXN samples, template
// length m = 2): the sweep of fam_entropy_bits.h with
//   * the per-sample arrays -- the float64 copy of the series, its sorted copy, the sample order, the ranges and the counters of
//     ALL tolerances of a batch -- in the workgroup's slot of HBM scratch (coalesced, streamed; the long-series build's barrier
//     orders global memory),
//   * the table of ONE diagonal word per column part (8-byte entries: the word and its halo) in LDS: (n + 1) x 8 bytes, 128 KB at
//     16 384 samples -- the random gathers of two entries per lane and task are what has to stay in LDS,
//   * the (strip, tolerance) tasks in BATCHES of as many as the sixteen wavefronts hold in registers: a series of 16 384 samples
//     has 274 strips x 6 tolerances = 1 644 tasks against 16 x TSFA_ENTH_MAXT register slots; every batch walks all column
//     parts (the table of a part is rebuilt per batch: O(n) against the batch's O(tasks) sweep of it -- about as much again).
// Replaces the O(n^2) float64 pair sweep of fam_entropy.h there: 2 000 series x 16 384 samples, ComprehensiveFCParameters,
// 0.90 s of a 0.98 s step (profiles/r04_long_entropy.md).  The counts are the same integers.
// Compiled into the long-series build only (tsfa_kernels_long.hip: TSFA_LONG).
#ifndef TSFA_FAM_ENTROPY_HBITS_H
#define TSFA_FAM_ENTROPY_HBITS_H

#if TSFA_GPU && defined(TSFA_LONG)

#if !defined(TSFA_ENTH_MAXT)
#define TSFA_ENTH_MAXT 16   // (20 / 24: the kernel's 128 registers no longer hold the tasks, their addresses are reloaded from scratch in the sweep)
#endif
//                // tasks per wavefront: three registers each (two entry addresses, the packed counts)

// TWO tasks over one column part.  Entries of two words; sh = lane & 31.  A task without work (beyond the batch's last one)
// points both addresses at entry 0: its row is Pref(0) xor Pref(0) = 0 and counts nothing -- no predicate, no branch.  The
// two tasks' DPP reads interleave, so every read follows the write of its source by the two required wait states; C_2 accumulates in the low half of the packed counter by the popcount-add itself, C_3 by one shift-add.
typedef unsigned int enth_u2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const enth_u2 *enth_lds_cv2;
TSFA_DEV void enth_sweep2(unsigned int rl0, unsigned int rh0, unsigned int rl1, unsigned int rh1, unsigned int sh, unsigned int &ct0,
                          unsigned int &ct1) {
    const enth_u2 l0 = *(enth_lds_cv2)rl0, h0 = *(enth_lds_cv2)rh0, l1 = *(enth_lds_cv2)rl1, h1 = *(enth_lds_cv2)rh1;
    const unsigned int e0 = __builtin_amdgcn_alignbit(h0.y ^ l0.y, h0.x ^ l0.x, sh);  // the row rotated left by the lane index
    const unsigned int e1 = __builtin_amdgcn_alignbit(h1.y ^ l1.y, h1.x ^ l1.x, sh);
    unsigned int m0, m1, c30, c31;
    asm("s_nop 1\n\t"
        "v_and_b32_dpp %4, %6, %6" TSFA_ENTB_DPP
        "v_and_b32_dpp %5, %7, %7" TSFA_ENTB_DPP
        "v_bcnt_u32_b32 %0, %4, %0\n\t"
        "v_bcnt_u32_b32 %1, %5, %1\n\t"
        "v_and_b32_dpp %4, %4, %4" TSFA_ENTB_DPP
        "v_and_b32_dpp %5, %5, %5" TSFA_ENTB_DPP
        "v_bcnt_u32_b32 %2, %4, 0\n\t"
        "v_bcnt_u32_b32 %3, %5, 0\n\t"
        "v_lshl_add_u32 %0, %2, 16, %0\n\t"
        "v_lshl_add_u32 %1, %3, 16, %1"
        : "+v"(ct0), "+v"(ct1), "=&v"(c30), "=&v"(c31), "=&v"(m0), "=&v"(m1)
        : "v"(e0), "v"(e1));
}

// Table of one column part.  Thread t owns the E = ceil(n / 1024) CONSECUTIVE ranks [t E, (t + 1) E) of the sample order (the
// running OR of a prefix needs them in order) and keeps their sample indices in registers for the whole series: the order
// lives in HBM here, and re-reading it per part was 90 % of the first version's time.  Entry of rank p: slot
// (p / E) * SE + p % E with SE = E | 1 -- an odd stride between the threads' blocks, so the 64-bit stores of a wavefront spread
// over all bank pairs (blocks 128 bytes apart, E = 16, are one bank for every lane: a 32-way conflict on each of 2 E stores).
//
// The exclusive prefix of every thread and part -- the bits of all lower ranks in the part's two words -- does not depend on the
// task batch: enth_prefix computes it ONCE per series (a sweep over the thread's ranks, a wavefront OR-scan and one barrier
// per part) into the slot's `pre` array; a batch then builds a part's table from one coalesced 8-byte load per thread and one
// sweep over its ranks.  (Recomputed per batch it was three quarters of the kernel's instructions.)
#define TSFA_ENTH_MAXE ((TSFA_ENTH_MAXN + 1023) / 1024)
typedef unsigned int enth_w2 __attribute__((ext_vector_type(2)));
// the two table words (low: word t0, high: word t0 + 1) that sample j sets; wrap: the part's second word is word 0
// (single: the row is ONE word, n <= 31 -- it is its own halo)
TSFA_DEV unsigned long long enth_bits(unsigned int j, unsigned int x0, bool wrap, bool single) {
    unsigned int d = j - x0;
    if (wrap) d = (j < 32u) ? (j + 32u) : d;
    unsigned long long v = (unsigned long long)(d < 64u ? 1u : 0u) << (d & 63u);
    if (single) v |= v >> 32;
    return v;
}
TSFA_DEV void enth_prefix(const Blk &b, int n, const unsigned int (&pj)[TSFA_ENTH_MAXE], int NW, enth_w2 *pre, unsigned int *wtot) {
    const int lane = b.tid & 63, wave = b.tid >> 6;
    for (int part = 0; part < NW; ++part) {
        const unsigned int x0 = (unsigned int)part << 5;
        const bool wrap = (part == NW - 1);
        unsigned long long tot = 0ull;
#pragma unroll
        for (int e = 0; e < TSFA_ENTH_MAXE; ++e) tot |= enth_bits(pj[e], x0, wrap, NW == 1);   // (ranks beyond the thread's: index 0xFFFFFFF0, no bit)
        const unsigned int inc0 = entb_wave_or_scan((unsigned int)tot), inc1 = entb_wave_or_scan((unsigned int)(tot >> 32));
        unsigned int *wt = wtot + (part & 1) * 2 * TSFA_ENTB_MAXWAVES;   // two buffers: one barrier per part
        if (lane == 63) { wt[wave * 2] = inc0; wt[wave * 2 + 1] = inc1; }
        unsigned int run0 = entb_from_prev(inc0), run1 = entb_from_prev(inc1);
        blk_sync();
        for (int v = 0; v < wave; ++v) { run0 |= wt[v * 2]; run1 |= wt[v * 2 + 1]; }
        enth_w2 w;
        w.x = run0;
        w.y = run1;
        pre[(size_t)part * b.nt + b.tid] = w;
    }
    blk_sync();
}
TSFA_DEV void enth_build_table(const Blk &b, int n, const unsigned int (&pj)[TSFA_ENTH_MAXE], int E, int part, int NW, enth_w2 start,
                               unsigned int *table) {
    const unsigned int x0 = (unsigned int)part << 5;
    const bool wrap = (part == NW - 1);
    const int SE = E | 1;
    enth_w2 *t2 = (enth_w2 *)(void *)table + (size_t)b.tid * SE;
    unsigned long long run = (unsigned long long)start.x | ((unsigned long long)start.y << 32);
    const int p0 = b.tid * E;
#pragma unroll
    for (int e = 0; e < TSFA_ENTH_MAXE; ++e) {
        if (e < E && p0 + e < n) {
            enth_w2 w;
            w.x = (unsigned int)run;
            w.y = (unsigned int)(run >> 32);
            t2[e] = w;
            run |= enth_bits(pj[e], x0, wrap, NW == 1);
        }
    }
    // entry n (every column): the thread that owns rank n - 1 has the total in `run`; its slot follows that rank's
    if (p0 < n && p0 + E >= n) {
        enth_w2 w;
        w.x = (unsigned int)run;
        w.y = (unsigned int)(run >> 32);
        ((enth_w2 *)(void *)table)[(n / E) * SE + (n % E)] = w;
    }
    blk_sync();
}

// racc[4 k .. 4 k + 3] of the nk <= TSFA_ENTB_MAXK tolerances thr[0 .. nk) (see entropy_bits_batch)
TSFA_DEV void entropy_hbits_batch(const Blk &b_in, const double *xs, int n, const double *thr, int nk, const unsigned short *perm,
                                  double *xsrt, unsigned int *rng, unsigned int *cnt, enth_w2 *pre, unsigned int *table,
                                  unsigned int *wtot, double *racc) {
    const Blk b = entb_opaque(b_in);
    const int S = TSFA_ENTH_S;
    const int nrow_m = n - 1, nrow_m1 = n - 2;
    const int NW = (n + 32) >> 5;  // row words: at least one zero guard column
    const int nstrips = ((nrow_m + TSFA_ENTB_STRIP - 1) / TSFA_ENTB_STRIP + 1) / 2;  // pairs of half-strips
    const int lane = b.tid & 63, wave = __builtin_amdgcn_readfirstlane(b.tid >> 6), nw = b.nt >> 6;
    const unsigned int tbase = entb_lds_addr(table);
    const int ntask = nstrips * nk;
    const unsigned int sh = (unsigned int)(lane & 31);
    const int lane_row = (lane >> 5) * TSFA_ENTB_STRIP + (lane & 31);  // template of the lane within its pair of half-strips
    const unsigned int lane_off = tbase;   // (the rotation is relative to the half-strip: every lane reads its entries' two words)
    const unsigned int kmagic = 65536u / (unsigned int)nk + 1u;  // id / nk == (id * kmagic) >> 16 for id < 10 000
    blk_sync();
    entb_ranges<TSFA_ENTH_S>(b, xs, n, thr, nk, perm, xsrt, rng);   // every tolerance of the batch: rng[k * n + sample]
    // this thread's slice of the sample order and, from it, its exclusive prefix in every column part
    const int E = (n + b.nt - 1) / b.nt, SE = E | 1;
    unsigned int pj[TSFA_ENTH_MAXE];
#pragma unroll
    for (int e = 0; e < TSFA_ENTH_MAXE; ++e) {
        const int p = b.tid * E + e;
        pj[e] = (e < E && p < n) ? (unsigned int)perm[p] : 0xFFFFFFF0u;
    }
    enth_prefix(b, n, pj, NW, pre, wtot);
    for (int id0 = 0; id0 < ntask; id0 += TSFA_ENTH_MAXT * nw) {
        unsigned int rl[TSFA_ENTH_MAXT], rh[TSFA_ENTH_MAXT], ct[TSFA_ENTH_MAXT];
#pragma unroll
        for (int tt = 0; tt < TSFA_ENTH_MAXT; ++tt) {
            const int id = id0 + wave + tt * nw;
            rl[tt] = lane_off;
            rh[tt] = lane_off;
            ct[tt] = 0u;
            if (id < ntask) {
                const int s = (int)(((unsigned int)id * kmagic) >> 16), k = id - s * nk;
                const int i = s * (2 * TSFA_ENTB_STRIP) + lane_row;
                const unsigned int r = (i < n) ? rng[k * n + i] : 0u;
                const unsigned int lo = (r & 0xFFFFu) >> 1, hi = r >> 17;   // ranks (entb_ranges packs rank * S)
                rl[tt] = lane_off + 8u * ((lo / (unsigned int)E) * (unsigned int)SE + lo % (unsigned int)E);
                rh[tt] = lane_off + 8u * ((hi / (unsigned int)E) * (unsigned int)SE + hi % (unsigned int)E);
            }
        }
        enth_w2 start = pre[b.tid];
        for (int part = 0; part < NW; ++part) {
            const enth_w2 cur = start;
            if (part + 1 < NW) start = pre[(size_t)(part + 1) * b.nt + b.tid];   // in flight while this part is built and swept
            enth_build_table(b, n, pj, E, part, NW, cur, table);
            {   // two tasks at a time (four, or a prefetch of the next pair's entries, want more registers than 128: the task
                //  addresses were then spilled and reloaded from scratch one by one inside this loop -- 15 .. 70 % slower)
                static_assert(TSFA_ENTH_MAXT % 2 == 0, "tasks are swept in pairs");
#pragma unroll
                for (int g2 = 0; g2 < TSFA_ENTH_MAXT; g2 += 2) enth_sweep2(rl[g2], rh[g2], rl[g2 + 1], rh[g2 + 1], sh, ct[g2], ct[g2 + 1]);
            }
            blk_sync();
        }
#pragma unroll
        for (int tt = 0; tt < TSFA_ENTH_MAXT; ++tt) {
            const int id = id0 + wave + tt * nw;
            if (id < ntask) {
                const int s = (int)(((unsigned int)id * kmagic) >> 16), k = id - s * nk;
                const int i = s * (2 * TSFA_ENTB_STRIP) + lane_row;
                if ((lane & 31) < TSFA_ENTB_STRIP && i < nrow_m) cnt[k * n + i] = ct[tt];
            }
        }
    }
    blk_sync();
    // totals, as in entropy_bits_batch: products of the counts, one logarithm per row of lanes (part: the table's storage)
    {
        const int K = TSFA_ENTB_MAXK;
        double pm[TSFA_ENTB_MAXK], pm1[TSFA_ENTB_MAXK];
        int sc[TSFA_ENTB_MAXK], sc1[TSFA_ENTB_MAXK], nm[TSFA_ENTB_MAXK], nm1[TSFA_ENTB_MAXK];
        double *part = (double *)(void *)table;
        // (a packed wavefront total holds sum C < 2^21: at most 2^21 / n rows per thread and chunk -- one row of lanes per chunk here)
        for (int c0 = 0; c0 < nrow_m; c0 += b.nt) {
#pragma unroll
            for (int k = 0; k < K; ++k) { pm[k] = 1.0; pm1[k] = 1.0; sc[k] = 0; sc1[k] = 0; nm[k] = 0; nm1[k] = 0; }
            const int i = c0 + b.tid;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (k < nk && i < nrow_m) {
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
            entb_totals(b, nk, 0, pm, pm1, sc, sc1, nm, nm1, nrow_m, nrow_m1, part, racc, c0 > 0);
        }
    }
    blk_sync();
}

// The ENTROPY specs (every one with m = 2) of one series of any length up to TSFA_ENTH_MAXN.  g: the samples in HBM.
// H: the workgroup's HBM slot (tsfa_layout.h: EntropyHugeSlot).
template <typename T, class SLOT>
TSFA_DEV void fam_entropy_series_hbits(const Blk &b, const T *g, int n, const TsfaSpec *specs, int nspecs, double *out_row,
                                       const SLOT &H, unsigned int *table, unsigned int *wtot, const double *stats) {
    double *xs = H.xs;
    for (int i = b.tid; i < n; i += b.nt) xs[i] = (double)g[i];
    blk_sync();
    const double dn = (double)n;
    const double mean = stats ? stats[TSFA_STATS_MEAN] : np_sum(b, n, [=](int i) { return xs[i]; }) / dn;
    const double var = stats ? stats[TSFA_STATS_VAR] : np_sum(b, n, [=](int i) { const double d = xs[i] - mean; return d * d; }) / dn;
    const double sd = sqrt(var);
    blk_sync();
    if (n >= 3) entropy_sort_templates(b, xs, n + 1, H.perm, next_pow2(n), sizeof(T) == 4);  // all n samples (a "template" per sample)
    double *thr = H.thr, *racc = H.thr + TSFA_ENTB_MAXK;
    for (int first = 0; first < nspecs; first += TSFA_ENTB_MAXK) {
        const int nk = (nspecs - first < TSFA_ENTB_MAXK) ? (nspecs - first) : TSFA_ENTB_MAXK;
        blk_sync();
        for (int k = b.tid; k < nk; k += b.nt) {
            const TsfaSpec sp = specs[first + k];
            thr[k] = ent_tolerance((sp.calc == TSFA_C_SAMPLE_ENTROPY) ? 0.2 * sd : sp.p[1] * sd);
        }
        blk_sync();
        if (n >= 3) entropy_hbits_batch(b, xs, n, thr, nk, H.perm, H.xsrt, H.rng, H.cnt, (enth_w2 *)(void *)H.pre, table, wtot, racc);
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

#endif  // TSFA_GPU && TSFA_LONG
#endif
