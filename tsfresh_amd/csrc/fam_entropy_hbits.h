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

#define TSFA_ENTH_MAXT 24                 // tasks per wavefront: three registers each (two entry addresses, the packed counts)

// one task over one column part: entries of two words; sh = lane & 31
TSFA_DEV unsigned int enth_task_part(unsigned int pl_addr, unsigned int ph_addr, unsigned int sh) {
    typedef unsigned int enth_u2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) const enth_u2 *enth_lds_cv2;
    const enth_u2 l2 = *(enth_lds_cv2)pl_addr, h2 = *(enth_lds_cv2)ph_addr;
    const unsigned int e = __builtin_amdgcn_alignbit(h2.y ^ l2.y, h2.x ^ l2.x, sh);  // the row rotated left by the lane index
    unsigned int c2 = 0u, c3 = 0u, m;
    // M2 = E & E(lane + 1), M3 = M2 & M2(lane + 1); two wait states between the write of a DPP source and its read
    asm("s_nop 1\n\t"
        "v_and_b32_dpp %2, %3, %3" TSFA_ENTB_DPP
        "v_bcnt_u32_b32 %0, %2, %0\n\t"
        "s_nop 1\n\t"
        "v_and_b32_dpp %2, %2, %2" TSFA_ENTB_DPP
        "v_bcnt_u32_b32 %1, %2, %1"
        : "+v"(c2), "+v"(c3), "=&v"(m)
        : "v"(e));
    return c2 | (c3 << 16);
}

// Table of one column part (entb_build_table) from the thread's OWN slice of the sample order held in registers: the order
// lives in HBM here, and the two passes over it per part -- 2 x 16 dependent global loads per thread at 16 384 samples -- were
// 90 % of the first version's time (18 us per part against ~2 us of LDS and vector work).  pj[e] = perm[tid * E + e].
#define TSFA_ENTH_MAXE ((TSFA_ENTH_MAXN + 1023) / 1024)
#define TSFA_ENTH_SLOT(p) ((p) + ((p) >> 4))
TSFA_DEV void enth_build_table(const Blk &b, int n, const int (&pj)[TSFA_ENTH_MAXE], int E, int w0, int NW, unsigned int *table,
                               unsigned int *wtot) {
    const int t0 = w0 % NW, t1 = (w0 + 1) % NW;   // the row words the entry's two words mirror
    const int p0 = b.tid * E;
    unsigned int tot0 = 0u, tot1 = 0u;
#pragma unroll
    for (int e = 0; e < TSFA_ENTH_MAXE; ++e) {
        if (e < E && p0 + e < n) {
            const int j = pj[e], jw = j >> 5;
            const unsigned int bit = 1u << (j & 31);
            tot0 |= (jw == t0) ? bit : 0u;
            tot1 |= (jw == t1) ? bit : 0u;
        }
    }
    const int lane = b.tid & 63, wave = b.tid >> 6;
    const unsigned int inc0 = entb_wave_or_scan(tot0), inc1 = entb_wave_or_scan(tot1);
    if (lane == 63) { wtot[wave * 2] = inc0; wtot[wave * 2 + 1] = inc1; }
    unsigned int run0 = entb_from_prev(inc0), run1 = entb_from_prev(inc1);
    blk_sync();
    for (int v = 0; v < wave; ++v) { run0 |= wtot[v * 2]; run1 |= wtot[v * 2 + 1]; }
    // A thread owns E CONSECUTIVE entries (the running OR needs them in order): at E = 16 the lanes' stores are 128 bytes
    // apart -- one bank for the whole wavefront, a 32-way conflict on every one of 2 E stores (the first version spent more
    // time here than in the sweep).  Entry p therefore lives at slot p + p / 16 (TSFA_ENTH_SLOT): neighbouring lanes land 17
    // entries apart, two lanes per bank pair at worst; the sweep's gathers use the same map.
    typedef unsigned int enth_w2 __attribute__((ext_vector_type(2)));
    enth_w2 *t2 = (enth_w2 *)(void *)table;
#pragma unroll
    for (int e = 0; e < TSFA_ENTH_MAXE; ++e) {
        if (e < E && p0 + e < n) {
            const int p = p0 + e;
            enth_w2 w;
            w.x = run0;
            w.y = run1;
            t2[TSFA_ENTH_SLOT(p)] = w;
            const int j = pj[e], jw = j >> 5;
            const unsigned int bit = 1u << (j & 31);
            run0 |= (jw == t0) ? bit : 0u;
            run1 |= (jw == t1) ? bit : 0u;
        }
    }
    if (b.tid == b.nt - 1) { enth_w2 w; w.x = run0; w.y = run1; t2[TSFA_ENTH_SLOT(n)] = w; }  // entry n: every column
    blk_sync();
}

// racc[4 k .. 4 k + 3] of the nk <= TSFA_ENTB_MAXK tolerances thr[0 .. nk) (see entropy_bits_batch)
TSFA_DEV void entropy_hbits_batch(const Blk &b_in, const double *xs, int n, const double *thr, int nk, const unsigned short *perm,
                                  double *xsrt, unsigned int *rng, unsigned int *cnt, unsigned int *table, unsigned int *wtot,
                                  double *racc) {
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
    // this thread's slice of the sample order, for the table builds
    const int E = (n + b.nt - 1) / b.nt;
    int pj[TSFA_ENTH_MAXE];
#pragma unroll
    for (int e = 0; e < TSFA_ENTH_MAXE; ++e) {
        const int p = b.tid * E + e;
        pj[e] = (e < E && p < n) ? (int)perm[p] : 0;
    }
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
                rl[tt] = lane_off + 8u * TSFA_ENTH_SLOT(lo);
                rh[tt] = lane_off + 8u * TSFA_ENTH_SLOT(hi);
            }
        }
        for (int part = 0; part < NW; ++part) {
            enth_build_table(b, n, pj, E, part, NW, table, wtot);
#pragma unroll
            for (int tt = 0; tt < TSFA_ENTH_MAXT; ++tt) {
                if (id0 + wave + tt * nw < ntask) ct[tt] += enth_task_part(rl[tt], rh[tt], sh);
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
        if (n >= 3) entropy_hbits_batch(b, xs, n, thr, nk, H.perm, H.xsrt, H.rng, H.cnt, table, wtot, racc);
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
