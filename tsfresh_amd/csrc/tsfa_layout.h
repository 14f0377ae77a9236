// LDS layouts of the per-family kernels.  The same carve routine runs on the device (to obtain pointers) and on
// the host with a null base (to obtain the dynamic-LDS size for the launch), so the two can never disagree.
#ifndef TSFA_LAYOUT_H
#define TSFA_LAYOUT_H

#include <stddef.h>
#include <stdint.h>

#include "tsfa_common.h"
#include "fam_cwt.h"
#include "fam_seq.h"
#include "tsfa_entb_params.h"

#if defined(__HIPCC__)
#define TSFA_HD __host__ __device__ inline
#else
#define TSFA_HD inline
#endif

#define TSFA_LDS_LIMIT (160 * 1024)

struct LdsCarve {
    unsigned char *base;
    size_t off;
    template <class U>
    TSFA_HD U *take(size_t count) {
        off = (off + 15) & ~(size_t)15;
        U *r = (U *)(base + off);
        off += count * sizeof(U);
        return r;
    }
};

TSFA_HD int tsfa_pow2_ceil(int n) {
    int p = 1;
    while (p < n) p <<= 1;
    return p;
}

struct BasicLds {
    double *red; NpScratch *np; void *xs; double *w; double *cum; double *altc; int *iw; TsfaSpec *stage; double *ctx;
    // xs_bytes: element size of the LDS-resident series (4: float32 input kept as float32, 8: float64)
    // part: 1 = k_basic (w holds the sliding maxima of number_peaks / the distance codes: maxn elements of the input
    //       precision, at least maxn shorts), 2 = k_trend (w = maxn float64: cumulative |x|, chunk aggregates), 3 = both
    // The numpy-order scratch (np_sum: the statistics / sum |x|, always finished before w is written) shares w.
    // small_w (k_trend, TsfaAltPlan::small_w): the plan's calculators need no n-double array -- w holds the scratch of
    //       the banded index_mass_quantile (nt offsets + 2 counts per q) only: half the LDS per series at n = 1024
    TSFA_HD size_t carve(unsigned char *base, int maxn, int nt, int xs_bytes = 8, int part = 3, int small_w = 0) {
        LdsCarve c{base, 0};
        // a one-wavefront workgroup never touches the cross-wavefront scratch (every blk_* reduction and blk_bcast0 test
        // nt > 64 first): 16 bytes instead of 512 -- k_trend at 1024 samples was 16 bytes above 160 KB / 16
        red = c.take<double>(nt <= 64 ? 2 : TSFA_RED_DOUBLES);
        xs = c.take<unsigned char>((size_t)maxn * xs_bytes);
        size_t wb = (part & 2) ? (size_t)maxn * sizeof(double) : (size_t)maxn * xs_bytes;
        if ((part & 2) && small_w) wb = (size_t)(nt + 2 * 16 + 8) * sizeof(double);
        if (wb < sizeof(NpScratch)) wb = sizeof(NpScratch);
        unsigned char *u = c.take<unsigned char>(wb);
        w = (double *)u;   // chunk aggregates (agg_linear_trend) ...
        cum = w;           // ... aliased with the cumulative |x| of index_mass_quantile (the cache is invalidated)
        np = (NpScratch *)u;
        altc = (part & 2) ? c.take<double>(8 * 16) : nullptr;
        ctx = c.take<double>(32);  // TSFA_BASIC_CTX: per-series values read by the epilogue columns
        iw = c.take<int>((4 * nt > 256) ? 4 * nt : 256);
#if defined(TSFA_SPEC_LDS)
        stage = c.take<TsfaSpec>(TSFA_SPEC_BATCH);
#else
        stage = nullptr;
#endif
        return c.off;
    }
};

// Row form of the BASIC / TREND families (tsfa_common.h: BlkRow): four series of at most TSFA_ROW_MAXN samples per wavefront,
// each with a carve of its own.  No numpy-order scratch (np_sum's row form keeps its three leaves in registers), no
// cross-wavefront scratch.  The stride between the rows' carves is 64 bytes past a multiple of 128: a ds_read_b32 serves lanes
// 0-31 in one cycle from 32 banks, so rows 0 / 1 (and 2 / 3), reading 16 consecutive dwords each, must start 16 banks apart.
#define TSFA_ROW_BINS 64
struct BasicRowLds {
    double *red; void *xs; double *w; double *cum; double *altc; int *iw; double *ctx;
    TSFA_HD static size_t row_bytes(int maxn, int xs_bytes, int part, int small_w) {
        BasicRowLds L;
        const size_t used = L.carve(nullptr, maxn, xs_bytes, part, small_w);
        return ((used + 127) & ~(size_t)127) + 64;
    }
    TSFA_HD size_t carve(unsigned char *base, int maxn, int xs_bytes, int part, int small_w) {
        LdsCarve c{base, 0};
        red = c.take<double>(2);
        xs = c.take<unsigned char>((size_t)maxn * xs_bytes);
        size_t wb = (part & 2) ? (size_t)maxn * sizeof(double) : (size_t)maxn * xs_bytes;
        if ((part & 2) && small_w) wb = (size_t)(TSFA_ROW_LANES + 2 * 16 + 8) * sizeof(double);
        if (wb < 64) wb = 64;
        w = (double *)c.take<unsigned char>(wb);
        cum = w;
        altc = (part & 2) ? c.take<double>(8 * 16) : nullptr;
        ctx = c.take<double>(32);
        // iw: BASIC -- the histogram of binned_entropy in rounds of TSFA_ROW_BINS counters (blk_binned_entropy's `cap`) and
        // benford's ten digit counters; TREND -- the six raw sums per agg_linear_trend regression (16 keys x 6 doubles).
        // (1 KB per row here was what held k_basic_rows at 11 workgroups per CU: 2.75 wavefronts per SIMD)
        iw = c.take<int>((part & 2) ? 16 * 6 * 2 : TSFA_ROW_BINS);
        return c.off;
    }
};

struct SortLds {
    double *red; NpScratch *np; void *xs; void *srt; double *w; int *iw; double *cq; TsfaSpec *stage; double *ctx;
    // xs_bytes: element size of the resident series and its sorted copy (4: float32 input kept as float32, 8: float64)
    // w_doubles: scratch of the Langevin fit (6 r + 16 + r (m + 1) doubles for the plan's largest (m, r): tsfa_prepare_family),
    //            at least 320; the ordinal-pattern histogram of permutation_entropy takes what is there and sweeps the
    //            pattern space in as many passes as it needs (fam_sort.h)
    TSFA_HD size_t carve(unsigned char *base, int maxn, int nt, int xs_bytes = 8, int w_doubles = 1280) {
        (void)nt;
        LdsCarve c{base, 0};
        red = c.take<double>(TSFA_RED_DOUBLES);
        np = c.take<NpScratch>(1);
        xs = c.take<unsigned char>((size_t)maxn * xs_bytes);
        srt = c.take<unsigned char>((size_t)tsfa_pow2_ceil(maxn) * xs_bytes);
        w = c.take<double>(w_doubles);
        iw = (int *)w;              // ... aliased with the ordinal-pattern histogram: never live together
        cq = c.take<double>(5 * TSFA_CQ_MAX);  // change_quantiles results per corridor
        ctx = c.take<double>(8);               // TSFA_SORT_CTX: values read by the epilogue columns
#if defined(TSFA_SPEC_LDS)
        stage = c.take<TsfaSpec>(TSFA_SPEC_BATCH);
#else
        stage = nullptr;
#endif
        return c.off;
    }
};

struct SpectralLds {
    double *red; void *xs; double *Xr; double *Xi; double *tc; double *ts; double *win; double *pxx; int *iw;
    // dft_n: length of the longest non-power-of-two series whose DFT twiddles must live in LDS (0 = none / global)
    // xs_bytes: element size of the resident series (4: float32 input kept as float32, 8: float64)
    TSFA_HD size_t carve(unsigned char *base, int maxn, int dft_n, int xs_bytes = 8) {
        LdsCarve c{base, 0};
        red = c.take<double>(TSFA_RED_DOUBLES);
        xs = c.take<unsigned char>((size_t)maxn * xs_bytes);
        const int nx = (maxn / 2 + 2 > 260) ? maxn / 2 + 2 : 260;
        Xr = c.take<double>(nx);
        Xi = c.take<double>(nx);
        // the table-driven DFT serves non-power-of-two lengths <= 256 (short series and their Welch segments); a batch
        // without any non-power-of-two length (dft_n == 0) needs no table: 4 KB of LDS, two more resident series per CU
        const int nd = (dft_n > 256) ? dft_n : (dft_n > 0 ? 256 : 2);
        tc = c.take<double>(nd);
        ts = c.take<double>(nd);
        win = c.take<double>(256);
        // tc | ts | win | pxx are contiguous; with nd = 256 and pxx padded to 260 they are also the 1024-double chirp table
        // of blk_rfft_bluestein (all four are dead by the time the full-length transform runs): chirp_tab()
        pxx = c.take<double>(dft_n > 0 ? 260 : 132);
        // the bin counters of fourier_entropy (<= 128 ints) live in the Hann window's storage: the window is dead once
        // blk_welch has returned, and the 512 bytes are what kept a tenth workgroup off a CU at 1024 samples (16 480 B)
        iw = (int *)(void *)win;
        has_chirp_tab = (nd == 256 && dft_n > 0);
        return c.off;
    }
    bool has_chirp_tab = false;
    TSFA_HD double *chirp_tab() const { return has_chirp_tab ? tc : nullptr; }
};

struct ArLds {
    double *red; NpScratch *np; double *xc; double *aw;
    // P: leading dimension of the normal matrices = (max regressors) + 1, chosen by the host for the batch
    TSFA_HD static int scratch_doubles(int P) { return 2 * P * P + 7 * P + 64 + 16 + 48 + 40 + 128; }
    // xs_bytes: element size of the resident series (4: float32 input kept as float32, 8: float64)
    TSFA_HD size_t carve(unsigned char *base, int maxn, int P, int xs_bytes = 8) {
        LdsCarve c{base, 0};
        red = c.take<double>(TSFA_RED_DOUBLES);
        xc = (double *)(void *)c.take<unsigned char>((size_t)(maxn + 64 + 16) * xs_bytes);  // TSFA_AR_PADL + n + TSFA_AR_PADR (fam_ar.h)
        // the numpy-order scratch is only used for x.mean(), before the matrices exist: it shares their storage
        const size_t ab = (size_t)scratch_doubles(P) * sizeof(double);
        unsigned char *u = c.take<unsigned char>(ab > sizeof(NpScratch) ? ab : sizeof(NpScratch));
        aw = (double *)u;
        np = (NpScratch *)u;
        return c.off;
    }
};

// second pass of the AR family (fam_ar_dd.h): one workgroup per listed series, matrices in double-double
struct ArDdLds {
    double *red; double *scratch;
    TSFA_HD static int scratch_doubles(int P) { return 4 * P * P + 19 * (P + 1) + P + 10; }
    TSFA_HD size_t carve(unsigned char *base, int P) {
        LdsCarve c{base, 0};
        red = c.take<double>(TSFA_RED_DOUBLES);
        scratch = c.take<double>((size_t)scratch_doubles(P));
        return c.off;
    }
};

struct EntropyLds {
    double *red; NpScratch *np; double *xs; double *thr; unsigned short *perm; unsigned int *refs; unsigned int *cnt;
    // idx_bytes: 2, or 4 for the pair sweep of the long-series build (32-bit sample order: series of any length)
    // with_cnt 1: per-template LDS counters + template references of the symmetric sweep (fam_entropy.h);
    // with_cnt 2: the work region of the bit-matrix sweep (fam_entropy_bits.h) in `cnt`
    TSFA_HD size_t carve(unsigned char *base, int maxn, int with_cnt, int idx_bytes = 2) {
        LdsCarve c{base, 0};
        red = c.take<double>(TSFA_RED_DOUBLES);
        thr = c.take<double>(56);
        xs = c.take<double>(maxn + 4);
        const int np2 = tsfa_pow2_ceil(maxn);
        const int nperm = ((np2 > 64) ? np2 : 64) + 32;
        perm = (unsigned short *)(void *)c.take<unsigned char>((size_t)nperm * (size_t)idx_bytes);
        if (with_cnt == 2 || with_cnt == 3) {  // the numpy-order scratch is dead before the ranges are computed: share its storage
            // (3: the long-series variant -- 16-byte table entries, as many tolerances per round as 16 wavefronts hold)
            refs = nullptr;
            const size_t cb = (with_cnt == 2 ? entb_work_words(maxn)
                                             : entb_work_words(maxn, TSFA_ENTB_QW_LONG + 1, entb_kround(maxn, TSFA_ENTB_MAXK, TSFA_ENTB_MAXWAVES))) * sizeof(unsigned int);
            unsigned char *u = c.take<unsigned char>(cb > sizeof(NpScratch) ? cb : sizeof(NpScratch));
            np = (NpScratch *)u;
            cnt = (unsigned int *)u;
        } else if (with_cnt) {  // the numpy-order scratch is dead before the first sweep: share its storage
            refs = c.take<unsigned int>(nperm);
            const size_t cb = (size_t)(maxn + 16) * ((maxn <= 1024) ? 4 : 3) * sizeof(unsigned int);  // staged sweep: 4 words / template
            unsigned char *u = c.take<unsigned char>(cb > sizeof(NpScratch) ? cb : sizeof(NpScratch));
            np = (NpScratch *)u;
            cnt = (unsigned int *)u;
        } else {
            np = c.take<NpScratch>(1);
            refs = nullptr;
            cnt = nullptr;
        }
        return c.off;
    }
};

// fam_entropy_hbits.h (long-series build): HBM slot of one workgroup
struct EntropyHugeSlot {
    double *thr; double *xs; double *xsrt; unsigned short *perm; unsigned int *rng; unsigned int *cnt; unsigned int *pre;
    TSFA_HD size_t carve(unsigned char *base, int maxn) {
        LdsCarve c{base, 0};
        thr = c.take<double>(64);
        xs = c.take<double>((size_t)maxn + 4);
        const int np2 = tsfa_pow2_ceil(maxn);
        xsrt = c.take<double>((size_t)np2);
        perm = c.take<unsigned short>((size_t)np2 + 32);
        rng = c.take<unsigned int>((size_t)TSFA_ENTB_MAXK * maxn);
        cnt = c.take<unsigned int>((size_t)TSFA_ENTB_MAXK * maxn + 8);
        pre = c.take<unsigned int>((size_t)((maxn + 32) >> 5) * 1024 * 2);   // [column part][thread]: two words (enth_prefix)
        return c.off;
    }
};
// LDS of the kernel: reduction scratch | table of one column part | cross-wavefront scratch of the table build
TSFA_HD size_t entropy_huge_lds_bytes(int maxn) {
    const size_t e = ((size_t)maxn + 1023) / 1024;   // ranks per thread; the threads' blocks are e | 1 entries apart (bank spread)
    size_t table = ((size_t)1024 * (e | 1) + 2) * TSFA_ENTH_S * sizeof(unsigned int);
    if (table < sizeof(NpScratch) + 64) table = sizeof(NpScratch) + 64;   // the numpy-order sums run before the first table
    const size_t totals = (2 * TSFA_ENTB_MAXK * TSFA_ENTB_MAXWAVES * 4) * sizeof(double) + (2 * TSFA_ENTB_MAXK * TSFA_ENTB_MAXWAVES + 4) * sizeof(unsigned int);
    if (table < totals) table = totals;   // ... and the partial products of the totals after the last one
    return TSFA_RED_DOUBLES * sizeof(double) + ((table + 15) & ~(size_t)15) + (size_t)2 * TSFA_ENTB_MAXWAVES * TSFA_ENTH_S * sizeof(unsigned int) + 64;
}


struct SeqLds {
    double *red; unsigned char *seq; uint32_t *tab; double *edges;
    // group: chains parsed side by side; stride: bytes per symbol row; tab_words / edge_doubles: TsfaSeqGroup totals
    // nwaves: wavefronts of the workgroup -- the kernel's only use of `red` is one slot per wavefront (blk_min / blk_max):
    // 16 bytes instead of 512 for the 128-thread workgroups of series up to 2048 samples, which is what takes a series from
    // 13 720 to 13 224 bytes of LDS: TWELVE series per CU instead of eleven (160 KB / 12 = 13 653)
    TSFA_HD size_t carve(unsigned char *base, int group, int stride, int tab_words, int edge_doubles, int nwaves = 16) {
        LdsCarve c{base, 0};
        red = c.take<double>(nwaves < 2 ? 2 : (nwaves + 1) & ~1);
        edges = c.take<double>(edge_doubles + 2);
        tab = c.take<uint32_t>((size_t)tab_words);
        seq = c.take<unsigned char>((size_t)group * stride + 16);
        return c.off;
    }
};


// k_perm (fam_perm.h): the series in the input precision, one histogram of every ordinal pattern of the set of dimensions,
// the table of logarithms.  nwaves x 5 doubles of reduction scratch (blk_sum_multi).
struct PermLds {
    double *red; double *ltab; int *iw; void *xs;
    TSFA_HD size_t carve(unsigned char *base, int maxn, int nt, int elem_bytes, int hist_words, int log_doubles) {
        LdsCarve c{base, 0};
        const int nwaves = (nt + 63) >> 6;
        red = c.take<double>(nwaves * 8 < 16 ? 16 : nwaves * 8);
        ltab = c.take<double>((size_t)log_doubles);
        iw = c.take<int>((size_t)hist_words + 4);
        xs = (void *)c.take<unsigned char>((size_t)(maxn + 8) * elem_bytes);
        return c.off;
    }
};

struct CwtPeaksLayout {
    CwtPeaksLds p;
    // mode 1: the series is staged between zero halos (register-tiled convolutions, no second row needed);
    // mode 0: no staging (very long series): the CWT rows are evaluated column by column from HBM
    // xs_bytes: element size of the padded copy (4: float32 input kept as float32, 8: float64)
    // idx_bytes: size of a column / line index (cwt_idx_t of the build that runs; the host asks with 4 for the long-series build)
    TSFA_HD size_t carve(unsigned char *base, int maxn, int mode, int xs_bytes = 8, int idx_bytes = (int)sizeof(cwt_idx_t)) {
        LdsCarve c{base, 0};
        p.red = c.take<double>(TSFA_RED_DOUBLES);
        p.row0 = c.take<double>(maxn + 4);
        p.rowv = nullptr;
        p.xpad = mode ? (void *)c.take<unsigned char>((size_t)(maxn + 2 * TSFA_CWTP_HALO + 8) * xs_bytes) : nullptr;
        p.taps = c.take<double>(TSFA_CWTP_MAXTAPS + 16);
        p.mask = c.take<unsigned short>(maxn);
        p.lcol = (cwt_idx_t *)(void *)c.take<unsigned char>((2 * (size_t)maxn + 8) * idx_bytes);  // lcol | linf, contiguous: phase A's edge values (16 B per 4 columns)
        p.linf = p.lcol + maxn;
        p.colmap = (cwt_idx_t *)(void *)c.take<unsigned char>(2 * (size_t)maxn * idx_bytes);  // colmap | mline, contiguous: phase C argsorts into both
        p.mline = p.colmap + maxn;
        p.misc = c.take<int>(8);
        return c.off;
    }
};

#endif
