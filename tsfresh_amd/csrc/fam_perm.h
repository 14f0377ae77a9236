// permutation_entropy for ALL dimensions of a plan at once (fc.py:1866-1916; ComprehensiveFCParameters asks for
// tau = 1, D = 3 .. 7, settings.py:264), in a kernel of its own (k_perm) beside the SORT family it belongs to.
//
// Why its own kernel: inside k_sort's column loop each dimension was a sweep of its own -- its window codes, its table
// of logarithms, its histogram passes (four for D = 7: the histogram shared the 2.5 KB Langevin scratch), 85 k of that
// kernel's ~330 k cycles per series -- and fusing the dimensions THERE (measured, profiles/r04_s_*) made the whole
// kernel slower: +100 VGPR spills in every other column, or, out of line, 40 k cycles around the call.  Here the
// working set is the series plus ONE histogram of all 5910 patterns (2 + .. + 7! bins, two 16-bit counters per word:
// 11.8 KB), so every dimension comes from one sweep:
//   1. the ordinal pattern of a window in the PREFIX form of its inversion table: digit r_j = #{l < j : a[l] > a[j]}
//      (j = 1 .. Dm - 1, r_j <= j), most significant first, so the code of the first D elements of a window is its
//      Dm-code divided by Dm! / D!: 21 comparisons per window serve every dimension (55 one by one); the codes of up to
//      eight windows per thread stay in registers;
//   2. one pass of LDS atomics over the windows, all histograms side by side;
//   3. per dimension the number G[c] of windows whose pattern occurs c times -- from the patterns, or from the windows where
//      those are fewer -- again LDS atomics; -sum_k p_k log p_k = -(1 / num) sum_c G[c] (log c - log num) with the logarithms
//      from ONE table (log c, log num per dimension), lane c of the first wavefront holding term c: integers up to the
//      last 64 multiplications, so the value depends neither on the thread count nor on the order of anything;
//   4. one block reduction for all dimensions.
// The codes differ from fam_sort.h's perm_code (another bijection of the same stable pattern): only the histogram
// matters.  Windows near the end of the series hold fewer than Dm elements: their missing digits are 0 and they only
// count for the dimensions they are long enough for.  Ties rank by position (np.argsort, kind="stable"... the oracle's
// note on the reference's quicksort applies unchanged: tests/parity.py R1).
#ifndef TSFA_FAM_PERM_H
#define TSFA_FAM_PERM_H

#include "tsfa_common.h"

#define TSFA_PE_MAXD 7
#define TSFA_PE_MASK 0xF8u   // dimensions 3 .. 7; any other set goes one dimension at a time (fam_sort.h)
#define TSFA_PE_LOGS 64      // counts below this take their logarithm from the table

// compile-time tables of one set of dimensions
template <unsigned dmask>
struct PeSet {
    static constexpr bool has(int D) { return D >= 2 && D <= TSFA_PE_MAXD && ((dmask >> D) & 1u); }
    static constexpr int dm() { int r = 0; for (int D = 2; D <= TSFA_PE_MAXD; ++D) if (has(D)) r = D; return r; }
    static constexpr int dmin() { for (int D = 2; D <= TSFA_PE_MAXD; ++D) if (has(D)) return D; return 0; }
    static constexpr int count() { int c = 0; for (int D = 2; D <= TSFA_PE_MAXD; ++D) if (has(D)) ++c; return c; }
    static constexpr int index(int D) { int c = 0; for (int d = 2; d < D; ++d) if (has(d)) ++c; return c; }
    static constexpr int fact(int D) { int f = 1; for (int k = 2; k <= D; ++k) f *= k; return f; }
    static constexpr int off(int D) { int o = 0; for (int d = 2; d < D; ++d) if (has(d)) o += fact(d); return o; }   // first bin of dimension D
    static constexpr int bins() { return off(dm()) + fact(dm()); }
    // code(D) = code(Dm) / div = (code(Dm) * magic) >> 24 with magic = ceil(2^24 / div): exact for code < 2^13
    static constexpr int magic(int D) { const int div = fact(dm()) / fact(D); return (1 << 24) / div + (((1 << 24) % div) ? 1 : 0); }
};
// 32-bit words of LDS: the histogram (2955, rounded to 2956) + per dimension TSFA_PE_LOGS + 1 counters of windows by their pattern's count
#define TSFA_PE_HIST_WORDS ((((PeSet<TSFA_PE_MASK>::bins() + 1) / 2 + 3) & ~3) + PeSet<TSFA_PE_MASK>::count() * (TSFA_PE_LOGS + 1))

template <class AT>
TSFA_DEV int perm_prefix_code(const AT *a, int L, int Dm, int last) {
    AT r[TSFA_PE_MAXD];
#pragma unroll
    for (int j = 0; j < TSFA_PE_MAXD; ++j) r[j] = a[(j < last) ? j : last];   // (clamped: never reads past the series)
    int code = 0;
#pragma unroll
    for (int j = 1; j < TSFA_PE_MAXD; ++j) {
        if (j < Dm) {   // uniform
            int c = 0;
#pragma unroll
            for (int l = 0; l < j; ++l) c += (r[l] > r[j]) ? 1 : 0;
            code = code * (j + 1) + ((j < L) ? c : 0);
        }
    }
    return code;
}

// code / div from magic (the 37-bit product needs its high word)
TSFA_DEV int pe_div(int code, int magic) {
    return (int)(((unsigned long long)(unsigned)code * (unsigned long long)(unsigned)magic) >> 24);
}

// res[D] for every D of the set: the entropy, NaN where the series is shorter than D.
// iw: TSFA_PE_HIST_WORDS words of LDS (histogram + counters); ltab: TSFA_PE_LOGS + TSFA_PE_MAXD + 1 doubles of LDS.
template <unsigned dmask, class ST>
TSFA_DEV void perm_entropy_all(const Blk &b, const ST *xs_raw, int n, int tau, int *iw, double *ltab,
                               double (&res)[TSFA_PE_MAXD + 1]) {
    typedef PeSet<dmask> S;
    constexpr int Dm = S::dm(), Dmin = S::dmin(), LT = TSFA_PE_LOGS, GW = LT + 1, GOFF = ((S::bins() + 1) / 2 + 3) & ~3;
    int *const G = iw + GOFF;
    static_assert(Dm >= 3 && Dmin < Dm && S::fact(Dm) <= 8192, "at least two dimensions, the largest at most 7");
    int num[TSFA_PE_MAXD + 1];                                         // (indexed by unrolled constants only)
#pragma unroll
    for (int D = 0; D <= TSFA_PE_MAXD; ++D) {
        num[D] = (S::has(D) && n >= D) ? ((n - D) / tau + 1) : 0;
        res[D] = TSFA_NAN;
    }
    const int nwin = num[Dmin];                                        // windows that count for at least one dimension
    if (nwin <= 0) return;
    TSFA_TICKER(tk, 0);
    const bool in_regs = (nwin <= 8 * b.nt);
    auto code_at = [=](int t) {
        const int st = t * tau, L = (n - st < Dm) ? (n - st) : Dm;
        return perm_prefix_code(xs_raw + st, L, Dm, n - 1 - st);
    };
    int codes[8];
    if (in_regs) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int t = b.tid + u * b.nt;
            codes[u] = (t < nwin) ? code_at(t) : -1;
        }
    }
    TSFA_TICK(tk, b, 230);
    for (int c = b.tid; c < LT + TSFA_PE_MAXD + 1; c += b.nt) {
        int arg = c;                                   // c < LT: log c; LT + D: log num[D]
#pragma unroll
        for (int D = 2; D <= TSFA_PE_MAXD; ++D)
            if (c == LT + D) arg = num[D];
        ltab[c] = (arg > 0) ? log((double)arg) : 0.0;
    }
    for (int k = b.tid; k < GOFF + S::count() * GW; k += b.nt) iw[k] = 0;
    blk_sync();
    TSFA_TICK(tk, b, 231);
    auto bump = [=](int bin) {
#if TSFA_GPU
        atomicAdd(&iw[bin >> 1], (bin & 1) ? 0x10000 : 1);
#else
        iw[bin >> 1] += (bin & 1) ? 0x10000 : 1;
#endif
    };
    auto count_of = [=](int bin) {
        const unsigned wv = (unsigned)iw[bin >> 1];
        return (bin & 1) ? (int)(wv >> 16) : (int)(wv & 0xffffu);
    };
    auto bin_of = [=](int code, int D) { return S::off(D) + ((D == Dm) ? code : pe_div(code, S::magic(D))); };
    auto windows = [&](auto fn) {   // fn(code, t) for every window of this thread
        if (in_regs) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (codes[u] >= 0) fn(codes[u], b.tid + u * b.nt);
        } else {
            for (int t = b.tid; t < nwin; t += b.nt) fn(code_at(t), t);
        }
    };
    windows([&](int code, int t) {
#pragma unroll
        for (int D = 2; D <= Dm; ++D)
            if (S::has(D) && t < num[D]) bump(bin_of(code, D));
    });
    blk_sync();
    TSFA_TICK(tk, b, 232);
    // G[D][c] = number of WINDOWS whose pattern occurs c times (c < LT; G[D][LT] != 0: some pattern occurs LT times or more),
    // from the patterns (+ c each) where they are fewer than the windows, else from the windows (+ 1 each): integers,
    // whichever way they are counted and however many threads count them
    auto add = [=](int *p, int v) {
#if TSFA_GPU
        atomicAdd(p, v);
#else
        *p += v;
#endif
    };
#pragma unroll
    for (int D = 2; D <= Dm; ++D) {
        if (!S::has(D) || num[D] <= 0) continue;
        int *g = G + S::index(D) * GW;
        if (S::fact(D) <= num[D]) {
            for (int k = b.tid; k < S::fact(D); k += b.nt) {
                const int c = count_of(S::off(D) + k);
                if (c > 0) add(&g[c < LT ? c : LT], c);
            }
        } else {
            windows([&](int code, int t) {
                const bool live = t < num[D];
                const int c = live ? count_of(bin_of(code, D)) : 0;
#if TSFA_GPU
                // (most patterns of a large dimension occur once: one atomic per wavefront for those instead of one per lane
                // on the same LDS word -- k_perm 0.88 -> 0.75 ms per 100k x 1024, profiles/r04_v_*)
                const unsigned long long once = __ballot(live && c == 1);
                if (live && c == 1) {
                    if ((int)__ffsll((long long)once) - 1 == (b.tid & 63)) atomicAdd(&g[1], (int)__popcll(once));
                } else if (live) {
                    atomicAdd(&g[c < LT ? c : LT], 1);
                }
#else
                if (live) add(&g[c < LT ? c : LT], 1);
#endif
            });
        }
    }
    blk_sync();
    TSFA_TICK(tk, b, 233);
    // -sum_k p_k log p_k = -(1 / num) sum_c G[c] (log c - log num): lane c of the FIRST wavefront holds the c-th term (and
    // its share of the rare patterns beyond the table), the other wavefronts hold exact zeros, so the value does not depend
    // on the number of threads the launch gave the series (it did while every thread summed its own windows: the same
    // series in another launch group -- another shard of a multi-device call -- came out one ulp off)
    const int st = (b.nt < 64) ? b.nt : 64;
    double e[S::count()];
#pragma unroll
    for (int D = 2; D <= Dm; ++D) {
        if (!S::has(D)) continue;
        double acc = 0.0;
        if (num[D] > 0 && b.tid < st) {
            const int *g = G + S::index(D) * GW;
            const double lnum = ltab[LT + D];
            for (int c = b.tid; c < LT; c += st) {
                const int gc = g[c];
                if (gc != 0) acc += (double)gc * (ltab[c] - lnum);
            }
            if (g[LT] != 0)
                for (int k = b.tid; k < S::fact(D); k += st) {
                    const int c = count_of(S::off(D) + k);
                    if (c >= LT) acc += (double)c * (log((double)c) - lnum);
                }
        }
        e[S::index(D)] = acc;
    }
    TSFA_TICK(tk, b, 234);
    blk_sum_multi<S::count()>(b, e);
#pragma unroll
    for (int D = 2; D <= Dm; ++D)
        if (S::has(D) && num[D] > 0) res[D] = -e[S::index(D)] / (double)num[D];
    TSFA_TICK(tk, b, 235);
}

// every permutation_entropy column of the SORT family's list (all of stride tau and a dimension of TSFA_PE_MASK:
// tsfa_prepare_family checks, hints[TSFA_FAM_SORT].d)
template <class ST>
TSFA_DEV void fam_perm_series(const Blk &b, const ST *xs_raw, int n, const TsfaSpec *specs, int nspecs, double *out_row,
                              int tau, int *iw, double *ltab) {
    double res[TSFA_PE_MAXD + 1];
    perm_entropy_all<TSFA_PE_MASK>(b, xs_raw, n, tau, iw, ltab, res);
    for (int s = b.tid; s < nspecs; s += b.nt) {   // lane = column
        if (specs[s].calc != TSFA_C_PERMUTATION_ENTROPY) continue;
        const int D = (int)specs[s].p[1];
        double r = TSFA_NAN;
#pragma unroll
        for (int d = 2; d <= TSFA_PE_MAXD; ++d)
            if (d == D) r = res[d];
        out_row[specs[s].col] = r;
    }
}

#endif
