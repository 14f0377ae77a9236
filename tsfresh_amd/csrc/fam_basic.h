// Family BASIC: reductions, scans and counts over the LDS-resident series.
// Each case cites the reference calculator it restates (fc.py = tsfresh/feature_extraction/feature_calculators.py).
#ifndef TSFA_FAM_BASIC_H
#define TSFA_FAM_BASIC_H

#include "tsfa_common.h"

// decimal table for benford_correlation: dectab[(k + TSFA_DEC_KMIN_OFF) * 9 + (d - 1)] = correctly rounded
// double of the decimal "d e k", k in [-324, 308]
#define TSFA_DEC_KMIN (-324)
#define TSFA_DEC_KMAX 308
#define TSFA_DEC_ROWS (TSFA_DEC_KMAX - TSFA_DEC_KMIN + 1)

// First character of np.format_float_scientific(v) for v >= 0 (fc.py:2369-2371): the leading digit of the
// SHORTEST round-tripping decimal representation of the float64 v.  Because repr() is monotone and
// "d e k" is itself a shortest representation, that digit is max{d : v >= RN(d * 10^k)} in the decade
// k = max{k : v >= RN(10^k)} -- comparisons against the correctly rounded table, no arithmetic.
TSFA_DEV int tsfa_leading_decimal_digit(double v, const double *dectab) {
    if (!(v > 0.0)) return 0;
    if (isinf(v)) v = 1.7976931348623157e308;  // np.nan_to_num
    int e;
    frexp(v, &e);  // v = m * 2^e, m in [0.5, 1)
    int k = (int)floor((double)(e - 1) * 0.30102999566398120);
    if (k < TSFA_DEC_KMIN) k = TSFA_DEC_KMIN;
    if (k > TSFA_DEC_KMAX) k = TSFA_DEC_KMAX;
    while (k > TSFA_DEC_KMIN && v < dectab[(k - TSFA_DEC_KMIN) * 9]) --k;
    while (k < TSFA_DEC_KMAX && v >= dectab[(k + 1 - TSFA_DEC_KMIN) * 9]) ++k;
    const double *row = dectab + (k - TSFA_DEC_KMIN) * 9;
    int d = 1;
    for (int c = 2; c <= 9; ++c)
        if (v >= row[c - 1]) d = c;
    return d;
}

struct BasicStats {
    int n;
    double sum, mean, var, std, vmin, vmax, sumsq;
    int first_max, last_max, first_min, last_min, cnt_max, cnt_min;
};

// want_loc: some column needs the positions / multiplicities of the extrema (first / last_location_of_*,
// has_duplicate_*); a plan without them (MinimalFCParameters) skips that sweep and its six reductions
template <class BT, class XS>
TSFA_DEV void basic_stats(const BT &b, XS xs, int n, BasicStats &st, bool want_loc = true) {
    st.n = n;
    st.sum = np_sum(b, n, [=](int i) { return xs[i]; });          // np.sum
    st.mean = st.sum / (double)n;                                   // np.mean = add.reduce / n
    const double mean = st.mean;
    const double ssd = np_sum(b, n, [=](int i) { const double d = xs[i] - mean; return d * d; });
    st.var = ssd / (double)n;                                       // np.var (numpy/_core/_methods.py:_var)
    st.std = sqrt(st.var);                                          // np.std
    double mn = TSFA_INF, mx = -TSFA_INF, sq = 0.0;
    for (int i = b.tid; i < n; i += b.nt) {
        const double v = xs[i];
        mn = fmin(mn, v);
        mx = fmax(mx, v);
        sq += v * v;
    }
    st.vmin = blk_min(b, mn);
    st.vmax = blk_max(b, mx);
    st.sumsq = blk_sum(b, sq);
    st.first_max = 0; st.last_max = 0; st.first_min = 0; st.last_min = 0; st.cnt_max = 0; st.cnt_min = 0;
    if (!want_loc) return;
    double fmx = (double)n, lmx = -1.0, fmn = (double)n, lmn = -1.0, cmx = 0.0, cmn = 0.0;
    for (int i = b.tid; i < n; i += b.nt) {
        const double v = xs[i];
        if (v == st.vmax) {
            fmx = fmin(fmx, (double)i);
            lmx = fmax(lmx, (double)i);
            cmx += 1.0;
        }
        if (v == st.vmin) {
            fmn = fmin(fmn, (double)i);
            lmn = fmax(lmn, (double)i);
            cmn += 1.0;
        }
    }
    st.first_max = (int)blk_min(b, fmx);
    st.last_max = (int)blk_max(b, lmx);
    st.first_min = (int)blk_min(b, fmn);
    st.last_min = (int)blk_max(b, lmn);
    st.cnt_max = (int)blk_sum(b, cmx);
    st.cnt_min = (int)blk_sum(b, cmn);
}

// longest run of `true` of pred(i), i in [0, n)  (fc.py:102 _get_length_sequences_where + max)
template <class BT, class P>
TSFA_DEV double blk_longest_run(const BT &b, int n, P pred, int *iw) {
    // each thread scans a contiguous segment; segments are stitched by thread 0
    const int chunk = (n + b.nt - 1) / b.nt;
    const int lo = b.tid * chunk;
    const int hi = (lo + chunk < n) ? lo + chunk : n;
    int pre = 0, suf = 0, best = 0, all = 1, run = 0;
    for (int i = lo; i < hi; ++i) {
        if (pred(i)) {
            ++run;
            if (run > best) best = run;
        } else {
            if (all) pre = run;
            all = 0;
            run = 0;
        }
    }
    suf = run;
    if (all) pre = run;
#if TSFA_GPU
    // (pre, suf, best, len) of adjacent segments combine associatively ("all true" <=> pre == len), so the wavefront
    // stitches its 64 segments with an ordered butterfly (6 shuffle steps) instead of a serial walk by thread 0
    int len = (hi > lo) ? (hi - lo) : 0;
    constexpr int LN = BlkLanes<BT>::n;   // lanes that stitch without LDS: the wavefront, or the 16-lane row of the row form
    const int lane = b.tid & (LN - 1);
#pragma unroll
    for (int d = 1; d < LN; d <<= 1) {
        const int opre = __shfl_xor(pre, d), osuf = __shfl_xor(suf, d), obest = __shfl_xor(best, d), olen = __shfl_xor(len, d);
        const bool left = ((lane & d) == 0);  // this lane's aggregate lies to the left of its partner's
        const int Lpre = left ? pre : opre, Lsuf = left ? suf : osuf, Lbest = left ? best : obest, Llen = left ? len : olen;
        const int Rpre = left ? opre : pre, Rsuf = left ? osuf : suf, Rbest = left ? obest : best, Rlen = left ? olen : len;
        int nb = (Lbest > Rbest) ? Lbest : Rbest;
        if (Lsuf + Rpre > nb) nb = Lsuf + Rpre;
        pre = (Lpre == Llen) ? Llen + Rpre : Lpre;
        suf = (Rsuf == Rlen) ? Rlen + Lsuf : Rsuf;
        best = nb;
        len = Llen + Rlen;
    }
    if (b.nt == LN) return (double)best;
    const int nw = b.nt >> 6, wv = b.tid >> 6;
    blk_sync();
    if (lane == 0) { iw[4 * wv + 0] = pre; iw[4 * wv + 1] = suf; iw[4 * wv + 2] = best; iw[4 * wv + 3] = len; }
    blk_sync();
    int carry = 0, gbest = 0;  // every thread walks the (<= 16) wave aggregates
    for (int t = 0; t < nw; ++t) {
        const int tp = iw[4 * t + 0], ts = iw[4 * t + 1], tb = iw[4 * t + 2], tl = iw[4 * t + 3];
        if (carry + tp > gbest) gbest = carry + tp;
        if (tb > gbest) gbest = tb;
        carry = (tp == tl) ? carry + tl : ts;
    }
    blk_sync();
    return (double)gbest;
#else
    (void)iw;
    return (double)best;  // nt = 1: one segment
#endif
}

// agg_linear_trend (fc.py:2171), all distinct (chunk_len, f_agg) regressions of the plan at once:
//   * one sweep per chunk_len computes every requested aggregate of a chunk together (lane = chunk),
//   * the regression sums of the aggregates of one chunk_len are accumulated side by side (independent chains) and
//     reduced back to back,
//   * the scalar tail of linregress (a dozen dependent float64 divisions / square roots, ~4k cycles) runs ONCE with
//     lane = regression instead of once per regression on every lane.
// raw: LDS, 6 doubles per key (m, mean, sxy, syy, y0, y1); w: LDS, >= n doubles; altc[8 * key + 2 + attr] = results.
template <class BT, class XS>
TSFA_DEV void alt_fill_all(const BT &b, XS xs, int n, const TsfaAltPlan &alt, double *w, double *raw,
                           double *altc) {
    const int nkeys = alt.nkeys;
    int k0 = 0;
    TSFA_TICKER(tka, 0);
    // The chunk aggregates are not stored: a chunk is cl consecutive samples of the LDS-resident series, its (up to four)
    // aggregates cost a handful of instructions, and the regression reads every aggregate exactly twice (the mean, then
    // the centred products).  Recomputing them in the second sweep instead of keeping 4 m doubles per series lets the
    // kernel run without an n-double work array (TsfaAltPlan::small_w): 15 -> 8 KB of LDS per series.  `w` is unused.
    (void)w;
    while (k0 < nkeys) {
        const int cl = alt.cl[k0];
        int k1 = k0 + 1;
        while (k1 < nkeys && alt.cl[k1] == cl && k1 - k0 < 4) ++k1;
        const int m = (n + cl - 1) / cl;
        const int ng = k1 - k0;
        if (cl >= n) {  // fc.py: chunk_len >= len(x) -> NaN, for every key of this chunk length
            blk_sync();
            if (b.tid == 0)
                for (int j = 0; j < ng; ++j) raw[6 * (k0 + j)] = 0.0;
            k0 = k1;
            continue;
        }
        int ag[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) ag[j] = (j < ng) ? alt.agg[k0 + j] : -1;
        bool need_max = false, need_min = false, need_mean = false, need_var = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            need_max |= (ag[j] == TSFA_AGG_MAX);
            need_min |= (ag[j] == TSFA_AGG_MIN);
            need_mean |= (ag[j] == TSFA_AGG_MEAN) || (ag[j] == TSFA_AGG_VAR);
            need_var |= (ag[j] == TSFA_AGG_VAR);
        }
        // fc.py:176 _aggregate_on_chunks: the aggregates of chunk c, in the order of the group's keys
        auto chunk_aggs = [=](int c, double (&r)[4]) {
            const int lo = c * cl;
            const int hi = (lo + cl < n) ? lo + cl : n;
            double vmx = xs[lo], vmn = xs[lo], vmean = 0.0, vvar = 0.0;
            if (need_max || need_min)
                for (int i = lo + 1; i < hi; ++i) { const double x = xs[i]; vmx = fmax(vmx, x); vmn = fmin(vmn, x); }
            if (need_mean) vmean = np_leaf_sum(lo, hi - lo, [=](int i) { return xs[i]; }) / (double)(hi - lo);
            if (need_var)
                vvar = np_leaf_sum(lo, hi - lo, [=](int i) { const double d = xs[i] - vmean; return d * d; }) / (double)(hi - lo);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                r[j] = (ag[j] == TSFA_AGG_MAX) ? vmx : (ag[j] == TSFA_AGG_MIN) ? vmn : (ag[j] == TSFA_AGG_MEAN) ? vmean : vvar;
        };
        blk_sync();
        const double dm = (double)m;
        const double xmean = (dm - 1.0) * 0.5;
        double sy[4] = {0.0, 0.0, 0.0, 0.0};
        for (int c = b.tid; c < m; c += b.nt) {
            double r[4];
            chunk_aggs(c, r);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < ng) sy[j] += r[j];
            if (c < 2) {   // linregress of two points reads the two values themselves
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < ng) raw[6 * (k0 + j) + 4 + c] = r[j];
            }
        }
        TSFA_TICK(tka, b, 200);
        double ym[4];
        blk_sum_multi<4>(b, sy);
#pragma unroll
        for (int j = 0; j < 4; ++j) ym[j] = (j < ng) ? sy[j] / dm : 0.0;
        double sxy[4] = {0.0, 0.0, 0.0, 0.0}, syy[4] = {0.0, 0.0, 0.0, 0.0};
        for (int c = b.tid; c < m; c += b.nt) {
            double r[4];
            chunk_aggs(c, r);
            const double dx = (double)c - xmean;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < ng) {
                    const double dy = r[j] - ym[j];
                    sxy[j] += dx * dy;
                    syy[j] += dy * dy;
                }
            }
        }
        double s8[8] = {sxy[0], sxy[1], sxy[2], sxy[3], syy[0], syy[1], syy[2], syy[3]};
        blk_sum_multi<8>(b, s8);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < ng) {
                const double a = s8[j], c2 = s8[4 + j];
                if (b.tid == 0) {
                    double *r = raw + 6 * (k0 + j);
                    r[0] = dm;
                    r[1] = ym[j];
                    r[2] = a;
                    r[3] = c2;
                    if (m < 2) r[5] = 0.0;
                }
            }
        }
        k0 = k1;
        TSFA_TICK(tka, b, 201);
    }
    blk_sync();
    // scalar tail of scipy.stats.linregress(range(m), y), lane = regression (blk_linregress_index restated)
    for (int k = b.tid; k < nkeys; k += b.nt) {
        const double *r = raw + 6 * k;
        double *o = altc + 8 * k + 2;
        const double dm = r[0];
        if (dm == 0.0) {
            for (int a = 0; a < 5; ++a) o[a] = TSFA_NAN;
            continue;
        }
        const double ymean = r[1], xmean = (dm - 1.0) * 0.5;
        const double ssxm = (dm * (dm * dm - 1.0) / 12.0) / dm;
        const double ssxym = r[2] / dm, ssym = r[3] / dm;
        double rr;
        if (ssxm == 0.0 || ssym == 0.0) {
            rr = 0.0;
        } else {
            rr = ssxym / sqrt(ssxm * ssym);
            if (rr > 1.0) rr = 1.0;
            else if (rr < -1.0) rr = -1.0;
        }
        const double slope = ssxym / ssxm;
        const double intercept = ymean - slope * xmean;
        double prob, stderr_;
        if (dm == 2.0) {
            prob = (r[4] == r[5]) ? 1.0 : 0.0;
            stderr_ = 0.0;
        } else {
            const double df = dm - 2.0;
            const double TINY = 1.0e-20;
            const double t = rr * sqrt(df / ((1.0 - rr + TINY) * (1.0 + rr + TINY)));
            prob = alt.want_p ? tsfa_t_pvalue2(t, df) : TSFA_NAN;
            stderr_ = sqrt((1.0 - rr * rr) * ssym / ssxm / df);
        }
        o[TSFA_ATTR_PVALUE] = prob;
        o[TSFA_ATTR_RVALUE] = rr;
        o[TSFA_ATTR_INTERCEPT] = intercept;
        o[TSFA_ATTR_SLOPE] = slope;
        o[TSFA_ATTR_STDERR] = stderr_;
    }
    blk_sync();
    TSFA_TICK(tka, b, 202);
}

// Per-series values the epilogue columns read (LDS, TSFA_BASIC_CTX doubles)
enum { TSFA_CTX_SUM = 0, TSFA_CTX_MEAN, TSFA_CTX_VAR, TSFA_CTX_STD, TSFA_CTX_VMIN, TSFA_CTX_VMAX, TSFA_CTX_SUMSQ,
       TSFA_CTX_FIRST_MAX, TSFA_CTX_LAST_MAX, TSFA_CTX_FIRST_MIN, TSFA_CTX_LAST_MIN, TSFA_CTX_CNT_MAX, TSFA_CTX_CNT_MIN,
       TSFA_CTX_X0, TSFA_CTX_X1, TSFA_CTX_XN2, TSFA_CTX_XN1, TSFA_CTX_LT = 20, TSFA_CTX_LTT = 25, TSFA_BASIC_CTX = 32 };

// Columns [first, nspecs): closed forms of the statistics in ctx and reads of the caches (altc, ctx), lane = column.
// Every lane fetches its own spec (one coalesced vector load for the whole group instead of a scalar-load round trip
// per column) and stores its own value.
template <class BT>
TSFA_DEV void basic_epilogue(const BT &b, const TsfaSpec *specs, int first, int nspecs, int n, const double *ctx,
                             const double *altc, double *out_row) {
    const double dn = (double)n;
    for (int s = first + b.tid; s < nspecs; s += b.nt) {
        const TsfaSpec sp = specs[s];
        const double p0 = sp.p[0];
        double v = TSFA_NAN;
        switch (sp.calc) {
        case TSFA_C_SUM_VALUES: v = ctx[TSFA_CTX_SUM]; break;
        case TSFA_C_MEAN: v = ctx[TSFA_CTX_MEAN]; break;
        case TSFA_C_LENGTH: v = dn; break;
        case TSFA_C_STANDARD_DEVIATION: v = ctx[TSFA_CTX_STD]; break;
        case TSFA_C_VARIANCE: v = ctx[TSFA_CTX_VAR]; break;
        case TSFA_C_ROOT_MEAN_SQUARE: v = sqrt(ctx[TSFA_CTX_SUMSQ] / dn); break;
        case TSFA_C_MAXIMUM: v = ctx[TSFA_CTX_VMAX]; break;
        case TSFA_C_ABSOLUTE_MAXIMUM: v = fmax(fabs(ctx[TSFA_CTX_VMAX]), fabs(ctx[TSFA_CTX_VMIN])); break;
        case TSFA_C_MINIMUM: v = ctx[TSFA_CTX_VMIN]; break;
        case TSFA_C_ABS_ENERGY: v = ctx[TSFA_CTX_SUMSQ]; break;
        case TSFA_C_VARIATION_COEFFICIENT: v = (ctx[TSFA_CTX_MEAN] == 0.0) ? TSFA_NAN : ctx[TSFA_CTX_STD] / ctx[TSFA_CTX_MEAN]; break;
        case TSFA_C_VAR_GT_STD: v = (ctx[TSFA_CTX_VAR] > sqrt(ctx[TSFA_CTX_VAR])) ? 1.0 : 0.0; break;
        case TSFA_C_LARGE_STD: v = (ctx[TSFA_CTX_STD] > p0 * (ctx[TSFA_CTX_VMAX] - ctx[TSFA_CTX_VMIN])) ? 1.0 : 0.0; break;
        case TSFA_C_FIRST_LOCATION_OF_MAXIMUM: v = ctx[TSFA_CTX_FIRST_MAX] / dn; break;
        case TSFA_C_LAST_LOCATION_OF_MAXIMUM: v = 1.0 - ((double)(n - 1) - ctx[TSFA_CTX_LAST_MAX]) / dn; break;
        case TSFA_C_FIRST_LOCATION_OF_MINIMUM: v = ctx[TSFA_CTX_FIRST_MIN] / dn; break;
        case TSFA_C_LAST_LOCATION_OF_MINIMUM: v = 1.0 - ((double)(n - 1) - ctx[TSFA_CTX_LAST_MIN]) / dn; break;
        case TSFA_C_HAS_DUPLICATE_MAX: v = (ctx[TSFA_CTX_CNT_MAX] >= 2.0) ? 1.0 : 0.0; break;
        case TSFA_C_HAS_DUPLICATE_MIN: v = (ctx[TSFA_CTX_CNT_MIN] >= 2.0) ? 1.0 : 0.0; break;
        case TSFA_C_QUERY_SIMILARITY_COUNT: v = TSFA_NAN; break;
        case TSFA_C_AGG_LINEAR_TREND: v = ((int)sp.p[1] >= n) ? TSFA_NAN : altc[8 * (((int)sp.p[3]) & 127) + 2 + (int)p0]; break;
        case TSFA_C_INDEX_MASS_QUANTILE: v = altc[8 * (((int)sp.p[1]) & 127) + 7]; break;
        case TSFA_C_LINEAR_TREND: v = ctx[TSFA_CTX_LT + (int)p0]; break;
        case TSFA_C_LINEAR_TREND_TIMEWISE: v = ctx[TSFA_CTX_LTT + (int)p0]; break;
        default: break;
        }
        out_row[sp.col] = v;
    }
}

// index_mass_quantile (fc.py:1275) for all q of the plan WITHOUT the serial cumulative sum.
// The reference takes the first i with np.cumsum(|x|)[i] / S >= q; np.cumsum rounds after every addend, so its values c_i
// can only be reproduced by one lane adding n terms in a row (that chain was 30 % of k_trend).  But the decision rarely
// depends on the rounding: any other summation order p_i of the same non-negative terms differs from c_i by less than
// 2 n u S, so  p_i < (q - delta) S  implies  c_i / S < q  and  p_i >= (q + delta) S  implies  c_i / S >= q  with
// delta = (2 n + 64) 2^-52.  p_i = o_j + s_i: thread j owns E consecutive samples, s_i its running sum, o_{j+1} =
// o_j + t_j the offsets (64 dependent adds instead of n) -- non-decreasing in i by construction.  Lane (q, bound) counts
// the i below its bound by bisection over the offsets and a walk through ONE thread's samples; where the two counts of a
// q agree, no c_i lies in the band and the count is the reference's index.  Otherwise (a prefix sum that hits q S to
// within 5e-13: integer-valued series do) the function returns false and the caller takes the serial route.
// cum: LDS scratch, >= nt + 2 * nq + 2 doubles.  Writes altc[8 k + 7] for every q.
template <class BT, class XS>
TSFA_DEV bool imq_banded(const BT &b, XS xs, int n, double S, const TsfaAltPlan &alt, double *cum, double *altc) {
    const int nq = alt.nq;
    if (!(S > 0.0) || !(S < TSFA_INF) || n < 1) return false;  // uniform
    const double dn = (double)n;
    const int E = (n + b.nt - 1) / b.nt;
    const int nown = (n + E - 1) / E;  // threads that own samples
    // scratch: nown + 1 + 2 nq doubles -- the small work array of the kernel holds that by construction, the n-double
    // buffer of the serial route only for series longer than it (short series: short chain)
    if (!alt.small_w && nown + 1 + 2 * nq > n) return false;
    const int beg = E * b.tid, end = (beg + E < n) ? beg + E : n;
    double t = 0.0;
    for (int i = beg; i < end; ++i) t += fabs(xs[i]);
    blk_sync();
#if TSFA_GPU
    if (!BlkIsRow<BT>::v && b.nt == 64) {
        double o = 0.0, mine = 0.0;
        for (int j = 0; j < nown; ++j) {
            if ((b.tid & 63) == j) mine = o;
            o += readlane_f64(t, j);
        }
        if (b.tid < nown) cum[b.tid] = mine;
        if (b.tid == 0) cum[nown] = o;
    } else
#endif
    {
        if (b.tid < nown) cum[b.tid] = t;
        blk_sync();
        if (b.tid == 0) {
            double o = 0.0;
            for (int j = 0; j < nown; ++j) {
                const double tj = cum[j];
                cum[j] = o;
                o += tj;
            }
            cum[nown] = o;
        }
    }
    blk_sync();
    const double delta = (double)(2 * n + 64) * 2.220446049250313e-16;
    double *cnt = cum + nown + 1;
    for (int l = b.tid; l < 2 * nq; l += b.nt) {
        const int k = l >> 1;
        double q = 0.0;
#pragma unroll
        for (int u = 0; u < TSFA_ALT_MAXKEYS; ++u)
            if (u == k) q = alt.q[u];
        const double T = (l & 1) ? (q + delta) * S : (q - delta) * S;
        int lo = 0, hi = nown;  // first thread whose offset is >= T: every sample from there on is >= T
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cum[mid] >= T) hi = mid;
            else lo = mid + 1;
        }
        int count = 0;
        if (lo > 0) {
            const int j = lo - 1;   // o_j < T <= o_{j+1} (or j is the last thread): the threads before j lie below T entirely
            const double oj = cum[j];
            double sacc = 0.0;
            count = E * j;
            for (int e = 0; e < E; ++e) {
                const int i = E * j + e;
                if (i < n) {
                    sacc += fabs(xs[i]);
                    count += (oj + sacc < T) ? 1 : 0;
                }
            }
        }
        cnt[l] = (double)count;
    }
    blk_sync();
    unsigned amb = 0;
    for (int k = b.tid; k < nq; k += b.nt) {
        const double ca = cnt[2 * k], cb = cnt[2 * k + 1];
        if (ca == cb) {
            const int idx = ((int)ca < n) ? (int)ca : 0;  // np.argmax of an all-False mask is 0
            altc[8 * k + 7] = (double)(idx + 1) / dn;
        } else {
            amb = 1;
        }
    }
    amb = blk_or16(b, amb);
    blk_sync();
    return amb == 0;
}

// The serial route without the n-double array (TsfaAltPlan::small_w): lane = q, every lane adds the samples in a row
// (the rounding of np.cumsum) and notes the first i with c_i / S >= q.  fl(c / S) is monotone in c, so far below
// q S the answer is no and far above it is yes without dividing; the division decides only within 1e-15 of q S.
template <class BT, class XS>
TSFA_DEV void imq_serial_walk(const BT &b, XS xs, int n, double S, const TsfaAltPlan &alt, double *altc) {
    const double dn = (double)n;
    for (int k = b.tid; k < alt.nq; k += b.nt) {
        double q = 0.0;
#pragma unroll
        for (int u = 0; u < TSFA_ALT_MAXKEYS; ++u)
            if (u == k) q = alt.q[u];
        double res = TSFA_NAN;
        if (S != 0.0) {
            const double lo = q * S * (1.0 - 1e-15), hi = q * S * (1.0 + 1e-15);
            const bool banded = (S > 0.0) && (S < TSFA_INF) && (q > 0.0);   // otherwise every step divides
            double acc = 0.0;
            int idx = -1;
            for (int i = 0; i < n; ++i) {
                acc += fabs(xs[i]);
                if (idx < 0) {
                    const bool yes = banded ? (acc > hi || (acc >= lo && acc / S >= q)) : (acc / S >= q);
                    if (yes) idx = i;
                }
            }
            if (idx < 0) idx = 0;  // np.argmax of an all-False mask is 0
            res = (double)(idx + 1) / dn;
        }
        altc[8 * k + 7] = res;
    }
    blk_sync();
}

// Count-type columns (ratio_beyond_r_sigma, count_above/below(_mean), value_count, range_count, number_crossing_m):
// the host moves them to the front of the spec list (tsfa_prepare_family, hint d = their number) and they are
// evaluated together here.  A wavefront keeps 1024 samples in registers (16 per lane); a predicate then costs one
// v_cmp per register, its count is the popcount of the compare mask (scalar ALU) -- no per-lane accumulators, no
// cross-lane reduction, no second trip to LDS.  Sign changes (number_crossing_m) are bit operations on the masks.
// iw: LDS, >= ncount ints.  Results are integers, identical to the column loop's.
TSFA_DEV bool basic_count_scaled(int calc) {  // count / n instead of the count
    return calc == TSFA_C_RATIO_BEYOND_R_SIGMA || calc == TSFA_C_COUNT_ABOVE || calc == TSFA_C_COUNT_BELOW;
}
template <class BT, class XS>
TSFA_DEV void basic_count_pass(const BT &b, XS xs, int n, const TsfaSpec *specs, int ncount, const BasicStats &st,
                               double *out_row, int *iw) {
    const double mean = st.mean, dn = (double)n;
#if TSFA_GPU
    if constexpr (BlkIsRow<BT>::v) {
        // row form (n <= 256 = 16 samples per lane): a predicate is a compare and an add-with-carry per register, the count one
        // integer DPP tree inside the row; the columns' spec loads and dispatch are shared by the four rows of the wavefront
        (void)iw;
        const int lane = b.tid;
        double xr[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = u * 16 + lane;
            xr[u] = (i < n) ? xs[i] : TSFA_NAN;  // NaN: every ordered comparison below is false
        }
        TsfaSpec nxt = specs[0];
        for (int e = 0; e < ncount; ++e) {
            const TsfaSpec sp = nxt;
            nxt = specs[(e + 1 < ncount) ? e + 1 : e];
            const double p0 = sp.p[0], p1 = sp.p[1];
            int c = 0;
            int ln = lane;
            asm volatile("" : "+v"(ln));
#pragma unroll
            for (int u = 0; u < 16; ++u) asm volatile("" : "+v"(xr[u]));
#define TSFA_ROWCOUNT_LOOP(PRED) \
    _Pragma("unroll") for (int u = 0; u < 16; ++u) { const double x = xr[u]; c += (PRED) ? 1 : 0; }
            switch (sp.calc) {
            case TSFA_C_RATIO_BEYOND_R_SIGMA: { const double thr = p0 * st.std; TSFA_ROWCOUNT_LOOP(fabs(x - mean) > thr) } break;
            case TSFA_C_COUNT_ABOVE_MEAN: TSFA_ROWCOUNT_LOOP(x > mean) break;
            case TSFA_C_COUNT_BELOW_MEAN: TSFA_ROWCOUNT_LOOP(x < mean) break;
            case TSFA_C_COUNT_ABOVE: TSFA_ROWCOUNT_LOOP(x >= p0) break;
            case TSFA_C_COUNT_BELOW: TSFA_ROWCOUNT_LOOP(x <= p0) break;
            case TSFA_C_RANGE_COUNT: TSFA_ROWCOUNT_LOOP(x >= p0 && x < p1) break;
            case TSFA_C_VALUE_COUNT:
                if (p0 != p0) {
#pragma unroll
                    for (int u = 0; u < 16; ++u) { const double x = xr[u]; c += (u * 16 + ln < n && x != x) ? 1 : 0; }
                } else {
                    TSFA_ROWCOUNT_LOOP(x == p0)
                }
                break;
            case TSFA_C_NUMBER_CROSSING_M: {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int i = u * 16 + ln + 1;
                    const double y = xs[(i < n) ? i : (n - 1)];
                    c += (i < n && ((xr[u] > p0) != (y > p0))) ? 1 : 0;
                }
            } break;
            default: break;
            }
#undef TSFA_ROWCOUNT_LOOP
            c = row_sum_i32(c);
            if (lane == 0) out_row[sp.col] = basic_count_scaled(sp.calc) ? (double)c / dn : (double)c;
        }
        return;
    }
    const int lane = b.tid & 63, wave = b.tid >> 6, nwave = b.nt >> 6;
    blk_sync();
    for (int e = b.tid; e < ncount; e += b.nt) iw[e] = 0;
    blk_sync();
    for (int base = wave * 1024; base < n; base += nwave * 1024) {
        double xr[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = base + u * 64 + lane;
            xr[u] = (i < n) ? xs[i] : TSFA_NAN;  // NaN: every ordered comparison below is false
        }
        const double xnext = (base + 1024 < n) ? xs[base + 1024] : TSFA_NAN;
        TsfaSpec nxt = specs[0];
        for (int e = 0; e < ncount; ++e) {
            const TsfaSpec sp = nxt;
            nxt = specs[(e + 1 < ncount) ? e + 1 : e];
            const double p0 = sp.p[0], p1 = sp.p[1];
            int c = 0;
            // keep the compares inside this iteration: hoisted out of the entry loop, their 64-bit masks (and the
            // validity masks) would have to live in scalar registers for the whole loop and get spilled
            int ln = lane;
            asm volatile("" : "+v"(ln));
#pragma unroll
            for (int u = 0; u < 16; ++u) asm volatile("" : "+v"(xr[u]));
#define TSFA_COUNT_LOOP(PRED)                                                         \
    _Pragma("unroll") for (int u = 0; u < 16; ++u) { const double x = xr[u]; c += __popcll(__ballot(PRED)); }
            switch (sp.calc) {
            case TSFA_C_RATIO_BEYOND_R_SIGMA: { const double thr = p0 * st.std; TSFA_COUNT_LOOP(fabs(x - mean) > thr) } break;
            case TSFA_C_COUNT_ABOVE_MEAN: TSFA_COUNT_LOOP(x > mean) break;
            case TSFA_C_COUNT_BELOW_MEAN: TSFA_COUNT_LOOP(x < mean) break;
            case TSFA_C_COUNT_ABOVE: TSFA_COUNT_LOOP(x >= p0) break;
            case TSFA_C_COUNT_BELOW: TSFA_COUNT_LOOP(x <= p0) break;
            case TSFA_C_RANGE_COUNT: TSFA_COUNT_LOOP(x >= p0 && x < p1) break;
            case TSFA_C_VALUE_COUNT:
                if (p0 != p0) {
#pragma unroll
                    for (int u = 0; u < 16; ++u) { const double x = xr[u]; c += __popcll(__ballot(base + u * 64 + ln < n && x != x)); }
                } else {
                    TSFA_COUNT_LOOP(x == p0)
                }
                break;
            case TSFA_C_NUMBER_CROSSING_M: {
                unsigned long long nb = (xnext > p0) ? 1ull : 0ull;  // (x[i + 1] > m) of the lane 63 element
#pragma unroll
                for (int u = 15; u >= 0; --u) {
                    const unsigned long long m = __ballot(xr[u] > p0);
                    const unsigned long long pv = __ballot(base + u * 64 + ln + 1 < n);  // pairs (i, i + 1) inside the series
                    c += __popcll((m ^ ((m >> 1) | (nb << 63))) & pv);
                    nb = m & 1ull;
                }
            } break;
            default: break;
            }
#undef TSFA_COUNT_LOOP
            if (lane == 0) __hip_atomic_fetch_add(&iw[e], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    blk_sync();
    for (int e = b.tid; e < ncount; e += b.nt) {
        const TsfaSpec sp = specs[e];
        const double c = (double)iw[e];
        out_row[sp.col] = basic_count_scaled(sp.calc) ? c / dn : c;
    }
    blk_sync();
#else
    (void)iw; (void)b;
    for (int e = 0; e < ncount; ++e) {
        const TsfaSpec sp = specs[e];
        const double p0 = sp.p[0], p1 = sp.p[1], thr = p0 * st.std;
        double c = 0.0;
        for (int i = 0; i < n; ++i) {
            const double x = xs[i];
            bool p = false;
            switch (sp.calc) {
            case TSFA_C_RATIO_BEYOND_R_SIGMA: p = fabs(x - mean) > thr; break;
            case TSFA_C_COUNT_ABOVE_MEAN: p = x > mean; break;
            case TSFA_C_COUNT_BELOW_MEAN: p = x < mean; break;
            case TSFA_C_COUNT_ABOVE: p = x >= p0; break;
            case TSFA_C_COUNT_BELOW: p = x <= p0; break;
            case TSFA_C_RANGE_COUNT: p = (x >= p0 && x < p1); break;
            case TSFA_C_VALUE_COUNT: p = (p0 != p0) ? (x != x) : (x == p0); break;
            case TSFA_C_NUMBER_CROSSING_M: p = (i + 1 < n) && ((x > p0) != (xs[i + 1] > p0)); break;
            default: break;
            }
            c += p ? 1.0 : 0.0;
        }
        out_row[sp.col] = basic_count_scaled(sp.calc) ? c / dn : c;
    }
#endif
}

// Sum-type columns (autocorrelation, c3, time_reversal_asymmetry_statistic, energy_ratio_by_chunks, cid_ce,
// mean_abs_change, absolute_sum_of_changes, skewness, kurtosis): the host places them behind the count-type ones
// (hint e = their number).  For a series of <= 1024 samples owned by ONE wavefront the samples stay in registers (16
// per lane), the lagged operands of a column are fetched from LDS all at once (16 independent reads in flight instead
// of a read -> multiply -> add round trip per loop iteration) and the lane sums run in the same order as the column
// loop's (identical results).  Other shapes leave these columns to the column loop.
template <class BT, class XS>
TSFA_DEV bool basic_sum_pass(const BT &b, XS xs, int n, const TsfaSpec *specs, int first, int nsum, const BasicStats &st,
                             double *out_row) {
#if TSFA_GPU
    constexpr int LN = BlkLanes<BT>::n;   // 64: one wavefront owns <= 1024 samples; 16: the row form (<= 256)
    if (b.nt != LN || n > 16 * LN || nsum <= 0) return false;
    const int lane0 = b.tid;
    const double mean = st.mean, dn = (double)n;
    double xr[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int i = u * LN + lane0;
        xr[u] = xs[(i < n) ? i : (n - 1)];  // unconditional reads (clamped index): every use below is masked
    }
#define TSFA_LOAD_LAG(Y, LAG)                                                                         \
    double Y[16];                                                                                     \
    _Pragma("unroll") for (int u = 0; u < 16; ++u) {                                                  \
        const int i = u * LN + lane + (LAG);                                                          \
        Y[u] = xs[(i < n) ? i : (n - 1)];                                                             \
    }
    TsfaSpec nxt = specs[first];
    for (int e = 0; e < nsum; ++e) {
        const TsfaSpec sp = nxt;
        nxt = specs[first + ((e + 1 < nsum) ? e + 1 : e)];
        const double p0 = sp.p[0], p1 = sp.p[1];
        double v = TSFA_NAN;
        // nothing of a column is hoisted out of the entry loop (centred copies, masks, addresses: they would be spilled)
#pragma unroll
        for (int u = 0; u < 16; ++u) asm volatile("" : "+v"(xr[u]));
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        switch (sp.calc) {
        case TSFA_C_AUTOCORRELATION: {                                   // fc.py:1919
            const int lag = (int)p0;
            if (n < lag) break;
            TSFA_LOAD_LAG(y, lag)
            double a = 0.0;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const double t = (xr[u] - mean) * (y[u] - mean);
                a += (u * LN + lane < n - lag) ? t : 0.0;
            }
            a = blk_sum(b, a);
            if (!(fabs(st.var) <= 1e-8)) v = a / ((double)(n - lag) * st.var);  // np.isclose(v, 0) -> NaN
        } break;
        case TSFA_C_C3:                                                  // fc.py:1600
        case TSFA_C_TIME_REVERSAL_ASYMMETRY_STATISTIC: {                 // fc.py:1557
            const int lag = (int)p0;
            if (2 * lag >= n) { v = 0.0; break; }
            const int m = n - 2 * lag;
            TSFA_LOAD_LAG(y1, lag)
            TSFA_LOAD_LAG(y2, 2 * lag)
            double a = 0.0;
            if (sp.calc == TSFA_C_C3) {
#pragma unroll
                for (int u = 0; u < 16; ++u) { const double t = y2[u] * y1[u] * xr[u]; a += (u * LN + lane < m) ? t : 0.0; }
            } else {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const double t = y2[u] * y2[u] * y1[u] - y1[u] * xr[u] * xr[u];
                    a += (u * LN + lane < m) ? t : 0.0;
                }
            }
            v = blk_sum(b, a) / (double)m;
        } break;
        case TSFA_C_ENERGY_RATIO_BY_CHUNKS: {                            // fc.py:2226 (np.array_split)
            const int nseg = (int)p0, foc = (int)p1;
            const int q = n / nseg, rem = n % nseg;
            const int lo = foc * q + (foc < rem ? foc : rem);
            const int hi = lo + q + (foc < rem ? 1 : 0);
            double a = 0.0;
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int i = u * LN + lane; a += (i >= lo && i < hi) ? xr[u] * xr[u] : 0.0; }
            a = blk_sum(b, a);
            v = (st.sumsq == 0.0) ? TSFA_NAN : a / st.sumsq;
        } break;
        case TSFA_C_CID_CE: {                                            // fc.py:567
            const bool normalize = (p0 != 0.0);
            if (normalize && st.std == 0.0) { v = 0.0; break; }
            const double sd = st.std;
            TSFA_LOAD_LAG(y, 1)
            double a = 0.0;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const double d = normalize ? ((y[u] - mean) / sd - (xr[u] - mean) / sd) : (y[u] - xr[u]);
                a += (u * LN + lane < n - 1) ? d * d : 0.0;
            }
            v = sqrt(blk_sum(b, a));
        } break;
        case TSFA_C_MEAN_ABS_CHANGE:                                     // fc.py:604
        case TSFA_C_ABSOLUTE_SUM_OF_CHANGES: {                           // fc.py:796
            TSFA_LOAD_LAG(y, 1)
            double a = 0.0;
#pragma unroll
            for (int u = 0; u < 16; ++u) a += (u * LN + lane < n - 1) ? fabs(y[u] - xr[u]) : 0.0;
            a = blk_sum(b, a);
            if (sp.calc == TSFA_C_MEAN_ABS_CHANGE) v = (n > 1) ? a / (double)(n - 1) : TSFA_NAN;
            else v = a;
        } break;
        case TSFA_C_SKEWNESS:                                            // fc.py:749 -> pandas nanops.nanskew
        case TSFA_C_KURTOSIS: {                                          // fc.py:766 -> pandas nanops.nankurt
            const bool skew = (sp.calc == TSFA_C_SKEWNESS);
            double m2 = 0.0, mh = 0.0;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const double a = xr[u] - mean;
                const double a2 = a * a;
                const bool in = (u * LN + lane < n);
                m2 += in ? a2 : 0.0;
                mh += in ? (skew ? a2 * a : a2 * a2) : 0.0;
            }
            m2 = blk_sum(b, m2);
            mh = blk_sum(b, mh);
            if (skew) {
                if (fabs(m2) < 1e-14) m2 = 0.0;
                if (fabs(mh) < 1e-14) mh = 0.0;
                if (n < 3) v = TSFA_NAN;
                else if (m2 == 0.0) v = 0.0;
                else v = (dn * sqrt(dn - 1.0) / (dn - 2.0)) * (mh / pow(m2, 1.5));
            } else if (n >= 4) {
                const double adj = 3.0 * (dn - 1.0) * (dn - 1.0) / ((dn - 2.0) * (dn - 3.0));
                double num = dn * (dn + 1.0) * (dn - 1.0) * mh;
                double den = (dn - 2.0) * (dn - 3.0) * m2 * m2;
                if (fabs(num) < 1e-14) num = 0.0;
                if (fabs(den) < 1e-14) den = 0.0;
                v = (den == 0.0) ? 0.0 : (num / den - adj);
            }
        } break;
        default: break;
        }
        if (lane == 0) out_row[sp.col] = v;
    }
#undef TSFA_LOAD_LAG
    return true;
#else
    (void)b; (void)xs; (void)n; (void)specs; (void)first; (void)nsum; (void)st; (void)out_row;
    return false;
#endif
}

// Evaluate the BASIC specs of one series.
//   xs   : series as float64 in LDS, length n (n >= 1)
//   w    : LDS work array of >= n doubles (chunk aggregates)
//   cum  : LDS, >= n doubles (cumulative |x| for index_mass_quantile); may alias w
//   altc : LDS, >= 8 * TSFA_ALT_CACHE doubles (agg_linear_trend regression cache)
//   iw   : LDS int array of >= max(4*nt, 128) ints
#define TSFA_ALT_CACHE 16
#define TSFA_PEAK_NEAR 10
#define TSFA_DEV_UNUSED
//   times: HBM, the series' timestamps as float64 hours since its first sample (linear_trend_timewise), or null
// PART: 1 = the BASIC family (k_basic), 2 = the TREND family (k_trend: index_mass_quantile, linear_trend(_timewise),
// agg_linear_trend -- the calculators that need a float64 work array of n entries), 3 = both.  Two kernels instead
// of one: each half needs far fewer registers and less LDS than the union, and both are latency-bound (the resident
// wavefronts per CU are what they gain from).
template <int PART, class BT, class XS>
TSFA_DEV void fam_basic_series(const BT &b0, XS xs, int n, const TsfaSpec *specs, int nspecs,
                               double *out_row, double *w, double *cum, double *altc, int *iw, const double *dectab,
                               int peaks_hint, int alt_want_p, const TsfaAltPlan &alt, TsfaSpec *stage,
                               const double *times = nullptr, int n_loop = -1, double *ctx = nullptr,
                               int n_count = 0, int n_sum = 0, double *stats_out = nullptr) {
    const BT &b = b0;
    TSFA_TICKER(tk, 0);
    BasicStats st;
    const int peaks_maxsup = peaks_hint & 0xFFFF;         // tsfa_prepare_family: largest number_peaks support
    const bool want_loc = ((peaks_hint >> 16) & 1) == 0;  // ... and bit 16: no column reads the extrema's positions
    if (PART & 1) {
        basic_stats(b, xs, n, st, want_loc);
        // the series' statistics for the families launched after this one (TSFA_STATS_*: numpy-order mean and variance,
        // extrema): k_entropy_bits / k_entropy, k_ar and k_seq used to recompute them -- 0.7 + 0.3 + 0.1 ms per 100 000 series
        if (stats_out != nullptr && b.tid == 0) {
            stats_out[TSFA_STATS_MEAN] = st.mean; stats_out[TSFA_STATS_VAR] = st.var;
            stats_out[TSFA_STATS_MIN] = st.vmin; stats_out[TSFA_STATS_MAX] = st.vmax;
        }
    } else {
        st.n = n; st.sum = 0.0; st.mean = 0.0; st.var = 0.0; st.std = 0.0; st.vmin = 0.0; st.vmax = 0.0; st.sumsq = 0.0;
        st.first_max = 0; st.last_max = 0; st.first_min = 0; st.last_min = 0; st.cnt_max = 0; st.cnt_min = 0;
    }
    TSFA_TICK(tk, b, 100);
    // n_loop columns go through the column loop; the rest are evaluated by basic_epilogue (lane = column) from ctx
    const int nloop = (n_loop >= 0 && ctx != nullptr) ? n_loop : nspecs;
    if (nloop < nspecs) {
        blk_sync();
        if (b.tid == 0) {
            ctx[TSFA_CTX_SUM] = st.sum; ctx[TSFA_CTX_MEAN] = st.mean; ctx[TSFA_CTX_VAR] = st.var; ctx[TSFA_CTX_STD] = st.std;
            ctx[TSFA_CTX_VMIN] = st.vmin; ctx[TSFA_CTX_VMAX] = st.vmax; ctx[TSFA_CTX_SUMSQ] = st.sumsq;
            ctx[TSFA_CTX_FIRST_MAX] = (double)st.first_max; ctx[TSFA_CTX_LAST_MAX] = (double)st.last_max;
            ctx[TSFA_CTX_FIRST_MIN] = (double)st.first_min; ctx[TSFA_CTX_LAST_MIN] = (double)st.last_min;
            ctx[TSFA_CTX_CNT_MAX] = (double)st.cnt_max; ctx[TSFA_CTX_CNT_MIN] = (double)st.cnt_min;
            for (int k = 0; k < 5; ++k) { ctx[TSFA_CTX_LT + k] = TSFA_NAN; ctx[TSFA_CTX_LTT + k] = TSFA_NAN; }
        }
    }
    const double dn = (double)n;
    const double mean = st.mean;
    bool have_cumsum = false, have_lt = false, have_ltt = false;
    double lt5[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    double ltt5[5] = {TSFA_NAN, TSFA_NAN, TSFA_NAN, TSFA_NAN, TSFA_NAN};
    double imq_sabs = 0.0;

    bool have_peaks = false;
    int peaks_p = 0;  // window length of the sliding maxima currently held in w (number_peaks fast path), 0 = none

    if (PART & 4) {  // k_basic_lite: every column is a closed form of the statistics (the plan has no other kind)
        if (nloop < nspecs) {
            blk_sync();
            basic_epilogue(b, specs, nloop, nspecs, n, ctx, altc, out_row);
        }
        return;
    }
    // the count-type columns at the front of the list (host hint) are evaluated together from registers
    const int ncnt = ((PART & 1) && n_count > 0 && n_count <= nloop) ? n_count : 0;
    if ((PART & 1) && ncnt > 0) basic_count_pass(b, xs, n, specs, ncnt, st, out_row, iw);
    TSFA_TICK(tk, b, 213);
    // ... and behind them the sum-type columns (single-wavefront series of <= 1024 samples; else the column loop)
    int sbeg = ncnt;
    if ((PART & 1) && ncnt == n_count && n_sum > 0 && ncnt + n_sum <= nloop && basic_sum_pass(b, xs, n, specs, ncnt, n_sum, st, out_row))
        sbeg = ncnt + n_sum;
    TSFA_TICK(tk, b, 214);
    TsfaSpec nxt = spec_fetch(b, specs, nspecs, (sbeg < nspecs) ? sbeg : 0, stage);
    for (int s = sbeg; s < nloop; ++s) {
#if TSFA_GPU
        // opaque thread index per column: nothing derived from it is hoisted out of the column loop, kept live across
        // all the other columns and spilled
        int tid_opaque = b0.tid;
        asm volatile("" : "+v"(tid_opaque));
        const BT b = blk_rebind(b0, tid_opaque);
#endif
        TSFA_TICKER(tkc, 0);
        const TsfaSpec sp = nxt;
#if !defined(TSFA_SPEC_LDS)
        nxt = specs[(s + 1 < nloop) ? s + 1 : s];  // scalar load in flight while this column is evaluated
#else
        if (s + 1 < nspecs) nxt = spec_fetch(b, specs, nspecs, s + 1, stage);
#endif
        TSFA_TICK(tkc, b, 210);
        const double p0 = sp.p[0], p1 = sp.p[1], p2 = sp.p[2];
        double v = TSFA_NAN;
        if (PART & 1) {  // k_basic proper
        switch (sp.calc) {
        case TSFA_C_SUM_VALUES: v = st.sum; break;                       // fc.py:371
        case TSFA_C_MEAN: v = st.mean; break;                            // fc.py:677
        case TSFA_C_LENGTH: v = dn; break;                               // fc.py:691
        case TSFA_C_STANDARD_DEVIATION: v = st.std; break;               // fc.py:705
        case TSFA_C_VARIANCE: v = st.var; break;                         // fc.py:735
        case TSFA_C_ROOT_MEAN_SQUARE: v = sqrt(st.sumsq / dn); break;    // fc.py:783
        case TSFA_C_MAXIMUM: v = st.vmax; break;                         // fc.py:2003
        case TSFA_C_ABSOLUTE_MAXIMUM: v = fmax(fabs(st.vmax), fabs(st.vmin)); break;  // fc.py:2017
        case TSFA_C_MINIMUM: v = st.vmin; break;                         // fc.py:2031
        case TSFA_C_ABS_ENERGY: v = st.sumsq; break;                     // fc.py:548
        case TSFA_C_VARIATION_COEFFICIENT:                               // fc.py:718
            v = (mean == 0.0) ? TSFA_NAN : st.std / mean;
            break;
        case TSFA_C_VAR_GT_STD: v = (st.var > sqrt(st.var)) ? 1.0 : 0.0; break;  // fc.py:239
        case TSFA_C_LARGE_STD:                                           // fc.py:273
            v = (st.std > p0 * (st.vmax - st.vmin)) ? 1.0 : 0.0;
            break;
        case TSFA_C_RATIO_BEYOND_R_SIGMA: {                              // fc.py:256
            const double thr = p0 * st.std;
            double c = 0.0;
            for (int i = b.tid; i < n; i += b.nt) c += (fabs(xs[i] - mean) > thr) ? 1.0 : 0.0;
            v = blk_sum(b, c) / dn;
        } break;
        case TSFA_C_SKEWNESS: {                                          // fc.py:749 -> pandas nanops.nanskew
            double m2 = 0.0, m3 = 0.0;
            for (int i = b.tid; i < n; i += b.nt) {
                const double a = xs[i] - mean;
                const double a2 = a * a;
                m2 += a2;
                m3 += a2 * a;
            }
            m2 = blk_sum(b, m2);
            m3 = blk_sum(b, m3);
            if (fabs(m2) < 1e-14) m2 = 0.0;
            if (fabs(m3) < 1e-14) m3 = 0.0;
            if (n < 3) v = TSFA_NAN;
            else if (m2 == 0.0) v = 0.0;
            else v = (dn * sqrt(dn - 1.0) / (dn - 2.0)) * (m3 / pow(m2, 1.5));
        } break;
        case TSFA_C_KURTOSIS: {                                          // fc.py:766 -> pandas nanops.nankurt
            double m2 = 0.0, m4 = 0.0;
            for (int i = b.tid; i < n; i += b.nt) {
                const double a = xs[i] - mean;
                const double a2 = a * a;
                m2 += a2;
                m4 += a2 * a2;
            }
            m2 = blk_sum(b, m2);
            m4 = blk_sum(b, m4);
            if (n < 4) {
                v = TSFA_NAN;
            } else {
                const double adj = 3.0 * (dn - 1.0) * (dn - 1.0) / ((dn - 2.0) * (dn - 3.0));
                double num = dn * (dn + 1.0) * (dn - 1.0) * m4;
                double den = (dn - 2.0) * (dn - 3.0) * m2 * m2;
                if (fabs(num) < 1e-14) num = 0.0;
                if (fabs(den) < 1e-14) den = 0.0;
                v = (den == 0.0) ? 0.0 : (num / den - adj);
            }
        } break;
        case TSFA_C_MEAN_ABS_CHANGE:                                     // fc.py:604
        case TSFA_C_ABSOLUTE_SUM_OF_CHANGES: {                           // fc.py:796
            double a = 0.0;
            for (int i = b.tid; i < n - 1; i += b.nt) a += fabs(xs[i + 1] - xs[i]);
            a = blk_sum(b, a);
            if (sp.calc == TSFA_C_MEAN_ABS_CHANGE) v = (n > 1) ? a / (double)(n - 1) : TSFA_NAN;
            else v = a;
        } break;
        case TSFA_C_MEAN_CHANGE:                                         // fc.py:624
            v = (n > 1) ? (xs[n - 1] - xs[0]) / (double)(n - 1) : TSFA_NAN;
            break;
        case TSFA_C_MEAN_SECOND_DERIVATIVE_CENTRAL:                      // fc.py:644
            v = (n > 2) ? (xs[n - 1] - xs[n - 2] - xs[1] + xs[0]) / (double)(2 * (n - 2)) : TSFA_NAN;
            break;
        case TSFA_C_CID_CE: {                                            // fc.py:567
            const bool normalize = (p0 != 0.0);
            if (normalize && st.std == 0.0) {
                v = 0.0;
                break;
            }
            const double sd = st.std;
            double a = 0.0;
            for (int i = b.tid; i < n - 1; i += b.nt) {
                double d;
                if (normalize) d = (xs[i + 1] - mean) / sd - (xs[i] - mean) / sd;
                else d = xs[i + 1] - xs[i];
                a += d * d;
            }
            v = sqrt(blk_sum(b, a));
        } break;
        case TSFA_C_COUNT_ABOVE_MEAN:                                    // fc.py:843
        case TSFA_C_COUNT_BELOW_MEAN: {                                  // fc.py:857
            const bool above = (sp.calc == TSFA_C_COUNT_ABOVE_MEAN);
            double c = 0.0;
            for (int i = b.tid; i < n; i += b.nt) c += (above ? (xs[i] > mean) : (xs[i] < mean)) ? 1.0 : 0.0;
            v = blk_sum(b, c);
        } break;
        case TSFA_C_COUNT_ABOVE:                                         // fc.py:2309
        case TSFA_C_COUNT_BELOW: {                                       // fc.py:2325
            const bool above = (sp.calc == TSFA_C_COUNT_ABOVE);
            double c = 0.0;
            for (int i = b.tid; i < n; i += b.nt) c += (above ? (xs[i] >= p0) : (xs[i] <= p0)) ? 1.0 : 0.0;
            v = blk_sum(b, c) / dn;
        } break;
        case TSFA_C_VALUE_COUNT: {                                       // fc.py:2044
            double c = 0.0;
            for (int i = b.tid; i < n; i += b.nt) c += (p0 != p0 ? (xs[i] != xs[i]) : (xs[i] == p0)) ? 1.0 : 0.0;
            v = blk_sum(b, c);
        } break;
        case TSFA_C_RANGE_COUNT: {                                       // fc.py:2065
            double c = 0.0;
            for (int i = b.tid; i < n; i += b.nt) c += (xs[i] >= p0 && xs[i] < p1) ? 1.0 : 0.0;
            v = blk_sum(b, c);
        } break;
        case TSFA_C_NUMBER_CROSSING_M: {                                 // fc.py:1980
            double c = 0.0;
            for (int i = b.tid; i < n - 1; i += b.nt) c += ((xs[i] > p0) != (xs[i + 1] > p0)) ? 1.0 : 0.0;
            v = blk_sum(b, c);
        } break;
        case TSFA_C_FIRST_LOCATION_OF_MAXIMUM: v = (double)st.first_max / dn; break;              // fc.py:886
        case TSFA_C_LAST_LOCATION_OF_MAXIMUM: v = 1.0 - (double)(n - 1 - st.last_max) / dn; break; // fc.py:871
        case TSFA_C_FIRST_LOCATION_OF_MINIMUM: v = (double)st.first_min / dn; break;              // fc.py:917
        case TSFA_C_LAST_LOCATION_OF_MINIMUM: v = 1.0 - (double)(n - 1 - st.last_min) / dn; break; // fc.py:902
        case TSFA_C_HAS_DUPLICATE_MAX: v = (st.cnt_max >= 2) ? 1.0 : 0.0; break;                  // fc.py:325
        case TSFA_C_HAS_DUPLICATE_MIN: v = (st.cnt_min >= 2) ? 1.0 : 0.0; break;                  // fc.py:340
        case TSFA_C_LONGEST_STRIKE_ABOVE_MEAN:                           // fc.py:828
            v = blk_longest_run(b, n, [=](int i) { return xs[i] > mean; }, iw);
            break;
        case TSFA_C_LONGEST_STRIKE_BELOW_MEAN:                           // fc.py:813
            v = blk_longest_run(b, n, [=](int i) { return xs[i] < mean; }, iw);
            break;
        case TSFA_C_NUMBER_PEAKS: {                                      // fc.py:1235
            // x[i] is a peak of support s iff no sample within s positions on either side is >= x[i], i.e. iff the
            // distances L[i], R[i] to the nearest such sample both exceed s.  One pass finds L and R up to
            // TSFA_PEAK_NEAR for every sample; the few samples still unblocked (1 in 2 * NEAR + 1 on i.i.d. data)
            // are compacted and scanned on to the largest support of the plan.  All supports then read (L, R).
            const int sup = (int)p0;
            if (sup < 1) { v = 0.0; break; }
#if TSFA_GPU
            if (n <= 16 * b.nt) {
                if (n <= 2 * sup) { v = 0.0; break; }
                // Sliding-window maxima by doubling: M_p[i] = max(x[i .. i + p - 1]), M_2p[i] = max(M_p[i], M_p[i + p]).
                // With p the largest power of two <= s, the s samples left of i are covered by M_p[i - s] and
                // M_p[i - p], the s samples right of it by M_p[i + 1] and M_p[i + 1 + s - p]: a peak test is four
                // reads, two maxima, two compares -- instead of 2 s compares (or the distance scan below).  The
                // maxima are kept in w in the input precision (exact) and doubled in place as the supports grow
                // (every thread holds its <= 16 new values in registers across the barrier).
                typedef typename XS::elem ST;
                ST *mb = (ST *)(void *)w;
                int tp = 1;
                while (2 * tp <= sup) tp *= 2;
                if (peaks_p == 0 || peaks_p > tp) peaks_p = 1;
                have_peaks = false;   // w no longer holds the distance codes ...
                have_cumsum = false;  // ... nor the cumulative sums
                while (peaks_p < tp) {
                    const ST *src = (peaks_p == 1) ? xs.p : (const ST *)mb;
                    ST r[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const int i = u * b.nt + b.tid;
                        const int i0 = (i < n) ? i : (n - 1), i1 = (i + peaks_p < n) ? (i + peaks_p) : (n - 1);
                        const ST a0 = src[i0], a1 = src[i1];
                        r[u] = (a0 > a1) ? a0 : a1;
                    }
                    blk_sync();
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const int i = u * b.nt + b.tid;
                        if (i < n) mb[i] = r[u];
                    }
                    blk_sync();
                    peaks_p *= 2;
                }
                const ST *src = (tp == 1) ? xs.p : (const ST *)mb;
                // all reads of all 16 positions are issued unconditionally (clamped index, bitwise predicate): a
                // short-circuit here turns into a chain of dependent LDS round trips
                int ci = 0;
#pragma unroll 1
                for (int h = 0; h < 16; h += 8) {  // eight positions at a time: 40 reads in flight, no spills
                    ST xi[8], l0[8], l1[8], r0[8], r1[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = (h + u) * b.nt + b.tid;
                        const int ic = ((i >= sup) & (i < n - sup)) ? i : sup;  // any in-range position (n > 2 sup here)
                        xi[u] = xs.p[ic];
                        l0[u] = src[ic - sup];
                        l1[u] = src[ic - tp];
                        r0[u] = src[ic + 1];
                        r1[u] = src[ic + 1 + sup - tp];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = (h + u) * b.nt + b.tid;
                        const ST lm = (l0[u] > l1[u]) ? l0[u] : l1[u], rm = (r0[u] > r1[u]) ? r0[u] : r1[u];
                        ci += (int)(i >= sup) & (int)(i < n - sup) & (int)(xi[u] > lm) & (int)(xi[u] > rm);
                    }
                }
                v = blk_sum(b, (double)ci);
                break;
            }
#endif
            if (sup > 254) {  // beyond the 8-bit distance code: direct evaluation
                double c = 0.0;
                for (int i = sup + b.tid; i < n - sup; i += b.nt) {
                    const double xi = xs[i];
                    bool pk = true;
                    for (int k = 1; k <= sup && pk; ++k) pk = (xi > xs[i - k]) && (xi > xs[i + k]);
                    c += pk ? 1.0 : 0.0;
                }
                v = blk_sum(b, c);
                break;
            }
            unsigned short *lr = (unsigned short *)w;  // L | R << 8, distances capped at 255 (= "none found")
            TSFA_TICKER(tkp, 0);
            if (!have_peaks) {
                const int maxsup = (peaks_maxsup > 1) ? peaks_maxsup : 1;  // tsfa_prepare_family (host)
                have_cumsum = false;  // w aliases cum
                const int near = (maxsup < TSFA_PEAK_NEAR) ? maxsup : TSFA_PEAK_NEAR;
                int *cand = iw;       // candidate indices (<= 4 * nt or 256 of them; the rest is scanned in place)
                const int cand_cap = (4 * b.nt > 256) ? 4 * b.nt : 256;
                blk_sync();
                int ncand = 0;
                for (int i0 = 0; i0 < n; i0 += b.nt) {
                    const int i = i0 + b.tid;
                    int L = 255, R = 255;
                    if (i < n) {
                        // all 2 * TSFA_PEAK_NEAR neighbours are fetched up front (independent LDS reads in flight
                        // together) and the nearest blocker on either side is picked by a branch-free descending
                        // scan, instead of a data-dependent loop that waits for a read in every iteration
                        const double xi = xs[i];
                        double xl[TSFA_PEAK_NEAR], xr[TSFA_PEAK_NEAR];
#pragma unroll
                        for (int k = 1; k <= TSFA_PEAK_NEAR; ++k) {
                            xl[k - 1] = (k <= near && i - k >= 0) ? xs[i - k] : 0.0;
                            xr[k - 1] = (k <= near && i + k < n) ? xs[i + k] : 0.0;
                        }
#pragma unroll
                        for (int k = TSFA_PEAK_NEAR; k >= 1; --k) {  // a neighbour outside the series never blocks
                            if (k <= near && i - k >= 0 && !(xi > xl[k - 1])) L = k;
                            if (k <= near && i + k < n && !(xi > xr[k - 1])) R = k;
                        }
                    }
                    const bool open = (i < n) && (maxsup > near) && (L == 255) && (R == 255);
                    int tot;
                    const int pos = ncand + blk_excl_count(b, open, &tot);
                    bool parked = false;
                    if (open && pos < cand_cap) { cand[pos] = i; parked = true; }
                    if (open && !parked) {  // candidate list full: finish this sample here
                        const double xi = xs[i];
                        for (int k = near + 1; k <= maxsup && (L == 255 || R == 255); ++k) {
                            if (L == 255 && i - k >= 0 && !(xi > xs[i - k])) L = k;
                            if (R == 255 && i + k < n && !(xi > xs[i + k])) R = k;
                        }
                    }
                    if (i < n) lr[i] = (unsigned short)(L | (R << 8));
                    ncand += tot;
                }
                if (ncand > cand_cap) ncand = cand_cap;
                blk_sync();
                TSFA_TICK(tkp, b, 225);
                for (int c = b.tid; c < ncand; c += b.nt) {
                    const int i = cand[c];
                    const double xi = xs[i];
                    int L = 255, R = 255;
                    for (int k = near + 1; k <= maxsup && (L == 255 || R == 255); ++k) {
                        if (L == 255 && i - k >= 0 && !(xi > xs[i - k])) L = k;
                        if (R == 255 && i + k < n && !(xi > xs[i + k])) R = k;
                    }
                    lr[i] = (unsigned short)(L | (R << 8));
                }
                blk_sync();
                have_peaks = true;
                TSFA_TICK(tkp, b, 226);
            }
            double c = 0.0;
            for (int i = sup + b.tid; i < n - sup; i += b.nt) {
                const int e = lr[i];
                c += ((e & 255) > sup && (e >> 8) > sup) ? 1.0 : 0.0;
            }
            v = blk_sum(b, c);
        } break;
        case TSFA_C_ENERGY_RATIO_BY_CHUNKS: {                            // fc.py:2226 (np.array_split)
            const int nseg = (int)p0, foc = (int)p1;
            const int q = n / nseg, rem = n % nseg;
            const int lo = foc * q + (foc < rem ? foc : rem);
            const int hi = lo + q + (foc < rem ? 1 : 0);
            double a = 0.0;
            for (int i = lo + b.tid; i < hi; i += b.nt) a += xs[i] * xs[i];
            a = blk_sum(b, a);
            v = (st.sumsq == 0.0) ? TSFA_NAN : a / st.sumsq;
        } break;
        case TSFA_C_C3: {                                                // fc.py:1600
            const int lag = (int)p0;
            if (2 * lag >= n) {
                v = 0.0;
                break;
            }
            const int m = n - 2 * lag;
            double a = 0.0;
            for (int i = b.tid; i < m; i += b.nt) a += xs[i + 2 * lag] * xs[i + lag] * xs[i];
            v = blk_sum(b, a) / (double)m;
        } break;
        case TSFA_C_TIME_REVERSAL_ASYMMETRY_STATISTIC: {                 // fc.py:1557
            const int lag = (int)p0;
            if (2 * lag >= n) {
                v = 0.0;
                break;
            }
            const int m = n - 2 * lag;
            double a = 0.0;
            for (int i = b.tid; i < m; i += b.nt) {
                const double x0 = xs[i], x1 = xs[i + lag], x2 = xs[i + 2 * lag];
                a += x2 * x2 * x1 - x1 * x0 * x0;
            }
            v = blk_sum(b, a) / (double)m;
        } break;
        case TSFA_C_AUTOCORRELATION: {                                   // fc.py:1919
            const int lag = (int)p0;
            if (n < lag) {
                v = TSFA_NAN;
                break;
            }
            double a = 0.0;
            for (int i = b.tid; i < n - lag; i += b.nt) a += (xs[i] - mean) * (xs[i + lag] - mean);
            a = blk_sum(b, a);
            if (fabs(st.var) <= 1e-8) v = TSFA_NAN;  // np.isclose(v, 0)
            else v = a / ((double)(n - lag) * st.var);
        } break;
        case TSFA_C_BINNED_ENTROPY:                                      // fc.py:1666
            v = blk_binned_entropy(b, n, [=](int i) { return xs[i]; }, (int)p0, st.vmin, st.vmax, iw, BlkIsRow<BT>::v ? 64 : 256);   // (64 = TSFA_ROW_BINS, tsfa_layout.h)
            break;
        case TSFA_C_BENFORD_CORRELATION: {                               // fc.py:2341
            blk_sync();
            for (int k = b.tid; k < 16; k += b.nt) iw[k] = 0;
            blk_sync();
            for (int i = b.tid; i < n; i += b.nt) {
                const int d = tsfa_leading_decimal_digit(fabs(xs[i]), dectab);
#if TSFA_GPU
                atomicAdd(&iw[d], 1);
#else
                iw[d] += 1;
#endif
            }
            blk_sync();
            double r = TSFA_NAN;
            if (b.tid == 0) {  // np.corrcoef(benford, data)[0, 1]
                // np.log10(1 + 1 / d), d = 1 .. 9, as numpy returns them (fc.py:2356): nine float64 logarithms on one lane
                // were ~1400 instructions per series in an issue-bound kernel
                const double bd[9] = {0.3010299956639812, 0.17609125905568124, 0.12493873660829993, 0.09691001300805642, 0.07918124604762482, 0.06694678963061322, 0.05799194697768673, 0.05115252244738129, 0.04575749056067514};
                double dd[9], mb = 0.0, md = 0.0;
                for (int k = 0; k < 9; ++k) {
                    dd[k] = (double)iw[k + 1] / dn;
                    mb += bd[k];
                    md += dd[k];
                }
                mb /= 9.0;
                md /= 9.0;
                double sbb = 0.0, sdd = 0.0, sbd = 0.0;
                for (int k = 0; k < 9; ++k) {
                    sbb += (bd[k] - mb) * (bd[k] - mb);
                    sdd += (dd[k] - md) * (dd[k] - md);
                    sbd += (bd[k] - mb) * (dd[k] - md);
                }
                // np.corrcoef: c / sqrt(d_i) / sqrt(d_j) on the ddof=1 covariance, clipped to [-1, 1]
                const double c01 = sbd / 8.0, c00 = sbb / 8.0, c11 = sdd / 8.0;
                r = c01 / sqrt(c00) / sqrt(c11);
                if (r > 1.0) r = 1.0;
                if (r < -1.0) r = -1.0;
            }
            v = blk_bcast0(b, r);
        } break;
        case TSFA_C_QUERY_SIMILARITY_COUNT:                              // fc.py:2475 with query=None
            v = TSFA_NAN;
            break;
        default: break;
        }
        }
        if (PART & 2) {  // the trend calculators (k_trend): cumulative sums, chunk aggregates, regressions
        switch (sp.calc) {
        case TSFA_C_INDEX_MASS_QUANTILE: {                               // fc.py:1275
            const bool imq_indexed = (alt.nq > 0 && p2 == 1.0);
            if (imq_indexed && (int)p1 < 128) {  // evaluated by the plan's first index_mass_quantile column
                v = altc[8 * ((int)p1 & 127) + 7];
                break;
            }
            bool imq_decided = false;
            if (!have_cumsum) {
                have_peaks = false;  // cum may alias the peak distances
                peaks_p = 0;
                imq_sabs = np_sum(b, n, [=](int i) { return fabs(xs[i]); });
                blk_sync();
                // all q of the plan from banded prefix sums where the rounding of np.cumsum cannot matter (imq_banded)
                if (imq_indexed) imq_decided = imq_banded(b, xs, n, imq_sabs, alt, cum, altc);
            }
            if (!imq_decided && imq_indexed && alt.small_w && !have_cumsum) {
                imq_serial_walk(b, xs, n, imq_sabs, alt, altc);   // no n-double array in this launch
                imq_decided = true;
            }
            if (imq_decided) {
                v = altc[8 * ((int)p1 & 127) + 7];
                break;
            }
            if (!have_cumsum) {
                // np.cumsum is a serial accumulation: one lane builds it once (in numpy's order, so that the >= q
                // comparison is bit-identical), every q then scans it in parallel
                if (b.tid == 0) {
                    double acc = 0.0;
                    int i = 0;
                    for (; i + 16 <= n; i += 16) {  // loads batched ahead of the dependent add chain
                        double v[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) v[u] = fabs(xs[i + u]);
#pragma unroll
                        for (int u = 0; u < 16; ++u) { acc += v[u]; cum[i + u] = acc; }
                    }
                    for (; i < n; ++i) {
                        acc += fabs(xs[i]);
                        cum[i] = acc;
                    }
                }
                blk_sync();
                have_cumsum = true;
            }
            if (imq_indexed) {
                // lane = q: cum / S is non-decreasing (a correctly rounded division is monotone), so the first index
                // with cum[i] / S >= q is found by bisection with the reference's own expression -- ~10 dependent
                // divisions for ALL q together instead of n / 64 divisions per lane for every q
                blk_sync();
                for (int k = b.tid; k < alt.nq; k += b.nt) {
                    double q = 0.0;
#pragma unroll
                    for (int u = 0; u < TSFA_ALT_MAXKEYS; ++u)
                        if (u == k) q = alt.q[u];
                    double res = TSFA_NAN;
                    if (imq_sabs != 0.0) {
                        int lo = 0, hi = n;  // first i in [0, n) with the predicate true, n if none
                        while (lo < hi) {
                            const int mid = (lo + hi) >> 1;
                            if (cum[mid] / imq_sabs >= q) hi = mid;
                            else lo = mid + 1;
                        }
                        const int idx = (lo < n) ? lo : 0;  // np.argmax of an all-False mask is 0
                        res = (double)(idx + 1) / dn;
                    }
                    altc[8 * k + 7] = res;
                }
                blk_sync();
                v = altc[8 * ((int)p1 & 127) + 7];
                break;
            }
            if (imq_sabs == 0.0) { v = TSFA_NAN; break; }
            double first = (double)n;
            for (int i = b.tid; i < n; i += b.nt)
                if (cum[i] / imq_sabs >= p0) { first = (double)i; break; }
            first = blk_min(b, first);
            const int idx = (first < (double)n) ? (int)first : 0;  // np.argmax of an all-False mask is 0
            v = (double)(idx + 1) / dn;
        } break;
        case TSFA_C_LINEAR_TREND: {                                      // fc.py:1343
            if (!have_lt) {  // one regression serves all five attributes
                blk_linregress_index(b, n, [=](int i) { return xs[i]; }, lt5);
                have_lt = true;
                if (nloop < nspecs && b.tid == 0)
                    for (int k = 0; k < 5; ++k) ctx[TSFA_CTX_LT + k] = lt5[k];
            }
            v = lt5[0];
#pragma unroll
            for (int k = 1; k < 5; ++k)
                if (k == (int)p0) v = lt5[k];
        } break;
        case TSFA_C_LINEAR_TREND_TIMEWISE: {                             // fc.py:2274
            if (!have_ltt) {
                // hours since the first stamp OF THIS SERIES: a no-op (x - 0.0) for a whole series, the rebase for a
                // window view into a longer one (tsfa_extract_windows)
                const double t_first = (times != nullptr) ? times[0] : 0.0;
                if (times != nullptr && n >= 2)
                    blk_linregress_xy(b, n, [=](int i) { return times[i] - t_first; }, [=](int i) { return xs[i]; }, ltt5);
                have_ltt = true;
                if (nloop < nspecs && b.tid == 0)
                    for (int k = 0; k < 5; ++k) ctx[TSFA_CTX_LTT + k] = ltt5[k];
            }
            v = ltt5[0];
#pragma unroll
            for (int k = 1; k < 5; ++k)
                if (k == (int)p0) v = ltt5[k];
        } break;
        case TSFA_C_AGG_LINEAR_TREND: {                                  // fc.py:2171
            const int attr = (int)p0, cl = (int)p1, agg = (int)p2;
            if (alt.nkeys > 0) {  // all regressions of the plan are computed by the first column
                if (((int)sp.p[3]) >= 128) {
                    have_cumsum = false;  // w may alias cum ...
                    have_peaks = false;   // ... and holds the peak distances
                    peaks_p = 0;
                    alt_fill_all(b, xs, n, alt, w, (double *)(void *)iw, altc);
                }
                v = altc[8 * (((int)sp.p[3]) & 127) + 2 + attr];
                break;
            }
            if (cl >= n) {
                v = TSFA_NAN;
                break;
            }
            // the regression of one (f_agg, chunk_len) pair serves all its attr columns: the host (tsfa_prepare_family)
            // assigned every column its slot of the LDS cache and marked the column that fills it
            const int slot = ((int)sp.p[3]) & 63;
            if (((int)sp.p[3]) >= 64) {
                const int m = (n + cl - 1) / cl;
                have_cumsum = false;  // w may alias cum ...
                have_peaks = false;   // ... and holds the peak distances
                peaks_p = 0;
                blk_sync();
                for (int c = b.tid; c < m; c += b.nt) {  // fc.py:176 _aggregate_on_chunks
                    const int lo = c * cl;
                    const int hi = (lo + cl < n) ? lo + cl : n;
                    double r;
                    if (agg == TSFA_AGG_MAX) {
                        r = xs[lo];
                        for (int i = lo + 1; i < hi; ++i) r = fmax(r, xs[i]);
                    } else if (agg == TSFA_AGG_MIN) {
                        r = xs[lo];
                        for (int i = lo + 1; i < hi; ++i) r = fmin(r, xs[i]);
                    } else {
                        const double cm = np_leaf_sum(lo, hi - lo, [=](int i) { return xs[i]; }) / (double)(hi - lo);
                        if (agg == TSFA_AGG_MEAN) r = cm;
                        else r = np_leaf_sum(lo, hi - lo, [=](int i) { const double d = xs[i] - cm; return d * d; }) /
                                 (double)(hi - lo);
                    }
                    w[c] = r;
                }
                blk_sync();
                double o5[5];
                const double *wc = w;
                blk_linregress_index(b, m, [=](int i) { return wc[i]; }, o5, alt_want_p != 0);
                blk_sync();
                if (b.tid == 0) {
                    for (int k = 0; k < 5; ++k) altc[8 * slot + 2 + k] = o5[k];
                }
                blk_sync();
            }
            v = altc[8 * slot + 2 + attr];
        } break;
        default: break;
        }
        }
        TSFA_TICK(tkc, b, 211);
        if (b.tid == 0) out_row[sp.col] = v;
        TSFA_TICK(tkc, b, 212);
        TSFA_TICK(tk, b, sp.calc);
    }
    if (nloop < nspecs) {
        blk_sync();
        basic_epilogue(b, specs, nloop, nspecs, n, ctx, altc, out_row);
    }
}

#endif
