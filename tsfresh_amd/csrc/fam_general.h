// Family GENERAL: parameter values beyond the tables of the tuned kernels (round 6).
//
// The reference accepts any parameter value; the family kernels hold fixed-size tables for the grids of the settings objects and
// a margin around them (one matrix entry per lane, one lag per lane, 64 quantile bins on the lanes of a wavefront, a symbol per
// byte, a bit per CWT width ...).  A `from_columns()` settings dict of an old feature matrix can ask for more -- maxlag 200, lag
// 100, r = 100 bins, a degree-5 drift polynomial, 1000 Lempel-Ziv bins, 30 CWT widths.  Until round 6 such a plan was refused
// (TSFA_ERR_UNSUPPORTED).  Now the CALCULATORS of a plan that hold such a value (all columns of that calculator: the combiners
// of the reference look at the largest value of the dict, fc.py:428, :470) are served here:
//
//   agg_autocorrelation (fc.py:387)          any maxlag      lag = thread, plain sums; median by a sort in the slot
//   partial_autocorrelation (fc.py:440)      any lag         Levinson-Durbin, the inner products dealt over the threads
//   friedrich_coefficients (fc.py:2082),
//   max_langevin_fixed_point (fc.py:2134)    any m, r        the tuned kernel's exact bin sums (fam_sort.h: friedrich_bin_means),
//                                                            np.polyfit as the double-double eigen-solve of fam_langevin_dd.h at
//                                                            run-time size; roots of a degree > 3 by Aberth's iteration
//   lempel_ziv_complexity (fc.py:1825)       any bins        32-bit symbols, one open-addressing table of (node, symbol) keys
//   number_cwt_peaks (fc.py:1320)            any n           fam_general_cwt.h
//   query_similarity_count (fc.py:2475)      a query         lane = window; the query lives in the plan's float64 pool
//
// One wavefront per series, the series read from HBM / L2 where it lies, every working array in a slot of HBM scratch per
// resident workgroup: no length limit, no table limit, and no claim on speed -- a slow path that returns the reference's numbers
// where the library used to return an error.  (ar_coefficient with k > 31 does not come here: the double-double second pass of
// the AR family takes any order, fam_ar_dd.h.)
#ifndef TSFA_FAM_GENERAL_H
#define TSFA_FAM_GENERAL_H

#include "tsfa_common.h"
#include "tsfa_dd.h"
#include "fam_sort.h"
#include "fam_cwt.h"

#if TSFA_GPU
#define TSFA_GEN_HD __host__ __device__ inline
#else
#define TSFA_GEN_HD inline
#endif
TSFA_GEN_HD size_t gen_pow2(size_t n) {
    size_t p = 1;
    while (p < n) p <<= 1;
    return p;
}

// a workgroup's slot of HBM scratch
struct GenSlot {
    double *srt;            // next_pow2(maxn): sorted copy (median of the autocorrelations, quantile edges of the Langevin bins)
    double *av;             // maxn + 2: autocorrelations / autocovariances
    double *lev;            // 3 (L + 2): Levinson-Durbin columns and the partial autocorrelations
    double *fw;             // friedrich_bin_means scratch + the bin means
    double *pf;             // polyfit: 4 c^2 + 12 c doubles (c = m + 1), roots 4 c
    unsigned int *seq;      // maxn symbols
    unsigned long long *ht; // Lempel-Ziv phrase table, 2 next_pow2(maxn) slots
    // number_cwt_peaks: the transform [W x maxn], the noise floor per column, the taps of one width, a byte per cell for the
    // relative maxima, this row's maxima + the line each one joins, the live ridge lines (last column / last row / points / gap)
    double *cw, *noise, *taps;
    unsigned char *mask;
    int *maxc, *choice, *l_col, *l_row, *l_len, *l_gap, *snap;
    TSFA_GEN_HD size_t carve(double *base, int maxn, const TsfaGenPlan &g) {
        size_t o = 0;
        const size_t np2 = gen_pow2(maxn < 2 ? 2 : (size_t)maxn);
        auto take = [&](size_t doubles) { double *p = base ? base + o : nullptr; o += (doubles + 1) & ~(size_t)1; return p; };
        const bool need_sorted = g.acf_maxlag >= 0 || g.fr_maxr > 0;
        srt = take(need_sorted ? np2 : 0);
        av = take((g.acf_maxlag >= 0 || g.pacf_maxlag >= 0) ? (size_t)maxn + 2 : 0);
        {
            size_t L = (g.pacf_maxlag >= 0) ? (size_t)((g.pacf_maxlag < maxn / 2) ? g.pacf_maxlag : maxn / 2) : 0;
            lev = take((g.pacf_maxlag >= 0) ? 3 * (L + 2) : 0);
        }
        const size_t r = (size_t)g.fr_maxr, c = (size_t)g.fr_maxm + 1;
        fw = take(r > 0 ? 6 * r + 16 + r * 2 + 2 * r + 8 : 0);
        pf = take(r > 0 ? 4 * c * c + 16 * c + 8 : 0);
        seq = (unsigned int *)(void *)take(g.lz ? ((size_t)maxn + 2) / 2 + 1 : 0);
        ht = (unsigned long long *)(void *)take(g.lz ? 2 * np2 : 0);
        {
            const size_t W = (size_t)g.cwt_maxw, N = (size_t)maxn, cap = 2 * N + 16;
            cw = take(W * N);
            noise = take(W ? N : 0);
            taps = take(W ? ((10 * W < N) ? 10 * W : N) + 8 : 0);
            mask = (unsigned char *)(void *)take(W ? (W * N) / 8 + 2 : 0);
            maxc = (int *)(void *)take(W ? N / 2 + 2 : 0);
            choice = (int *)(void *)take(W ? N / 2 + 2 : 0);
            l_col = (int *)(void *)take(W ? cap / 2 + 1 : 0);
            l_row = (int *)(void *)take(W ? cap / 2 + 1 : 0);
            l_len = (int *)(void *)take(W ? cap / 2 + 1 : 0);
            l_gap = (int *)(void *)take(W ? cap / 2 + 1 : 0);
            snap = (int *)(void *)take(W ? cap / 2 + 1 : 0);
        }
        return o;
    }
};

// ---------------------------------------------------------------------------------------------------------------
// np.polyfit(x, y, deg = m) for k points, any m >= 1: fam_langevin_dd.h's polyfit_svd_dd at run-time size (serial; one thread).
// work: 4 c^2 + 12 c doubles, c = m + 1.  coef[0 .. m], highest power first.
// ---------------------------------------------------------------------------------------------------------------
template <class XF, class YF>
TSFA_DEV void polyfit_svd_dd_n(XF xm, YF ym, int k, int m, double *coef, double *work) {
    const int c = m + 1;
    dd *G = (dd *)(void *)work;           // c x c, by exponent pair
    dd *V = G + c * c;                    // eigenvectors (columns)
    dd *bv = V + c * c;                   // c
    dd *sol = bv + c;                     // c
    double *sc = (double *)(void *)(sol + c);   // c: column scales, by exponent
    double *a = sc + c;                   // c: a row of the scaled design
    for (int e = 0; e < c; ++e) sc[e] = 0.0;
    for (int i = 0; i < k; ++i) {         // scale = sqrt((lhs * lhs).sum(axis = 0)): rows added in order
        const double x = xm(i);
        double pw = 1.0;
        for (int e = 0; e < c; ++e) { sc[e] += pw * pw; pw *= x; }
    }
    for (int e = 0; e < c; ++e) sc[e] = sqrt(sc[e]);
    for (int p = 0; p < c; ++p) {
        bv[p] = dd_from(0.0);
        for (int q = 0; q < c; ++q) { G[p * c + q] = dd_from(0.0); V[p * c + q] = dd_from(p == q ? 1.0 : 0.0); }
    }
    for (int i = 0; i < k; ++i) {
        const double x = xm(i), y = ym(i);
        double pw = 1.0;
        for (int e = 0; e < c; ++e) { a[e] = pw / sc[e]; pw *= x; }
        for (int p = 0; p < c; ++p) {
            bv[p] = dd_add_prod(bv[p], a[p], y);
            for (int q = p; q < c; ++q) G[p * c + q] = dd_add_prod(G[p * c + q], a[p], a[q]);
        }
    }
    for (int p = 0; p < c; ++p)
        for (int q = 0; q < p; ++q) G[p * c + q] = G[q * c + p];
    const int sweeps = 12 + 2 * c;
    for (int sweep = 0; sweep < sweeps; ++sweep) {   // cyclic Jacobi, G -> diag
        bool rotated = false;
        for (int p = 0; p < c - 1; ++p) {
            for (int q = p + 1; q < c; ++q) {
                const dd apq = G[p * c + q];
                const double lim = 3.0e-33 * sqrt(fabs(G[p * c + p].hi) * fabs(G[q * c + q].hi));
                if (!(fabs(apq.hi) > lim)) {
                    G[p * c + q] = G[q * c + p] = dd_from(0.0);
                    continue;
                }
                rotated = true;
                const dd th = dd_div(dd_sub(G[q * c + q], G[p * c + p]), dd_mul_d(apq, 2.0));
                dd t;
                if (fabs(th.hi) > 1e150) {
                    t = dd_div(dd_from(0.5), th);
                } else {
                    const dd ath = (th.hi < 0.0) ? dd_neg(th) : th;
                    t = dd_div(dd_from(1.0), dd_add(ath, dd_sqrt(dd_add(dd_mul(th, th), dd_from(1.0)))));
                    if (th.hi < 0.0) t = dd_neg(t);
                }
                const dd cs = dd_div(dd_from(1.0), dd_sqrt(dd_add(dd_mul(t, t), dd_from(1.0))));
                const dd sn = dd_mul(t, cs);
                const dd tap = dd_mul(t, apq);
                G[p * c + p] = dd_sub(G[p * c + p], tap);
                G[q * c + q] = dd_add(G[q * c + q], tap);
                G[p * c + q] = G[q * c + p] = dd_from(0.0);
                for (int r = 0; r < c; ++r) {
                    if (r != p && r != q) {
                        const dd arp = G[r * c + p], arq = G[r * c + q];
                        G[r * c + p] = G[p * c + r] = dd_sub(dd_mul(cs, arp), dd_mul(sn, arq));
                        G[r * c + q] = G[q * c + r] = dd_add(dd_mul(sn, arp), dd_mul(cs, arq));
                    }
                    const dd vrp = V[r * c + p], vrq = V[r * c + q];
                    V[r * c + p] = dd_sub(dd_mul(cs, vrp), dd_mul(sn, vrq));
                    V[r * c + q] = dd_add(dd_mul(sn, vrp), dd_mul(cs, vrq));
                }
            }
        }
        if (!rotated) break;
    }
    double lmax = 0.0;
    for (int p = 0; p < c; ++p) lmax = fmax(lmax, G[p * c + p].hi);
    const double rcond = (double)k * 2.220446049250313e-16;
    const double cut = rcond * rcond * lmax;
    for (int r = 0; r < c; ++r) sol[r] = dd_from(0.0);
    const bool bad = !(lmax == lmax) || isinf(lmax);
    for (int p = 0; p < c; ++p) {
        if (!(G[p * c + p].hi > cut)) continue;   // truncated direction (s_i <= rcond s_max)
        dd w = dd_from(0.0);
        for (int r = 0; r < c; ++r) w = dd_add(w, dd_mul(V[r * c + p], bv[r]));
        w = dd_div(w, G[p * c + p]);
        for (int r = 0; r < c; ++r) sol[r] = dd_add(sol[r], dd_mul(V[r * c + p], w));
    }
    for (int e = 0; e < c; ++e) {                 // exponent e -> coefficient index m - e
        const dd q = dd_div(sol[e], dd_from(sc[e]));
        coef[m - e] = bad ? TSFA_NAN : (q.hi + q.lo);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// max(real(np.roots(c))) for any degree: leading zeros stripped, trailing zeros are roots at 0 (numpy/lib/_polynomial_impl.py:
// roots), the rest by Aberth-Ehrlich iteration on all roots at once (complex float64; numpy takes the eigenvalues of the
// companion matrix: both deliver the roots to the conditioning of the polynomial).  work: 4 (deg + 1) doubles.  Serial.
// ---------------------------------------------------------------------------------------------------------------
TSFA_DEV double max_real_root_any(const double *cin, int ncoef, double *work) {
    for (int i = 0; i < ncoef; ++i)
        if (cin[i] != cin[i] || isinf(cin[i])) return TSFA_NAN;
    int lo = 0, hi = ncoef - 1;
    while (lo <= hi && cin[lo] == 0.0) ++lo;
    if (lo > hi) return TSFA_NAN;
    int tz = 0;
    while (hi > lo && cin[hi] == 0.0) { --hi; ++tz; }
    const int deg = hi - lo;
    if (deg <= 3) return max_real_root_deg3(cin, ncoef);
    const double *c = cin + lo;   // c[0] x^deg + ... + c[deg]
    double *zr = work, *zi = work + deg, *nr = zi + deg, *ni = nr + deg;
    // starting points: a circle of the Cauchy-type radius  max_k |c_k / c_0|^(1/k), slightly rotated
    double rad = 0.0;
    for (int k = 1; k <= deg; ++k) rad = fmax(rad, pow(fabs(c[k] / c[0]), 1.0 / (double)k));
    if (!(rad > 0.0)) rad = 1.0;
    const double shift = -c[1] / (c[0] * (double)deg);   // centroid of the roots
    for (int j = 0; j < deg; ++j) {
        const double ang = 2.0 * M_PI * ((double)j + 0.25) / (double)deg + 0.4;
        zr[j] = shift + rad * cos(ang);
        zi[j] = rad * sin(ang);
    }
    for (int it = 0; it < 200; ++it) {
        double moved = 0.0, size = 0.0;
        for (int j = 0; j < deg; ++j) {
            // p(z), p'(z) by Horner
            double pr = c[0], pi = 0.0, dr = 0.0, di = 0.0;
            const double x = zr[j], y = zi[j];
            for (int k = 1; k <= deg; ++k) {
                const double ndr = dr * x - di * y + pr, ndi = dr * y + di * x + pi;
                dr = ndr; di = ndi;
                const double npr = pr * x - pi * y + c[k], npi = pr * y + pi * x;
                pr = npr; pi = npi;
            }
            // w = p / p'
            double wr, wi;
            {
                const double den = dr * dr + di * di;
                if (den == 0.0) { wr = 0.0; wi = 0.0; }
                else { wr = (pr * dr + pi * di) / den; wi = (pi * dr - pr * di) / den; }
            }
            // s = sum_{l != j} 1 / (z_j - z_l)
            double sr = 0.0, si = 0.0;
            for (int l = 0; l < deg; ++l) {
                if (l == j) continue;
                const double ex = x - zr[l], ey = y - zi[l];
                const double den = ex * ex + ey * ey;
                if (den == 0.0) continue;
                sr += ex / den;
                si -= ey / den;
            }
            // step = w / (1 - w s)
            const double qr = 1.0 - (wr * sr - wi * si), qi = -(wr * si + wi * sr);
            const double den = qr * qr + qi * qi;
            double stepr = wr, stepi = wi;
            if (den != 0.0) { stepr = (wr * qr + wi * qi) / den; stepi = (wi * qr - wr * qi) / den; }
            nr[j] = x - stepr;
            ni[j] = y - stepi;
            moved = fmax(moved, fabs(stepr) + fabs(stepi));
            size = fmax(size, fabs(nr[j]) + fabs(ni[j]));
        }
        for (int j = 0; j < deg; ++j) { zr[j] = nr[j]; zi[j] = ni[j]; }
        if (!(moved > 1e-15 * size)) break;
    }
    double best = (tz > 0) ? 0.0 : -TSFA_INF;
    for (int j = 0; j < deg; ++j) best = fmax(best, zr[j]);
    return best;
}

// ---------------------------------------------------------------------------------------------------------------
// lag sums: out[k] = sum_{t < n - k} (x[t] - mean)(x[t + k] - mean), k = k0 .. k1, thread = lag (the sum of a lag runs in the
// order of t on one thread: the same bits from every workgroup size, the emulation included)
// ---------------------------------------------------------------------------------------------------------------
template <class X>
TSFA_DEV void gen_lag_sums(const Blk &b, X xv, int n, double mean, int k0, int k1, double *out) {
    for (int k = k0 + b.tid; k <= k1; k += b.nt) {
        double s = 0.0;
        for (int t = 0; t + k < n; ++t) s += (xv(t) - mean) * (xv(t + k) - mean);
        out[k] = s;
    }
    blk_sync();
}

// ---------------------------------------------------------------------------------------------------------------
// number_cwt_peaks (fc.py:1320) = len(scipy.signal.find_peaks_cwt(x, widths = 1 .. W, wavelet = _ricker)), any W: the
// algorithm of scipy/signal/_peak_finding.py as written there -- the whole transform first (thread = column), the relative
// maxima of every row, then the rows from the widest down: every maximum picks the closest live line of the row's snapshot
// (thread = maximum; np.argmin: the first of equals) within widths[row] / 4, or starts a line; a line that misses more than
// gap_thresh = 1 rows ends.  A line counts when it has ceil(W / 4) points and |cwt[row, col] / noise[col]| >= 1 at its LAST
// point (the smallest row: scipy sorts a line by row before it reads element 0), noise = the 10th percentile
// (scoreatpercentile) of row 0 in a window of ceil(n / 20) columns.  Bookkeeping by thread 0.
// ---------------------------------------------------------------------------------------------------------------
template <class X>
TSFA_DEV double gen_cwt_peaks(const Blk &b, X xv, int n, int W, const GenSlot &S, int transform_rows) {
    if (n < 3) return 0.0;
    if (transform_rows > 0) {   // (row w - 1 does not depend on how many follow: once, for the largest n of the plan)
        for (int row = 0; row < transform_rows; ++row) {
            const int w = row + 1;
            const int nw = (10 * w < n) ? 10 * w : n;
            blk_sync();
            for (int k = b.tid; k < nw; k += b.nt) S.taps[k] = ricker_tap(nw, (double)w, nw - 1 - k);   // conj(wavelet(N, width)[::-1])
            blk_sync();
            double *o = S.cw + (size_t)row * n;
            for (int c = b.tid; c < n; c += b.nt) o[c] = conv_same_at(xv, n, S.taps, nw, c);
        }
        blk_sync();
        // noise floor per column
        const int ws_ = (n + 19) / 20;   // ceil(n / 20)
        const int hf = ws_ / 2, odd = ws_ % 2;
        const double *r0 = S.cw;
        for (int c = b.tid; c < n; c += b.nt) {
            const int ws = (c - hf > 0) ? c - hf : 0;
            const int we = (c + hf + odd < n) ? c + hf + odd : n;
            const int m = we - ws;
            const double idx = 10.0 / 100.0 * (double)(m - 1);
            const int i0 = (int)idx;
            double s0 = 0.0, s1 = 0.0;
            for (int a = ws; a < we; ++a) {
                const double ea = r0[a];
                int rank = 0;
                for (int e = ws; e < we; ++e) {
                    const double ec = r0[e];
                    rank += (ec < ea || (ec == ea && e < a)) ? 1 : 0;
                }
                if (rank == i0) s0 = ea;
                if (rank == i0 + 1) s1 = ea;
            }
            double nz;
            if ((double)i0 == idx) nz = s0;
            else {
                const double j = (double)(i0 + 1);
                const double w0 = j - idx, w1 = idx - (double)i0;
                nz = (s0 * w0 + s1 * w1) / (w0 + w1);
            }
            S.noise[c] = nz;
        }
        blk_sync();
    }
    // relative maxima (mode="clip": the two end columns compare with themselves and never qualify)
    for (long long e = b.tid; e < (long long)W * n; e += b.nt) {
        const int c = (int)(e % n);
        const double *o = S.cw + (size_t)(e - c);
        S.mask[e] = (c >= 1 && c < n - 1 && o[c] > o[c - 1] && o[c] > o[c + 1]) ? 1 : 0;
    }
    blk_sync();
    const int min_len = (W + 3) / 4;
    int nlines = 0, kept = 0, started = 0;
    int *red_i = (int *)(void *)(S.snap + 2 * n + 8);   // two ints behind the snapshot: [maxima of the row, live lines]
    for (int row = W - 1; row >= 0; --row) {
        blk_sync();
        if (b.tid == 0) {
            int nm = 0;
            const unsigned char *mk = S.mask + (size_t)row * n;
            for (int c = 0; c < n; ++c)
                if (mk[c]) S.maxc[nm++] = c;
            red_i[0] = nm;
        }
        blk_sync();
        const int nm = red_i[0];
        if (!started) {
            if (nm == 0) continue;
            started = 1;
            for (int i = b.tid; i < nm; i += b.nt) { S.l_col[i] = S.maxc[i]; S.l_row[i] = row; S.l_len[i] = 1; S.l_gap[i] = 0; }
            nlines = nm;
            continue;
        }
        for (int i = b.tid; i < nlines; i += b.nt) { S.l_gap[i] += 1; S.snap[i] = S.l_col[i]; }
        blk_sync();
        const double maxdist = (double)(row + 1) / 4.0;
        for (int q = b.tid; q < nm; q += b.nt) {
            const int col = S.maxc[q];
            int best = -1, bd = 0x7fffffff;
            for (int i = 0; i < nlines; ++i) {
                int d = col - S.snap[i];
                d = d < 0 ? -d : d;
                if (d < bd) { bd = d; best = i; }
            }
            S.choice[q] = (best >= 0 && (double)bd <= maxdist) ? best : -1;
        }
        blk_sync();
        if (b.tid == 0) {
            for (int q = 0; q < nm; ++q) {
                const int col = S.maxc[q], ch = S.choice[q];
                if (ch >= 0) { S.l_col[ch] = col; S.l_row[ch] = row; S.l_len[ch] += 1; S.l_gap[ch] = 0; }
                else { S.l_col[nlines] = col; S.l_row[nlines] = row; S.l_len[nlines] = 1; S.l_gap[nlines] = 0; ++nlines; }
            }
            int live = 0;
            for (int i = 0; i < nlines; ++i) {
                if (S.l_gap[i] > 1) {   // gap_thresh = ceil(widths[0]) = 1
                    if (S.l_len[i] >= min_len) {
                        const double snr = fabs(S.cw[(size_t)S.l_row[i] * n + S.l_col[i]] / S.noise[S.l_col[i]]);
                        if (!(snr < 1.0)) ++kept;
                    }
                } else {
                    if (live != i) { S.l_col[live] = S.l_col[i]; S.l_row[live] = S.l_row[i]; S.l_len[live] = S.l_len[i]; S.l_gap[live] = S.l_gap[i]; }
                    ++live;
                }
            }
            red_i[1] = live;
        }
        blk_sync();
        nlines = red_i[1];
    }
    blk_sync();
    if (b.tid == 0) {
        for (int i = 0; i < nlines; ++i) {
            if (S.l_len[i] >= min_len) {
                const double snr = fabs(S.cw[(size_t)S.l_row[i] * n + S.l_col[i]] / S.noise[S.l_col[i]]);
                if (!(snr < 1.0)) ++kept;
            }
        }
        red_i[0] = kept;
    }
    blk_sync();
    return (double)red_i[0];
}

template <class X>
struct GenIdx {
    X f;
    TSFA_MEM double operator[](int i) const { return f(i); }
};

// ---------------------------------------------------------------------------------------------------------------
// query_similarity_count (fc.py:2475-2519) with a query subsequence Q of m >= 3 samples:
//     count = #{ i : dist(Q, T[i .. i + m)) <= threshold },  i = 0 .. n - m,
// dist = stumpy.core.mass (z-normalised Euclidean distance, normalize=True) or stumpy.core.mass_absolute (plain Euclidean).
// stumpy (>= 1.11.1, the reference's dependency, setup.cfg:47) is not part of the reference's tree and not installed in this
// image; this is its PUBLISHED definition (core.py: _calculate_squared_distance / _mass_absolute), evaluated directly:
//     normalised:  both constant -> 0;  one constant -> sqrt(m);  else  rho = sum (q - mu_Q)(t - mu_T) / (m sigma_Q sigma_T),
//                  D^2 = |2 m (1 - min(rho, 1))|,  D^2 < 1e-14 -> 0  (stumpy's STUMPY_D_SQUARED_THRESHOLD: an exact match counts
//                  at the default threshold 0);
//     absolute:    D^2 = sum (q - t)^2.
// stumpy forms the same quantities from an FFT sliding dot product and rolling moments, QT - m mu_Q mu_T; the two agree to
// round-off, so a count can differ only where a distance equals the threshold to ~1e-7 relative (tests/parity.py R15).
// Parity is anchored on the reference's own unit test (test_feature_calculations.py:2017-2037: 0 / 6 / 0 / 91).
// A series shorter than the query: NaN here; the host raises the ValueError stumpy raises (reference_errors.py).
template <class X>
TSFA_DEV double gen_query_count(const Blk &b, X xv, int n, const double *q, int m, double thr, bool normalize) {
    if (m < 3 || q == nullptr) return TSFA_NAN;   // fc.py:2511: Q.size >= 3, else np.nan
    const int k = n - m + 1;
    if (k <= 0) return TSFA_NAN;
    double mq = 0.0, sq = 0.0, qmin = TSFA_INF, qmax = -TSFA_INF;
    bool qfinite = true;
    if (normalize) {   // uniform: every lane walks the (short) query
        for (int j = 0; j < m; ++j) { const double v = q[j]; mq += v; qmin = fmin(qmin, v); qmax = fmax(qmax, v); qfinite = qfinite && (fabs(v) <= 1.7976931348623157e308); }
        mq /= (double)m;
        for (int j = 0; j < m; ++j) { const double d = q[j] - mq; sq += d * d; }
        sq = sqrt(sq / (double)m);
    } else {
        for (int j = 0; j < m; ++j) qfinite = qfinite && (fabs(q[j]) <= 1.7976931348623157e308);
    }
    const bool qconst = (qmax == qmin);
    const double dm = (double)m;
    double cnt = 0.0;
    for (int i = b.tid; i < k; i += b.nt) {
        double d2;
        if (normalize) {
            double mt = 0.0, tmin = TSFA_INF, tmax = -TSFA_INF;
            bool tfinite = true;
            for (int j = 0; j < m; ++j) { const double v = xv(i + j); mt += v; tmin = fmin(tmin, v); tmax = fmax(tmax, v); tfinite = tfinite && (fabs(v) <= 1.7976931348623157e308); }
            mt /= dm;
            double st = 0.0, c = 0.0;
            for (int j = 0; j < m; ++j) { const double d = xv(i + j) - mt; st += d * d; c += (q[j] - mq) * d; }
            st = sqrt(st / dm);
            const bool tconst = (tmax == tmin);
            if (!tfinite || !qfinite) d2 = TSFA_INF;            // a window that holds a non-finite sample matches nothing
            else if (qconst && tconst) d2 = 0.0;
            else if (qconst || tconst) d2 = dm;
            else {
                double rho = c / (dm * sq * st);
                rho = (rho > 1.0) ? 1.0 : rho;
                d2 = fabs(2.0 * dm * (1.0 - rho));
            }
            if (d2 < 1e-14) d2 = 0.0;
        } else {
            d2 = 0.0;
            bool tfinite = true;
            for (int j = 0; j < m; ++j) { const double v = xv(i + j); const double d = q[j] - v; d2 += d * d; tfinite = tfinite && (fabs(v) <= 1.7976931348623157e308); }
            if (!tfinite || !qfinite) d2 = TSFA_INF;
        }
        cnt += (sqrt(d2) <= thr) ? 1.0 : 0.0;
    }
    return blk_sum(b, cnt);   // integers below 2^53: exact in any order
}

template <class X>
TSFA_DEV void fam_general_series(const Blk &b, X xv, int n, const TsfaSpec *specs, int nspecs, double *out_row,
                                 const GenSlot &S, const TsfaGenPlan &g, const double *pool = nullptr) {
    const double dn = (double)n;
    const double mean = np_sum(b, n, [=](int i) { return xv(i); }) / dn;
    const double var = np_sum(b, n, [=](int i) { const double d = xv(i) - mean; return d * d; }) / dn;
    blk_sync();

    // ---- agg_autocorrelation (fc.py:387): a = acf(x, adjusted=True, nlags=max maxlag)[1:], f_agg(a[:maxlag]) ----
    if (g.acf_maxlag >= 0) {
        const bool flat = (fabs(var) < 1e-10) || n == 1;   // a = [0] * len(x)
        const int amax = (g.acf_maxlag < n - 1) ? g.acf_maxlag : (n - 1);
        if (!flat && amax >= 1) {
            gen_lag_sums(b, xv, n, mean, 0, amax, S.av);
            const double a0 = S.av[0] / dn;
            blk_sync();
            for (int k = 1 + b.tid; k <= amax; k += b.nt) S.av[k] = (S.av[k] / (double)(n - k)) / a0;
            blk_sync();
        }
        for (int s = 0; s < nspecs; ++s) {
            const TsfaSpec sp = specs[s];
            if (sp.calc != TSFA_C_AGG_AUTOCORRELATION) continue;
            const int agg = (int)sp.p[0], ml = (int)sp.p[1];
            double r = TSFA_NAN;
            if (flat) {
                r = 0.0;
            } else {
                const int len = (ml < n - 1) ? ml : (n - 1);
                const double *a = S.av + 1;
                if (len <= 0) {
                    r = TSFA_NAN;
                } else if (agg == TSFA_AGG_MEAN) {
                    r = np_sum(b, len, [=](int i) { return a[i]; }) / (double)len;
                } else if (agg == TSFA_AGG_VAR) {
                    const double m = np_sum(b, len, [=](int i) { return a[i]; }) / (double)len;
                    r = np_sum(b, len, [=](int i) { const double d = a[i] - m; return d * d; }) / (double)len;
                } else {   // median
                    const int np2 = next_pow2(len < 2 ? 2 : len);
                    blk_sync();
                    for (int i = b.tid; i < np2; i += b.nt) S.srt[i] = (i < len) ? a[i] : TSFA_INF;
                    blk_bitonic_sort(b, S.srt, np2);
                    r = (len & 1) ? S.srt[(len - 1) / 2] : (0.0 + S.srt[len / 2 - 1] + S.srt[len / 2]) / 2.0;
                    blk_sync();
                }
            }
            if (b.tid == 0) out_row[sp.col] = r;
        }
        blk_sync();
    }

    // ---- partial_autocorrelation (fc.py:440): pacf(x, method="ld", nlags=max_lag) ----
    if (g.pacf_maxlag >= 0) {
        const int want = g.pacf_maxlag;
        int max_lag = 0;
        if (n > 1) max_lag = (want >= n / 2) ? (n / 2 - 1) : want;
        double *prev = S.lev, *cur = prev + (max_lag + 2), *pac = cur + (max_lag + 2);
        if (max_lag > 0) {
            gen_lag_sums(b, xv, n, mean, 0, max_lag, S.av);
            for (int k = b.tid; k <= max_lag; k += b.nt) S.av[k] = S.av[k] / (double)(n - k);   // acovf(adjusted=True)
            blk_sync();
            const double *acv = S.av;
            double sig = 0.0;
            if (b.tid == 0) {
                const double p1 = acv[1] / acv[0];
                prev[1] = p1;
                pac[0] = 1.0;
                pac[1] = p1;
            }
            blk_sync();
            sig = acv[0] - prev[1] * acv[1];
            for (int k = 2; k <= max_lag; ++k) {
                double d = 0.0;
                for (int j = 1 + b.tid; j < k; j += b.nt) d += prev[j] * acv[k - j];
                d = blk_sum(b, d);
                const double pkk = (acv[k] - d) / sig;
                for (int j = 1 + b.tid; j < k; j += b.nt) cur[j] = prev[j] - pkk * prev[k - j];
                if (b.tid == 0) { cur[k] = pkk; pac[k] = pkk; }
                sig = sig * (1.0 - pkk * pkk);
                blk_sync();
                double *t = prev; prev = cur; cur = t;
            }
        }
        blk_sync();
        for (int s = b.tid; s < nspecs; s += b.nt) {
            const TsfaSpec sp = specs[s];
            if (sp.calc != TSFA_C_PARTIAL_AUTOCORRELATION) continue;
            const int l = (int)sp.p[0];
            out_row[sp.col] = (max_lag > 0 && l >= 0 && l <= max_lag) ? pac[l] : TSFA_NAN;
        }
        blk_sync();
    }

    // ---- friedrich_coefficients (fc.py:2082) / max_langevin_fixed_point (fc.py:2134) ----
    if (g.fr_maxr > 0) {
        const int ns = n - 1;
        bool sorted = false;
        int done_m = -1, done_r = -1;
        double *coef = S.pf;                       // fr_maxm + 1 coefficients of the last fit
        double *pwork = S.pf + (g.fr_maxm + 2);
        for (int s = 0; s < nspecs; ++s) {
            const TsfaSpec sp = specs[s];
            if (sp.calc != TSFA_C_FRIEDRICH_COEFFICIENTS && sp.calc != TSFA_C_MAX_LANGEVIN_FIXED_POINT) continue;
            int coeff = 0, m, r;
            if (sp.calc == TSFA_C_FRIEDRICH_COEFFICIENTS) { coeff = (int)sp.p[0]; m = (int)sp.p[1]; r = (int)sp.p[2]; }
            else { m = (int)sp.p[0]; r = (int)sp.p[1]; }
            if (m != done_m || r != done_r) {
                blk_sync();
                if (!sorted && ns >= 1) {
                    const int np2 = next_pow2(ns < 2 ? 2 : ns);
                    for (int i = b.tid; i < np2; i += b.nt) S.srt[i] = (i < ns) ? xv(i) : TSFA_INF;
                    blk_bitonic_sort(b, S.srt, np2);
                    sorted = true;
                }
                const double *sr = S.srt;
                bool ok = false;
                if (ns >= 1) ok = friedrich_bin_means(b, GenIdx<X>{xv}, n, [=](int i) { return sr[i]; }, r, S.fw);
                blk_sync();
                if (b.tid == 0) {
                    for (int c = 0; c <= m; ++c) coef[c] = TSFA_NAN;
                    if (ok) {
                        const double *sx = S.fw + (r + 1), *sy = sx + r, *cnt = sy + r;
                        double *xm = S.fw + (6 * r + 16), *ym = xm + r;
                        int k = 0;
                        for (int j = 0; j < r; ++j)
                            if (cnt[j] > 0.0) { xm[k] = sx[j] / cnt[j]; ym[k] = sy[j] / cnt[j]; ++k; }
                        if (k >= 1) {
                            const double *cx = xm, *cy = ym;
                            polyfit_svd_dd_n([=](int i) { return cx[i]; }, [=](int i) { return cy[i]; }, k, m, coef, pwork);
                        }
                    }
                }
                blk_sync();
                done_m = m;
                done_r = r;
            }
            if (b.tid == 0) {
                double v;
                if (sp.calc == TSFA_C_FRIEDRICH_COEFFICIENTS) v = (coeff >= 0 && coeff <= m) ? coef[coeff] : TSFA_NAN;
                else v = max_real_root_any(coef, m + 1, pwork);
                out_row[sp.col] = v;
            }
        }
        blk_sync();
    }

    // ---- lempel_ziv_complexity (fc.py:1825) ----
    if (g.lz) {
        double mn = TSFA_INF, mx = -TSFA_INF;
        for (int i = b.tid; i < n; i += b.nt) {
            const double x = xv(i);
            mn = fmin(mn, x);
            mx = fmax(mx, x);
        }
        const double vmin = blk_min(b, mn), vmax = blk_max(b, mx);
        const int cap = 2 * next_pow2(n < 2 ? 2 : n);
        int lg = 1;
        while ((1 << lg) < cap) ++lg;
        for (int s = 0; s < nspecs; ++s) {
            const TsfaSpec sp = specs[s];
            if (sp.calc != TSFA_C_LEMPEL_ZIV_COMPLEXITY) continue;
            const int bins = (int)sp.p[0];
            blk_sync();
            // symbols: np.searchsorted(np.linspace(min, max, bins + 1)[1:], x, side="left") = #{edges < x}
            for (int i = b.tid; i < n; i += b.nt) {
                const double x = xv(i);
                int lo = 0, hi = bins;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (np_linspace_at(vmin, vmax, bins + 1, mid + 1) < x) lo = mid + 1; else hi = mid;
                }
                S.seq[i] = (unsigned int)lo;
            }
            for (int i = b.tid; i < cap; i += b.nt) S.ht[i] = 0ull;
            blk_sync();
            if (b.tid == 0) {
                // the phrase set is prefix-closed (a phrase enters when its prefix is known): a trie whose nodes are the slots
                // of an open-addressing table keyed by (parent slot + 1, symbol)
                long long count = 0;
                unsigned long long node = 0ull;
                const unsigned long long mask = (unsigned long long)cap - 1ull;
                for (int i = 0; i < n; ++i) {
                    const unsigned long long key = ((node << 32) | (unsigned long long)S.seq[i]) + 1ull;
                    unsigned long long h = (key * 0x9E3779B97F4A7C15ull) >> (64 - lg);
                    unsigned long long curk = S.ht[h];
                    while (curk != key && curk != 0ull) {
                        h = (h + 1ull) & mask;
                        curk = S.ht[h];
                    }
                    const bool fresh = (curk == 0ull);
                    if (fresh) { S.ht[h] = key; ++count; node = 0ull; }
                    else node = h + 1ull;
                }
                out_row[sp.col] = (double)count / dn;
            }
        }
        blk_sync();
    }

    // ---- query_similarity_count (fc.py:2475) with a query: p = (threshold, normalize, offset into the pool, m) ----
    if (g.query > 0) {
        for (int s = 0; s < nspecs; ++s) {
            const TsfaSpec sp = specs[s];
            if (sp.calc != TSFA_C_QUERY_SIMILARITY_COUNT) continue;
            const int m = (int)sp.p[3];
            const double v = gen_query_count(b, xv, n, (pool && m > 0) ? pool + (long long)sp.p[2] : nullptr, m, sp.p[0], sp.p[1] != 0.0);
            if (b.tid == 0) out_row[sp.col] = v;
        }
        blk_sync();
    }

    // ---- number_cwt_peaks (fc.py:1320): the transform once for the largest n (row w - 1 does not depend on how many follow) ----
    if (g.cwt_maxw > 0) {
        int todo = g.cwt_maxw;
        for (int s = 0; s < nspecs; ++s) {
            const TsfaSpec sp = specs[s];
            if (sp.calc != TSFA_C_NUMBER_CWT_PEAKS) continue;
            const double v = gen_cwt_peaks(b, xv, n, (int)sp.p[0], S, todo);
            todo = 0;
            if (b.tid == 0) out_row[sp.col] = v;
        }
        blk_sync();
    }
}

#endif
