// Family ENTROPY: sample_entropy (fc.py:1701) and approximate_entropy (fc.py:1759).
//
// Both count, for every template i of length m (and m+1), the templates j whose Chebyshev distance is within a
// tolerance r = c * np.std(x).  All requested tolerances (5 approximate-entropy r's + the sample-entropy 0.2)
// share ONE sweep over the (i, j) template pairs: thread (lane) = row i, the loop runs over columns j, x[j..j+m]
// is a wave-uniform broadcast read, and per-row counters live in registers.  Distances and comparisons are done in
// float64 on the exact sample values so the counts are bit-identical to the reference's float64 arithmetic
// (a float32 subtraction would mis-round |x_i - x_j| next to a threshold; see DESIGN.md "entropy exactness").
#ifndef TSFA_FAM_ENTROPY_H
#define TSFA_FAM_ENTROPY_H

#include "tsfa_common.h"

#define TSFA_ENT_MAXK 8

struct EntAcc {          // per-threshold accumulators of one sweep (template length m)
    double sum_log_m;    // sum_i log(C_m[i] / (N - m + 1))
    double sum_log_m1;   // sum_i log(C_{m+1}[i] / (N - m))
    double sum_cnt_m;    // sum_i C_m[i]
    double sum_cnt_m1;   // sum_i C_{m+1}[i]
};

// One sweep for template length m = 2 with NK thresholds thr[0..NK).  Every thread returns the block totals.
//
// Pruning (exact): a pair can only match if |x_i - x_j| <= r_max, so the length-2 templates are sorted by their first
// sample (perm[], bitonic sort in LDS) and each pass of nt rows -- a narrow band of values -- only visits the columns
// whose first sample lies within r_max of that band (two binary searches).  For N(0,1) data and r_max = 0.9 sigma
// this skips ~45% of the pairs.  xs[n] and xs[n+1] must hold +inf (the length-3 extension of the last templates then
// never matches).
template <int NK>
TSFA_DEV void entropy_sweep_m2(const Blk &b, const double *xs, int n, const double *thr, const unsigned short *perm,
                               EntAcc *acc) {
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
        const double xi0 = xs[ri], xi1 = xs[ri + 1], xi2 = xs[ri + 2];
        // column window: first samples within r_max of [band_lo, band_hi], widened by a rounding margin
        const double band_lo = xs[perm[q0]], band_hi = xs[perm[q1 - 1]];
        const double margin = 1e-9 * (fabs(band_lo) + fabs(band_hi) + rmax);
        const double key_lo = band_lo - rmax - margin, key_hi = band_hi + rmax + margin;
        int jlo, jhi;
        {
            int lo = 0, hi = nrow_m;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (xs[perm[mid]] < key_lo) lo = mid + 1; else hi = mid; }
            jlo = lo;
            lo = jlo; hi = nrow_m;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (xs[perm[mid]] <= key_hi) lo = mid + 1; else hi = mid; }
            jhi = lo;
        }
        int c2[NK], c3[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) { c2[k] = 0; c3[k] = 0; }
        for (int q = jlo; q < jhi; ++q) {
            const int c = perm[q];
            const double xj0 = xs[c], xj1 = xs[c + 1], xj2 = xs[c + 2];  // xs[n] = +inf: no length-3 template there
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
    for (int k = 0; k < NK; ++k) {
        acc[k].sum_log_m = blk_sum(b, slm[k]);
        acc[k].sum_log_m1 = blk_sum(b, slm1[k]);
        acc[k].sum_cnt_m = blk_sum(b, scm[k]);
        acc[k].sum_cnt_m1 = blk_sum(b, scm1[k]);
    }
}

// perm[0 .. n-2] = indices of the length-2 templates sorted by their first sample (ties by index); np2 = padded size
TSFA_DEV void entropy_sort_templates(const Blk &b, const double *xs, int n, unsigned short *perm, int np2) {
    const int nrow_m = n - 1;
    blk_sync();
    for (int i = b.tid; i < np2; i += b.nt) perm[i] = (unsigned short)((i < nrow_m) ? i : 0xFFFF);
    for (int k = 2; k <= np2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            blk_sync();
            for (int t = b.tid; t < (np2 >> 1); t += b.nt) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const bool up = ((i & k) == 0);
                const unsigned short a = perm[i], c = perm[l];
                const double ka = (a == 0xFFFF) ? TSFA_INF : xs[a], kc = (c == 0xFFFF) ? TSFA_INF : xs[c];
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
TSFA_DEV void entropy_sweep_generic(const Blk &b, const double *xs, int n, int m, double thr, EntAcc *acc) {
    const int nrow_m = n - m + 1, nrow_m1 = n - m;
    double slm = 0.0, slm1 = 0.0, scm = 0.0, scm1 = 0.0;
    for (int i = b.tid; i < nrow_m; i += b.nt) {
        int cm = 0, cm1 = 0;
        for (int j = 0; j < nrow_m; ++j) {
            double d = 0.0;
            for (int t = 0; t < m; ++t) d = fmax(d, fabs(xs[i + t] - xs[j + t]));
            cm += (d <= thr) ? 1 : 0;
            if (i < nrow_m1 && j < nrow_m1) {
                d = fmax(d, fabs(xs[i + m] - xs[j + m]));
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
//   xs   : LDS, n + 2 doubles (xs[n], xs[n+1] are overwritten with +inf sentinels)
//   thr  : LDS scratch >= TSFA_ENT_MAXK doubles;   perm : LDS, next_pow2(n) unsigned shorts
TSFA_DEV void fam_entropy_series(const Blk &b, double *xs, int n, const TsfaSpec *specs, int nspecs,
                                 double *out_row, double *thr, unsigned short *perm) {
    // np.std(x), numpy summation order (the tolerances are c * np.std(x))
    const double dn = (double)n;
    const double mean = np_sum(b, n, [=](int i) { return xs[i]; }) / dn;
    const double var = np_sum(b, n, [=](int i) { const double d = xs[i] - mean; return d * d; }) / dn;
    const double sd = sqrt(var);
    blk_sync();
    if (b.tid == 0) { xs[n] = TSFA_INF; xs[n + 1] = TSFA_INF; }
    bool sorted = false;

    // m = 2 specs are batched TSFA_ENT_MAXK at a time
    int done = 0;
    while (done < nspecs) {
        int idx[TSFA_ENT_MAXK];
        int nk = 0;
        int s = done;
        for (; s < nspecs && nk < TSFA_ENT_MAXK; ++s) {
            const TsfaSpec sp = specs[s];
            const bool is_m2 = (sp.calc == TSFA_C_SAMPLE_ENTROPY) ||
                               (sp.calc == TSFA_C_APPROXIMATE_ENTROPY && (int)sp.p[0] == 2);
            if (!is_m2) continue;
            idx[nk] = s;
            blk_sync();
            if (b.tid == 0) thr[nk] = (sp.calc == TSFA_C_SAMPLE_ENTROPY) ? 0.2 * sd : sp.p[1] * sd;
            ++nk;
        }
        done = s;
        if (nk == 0) break;
        blk_sync();
        if (b.tid == 0)
            for (int k = nk; k < TSFA_ENT_MAXK; ++k) thr[k] = -1.0;  // never matches
        blk_sync();
        EntAcc acc[TSFA_ENT_MAXK];
        if (n >= 3) {
            if (!sorted) {
                entropy_sort_templates(b, xs, n, perm, next_pow2(n - 1));
                sorted = true;
            }
            if (nk <= 1) entropy_sweep_m2<1>(b, xs, n, thr, perm, acc);
            else if (nk <= 2) entropy_sweep_m2<2>(b, xs, n, thr, perm, acc);
            else if (nk <= 4) entropy_sweep_m2<4>(b, xs, n, thr, perm, acc);
            else if (nk <= 6) entropy_sweep_m2<6>(b, xs, n, thr, perm, acc);
            else entropy_sweep_m2<8>(b, xs, n, thr, perm, acc);
        }
        for (int k = 0; k < nk; ++k) {
            const TsfaSpec sp = specs[idx[k]];
            double v;
            if (sp.calc == TSFA_C_APPROXIMATE_ENTROPY) {
                if (n <= 3) v = 0.0;  // N <= m + 1
                else v = apen_from_acc(acc[k], n, 2);
            } else {
                if (n < 3) v = TSFA_NAN;  // no length-3 template: A = 0, B = 0 -> -log(0/0)
                else v = sampen_from_acc(acc[k], n, 2);
            }
            if (b.tid == 0) out_row[sp.col] = v;
        }
    }
    // generic m
    for (int s = 0; s < nspecs; ++s) {
        const TsfaSpec sp = specs[s];
        if (sp.calc != TSFA_C_APPROXIMATE_ENTROPY || (int)sp.p[0] == 2) continue;
        const int m = (int)sp.p[0];
        double v;
        if (n <= m + 1 || m < 1) {
            v = 0.0;
        } else {
            EntAcc a;
            entropy_sweep_generic(b, xs, n, m, sp.p[1] * sd, &a);
            v = apen_from_acc(a, n, m);
        }
        if (b.tid == 0) out_row[sp.col] = v;
    }
}

#endif
