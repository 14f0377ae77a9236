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
template <int NK>
TSFA_DEV void entropy_sweep_m2(const Blk &b, const double *xs, int n, const double *thr, EntAcc *acc) {
    const int nrow_m = n - 1;   // templates of length 2: i in [0, n-2]
    const int nrow_m1 = n - 2;  // templates of length 3: i in [0, n-3]
    double r[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) r[k] = thr[k];
    double slm[NK], slm1[NK], scm[NK], scm1[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) { slm[k] = 0.0; slm1[k] = 0.0; scm[k] = 0.0; scm1[k] = 0.0; }
    const double dm = (double)nrow_m, dm1 = (double)nrow_m1;

    const int npass = (nrow_m + b.nt - 1) / b.nt;
    for (int pass = 0; pass < npass; ++pass) {
        const int i = pass * b.nt + b.tid;
        const bool row_m = (i < nrow_m), row_m1 = (i < nrow_m1);
        const double xi0 = row_m ? xs[i] : 0.0;
        const double xi1 = row_m ? xs[i + 1] : 0.0;
        const double xi2 = row_m1 ? xs[i + 2] : 0.0;
        int c2[NK], c3[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) { c2[k] = 0; c3[k] = 0; }
        double xj0 = xs[0], xj1 = xs[1];
        // columns valid for both template lengths
        for (int j = 0; j < nrow_m1; ++j) {
            const double xj2 = xs[j + 2];
            const double d0 = fabs(xi0 - xj0), d1 = fabs(xi1 - xj1), d2 = fabs(xi2 - xj2);
            const double m2 = fmax(d0, d1);
            const double m3 = fmax(m2, d2);
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                c2[k] += (m2 <= r[k]) ? 1 : 0;
                c3[k] += (m3 <= r[k]) ? 1 : 0;
            }
            xj0 = xj1;
            xj1 = xj2;
        }
        {   // last column j = n-2: only a length-2 template
            const double d0 = fabs(xi0 - xj0), d1 = fabs(xi1 - xj1);
            const double m2 = fmax(d0, d1);
#pragma unroll
            for (int k = 0; k < NK; ++k) c2[k] += (m2 <= r[k]) ? 1 : 0;
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

// Evaluate the ENTROPY specs of one series.   thr : LDS scratch >= TSFA_ENT_MAXK doubles
TSFA_DEV void fam_entropy_series(const Blk &b, const double *xs, int n, const TsfaSpec *specs, int nspecs,
                                 double *out_row, double *thr) {
    // np.std(x), numpy summation order (the tolerances are c * np.std(x))
    const double dn = (double)n;
    const double mean = np_sum(b, n, [=](int i) { return xs[i]; }) / dn;
    const double var = np_sum(b, n, [=](int i) { const double d = xs[i] - mean; return d * d; }) / dn;
    const double sd = sqrt(var);

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
            if (nk <= 1) entropy_sweep_m2<1>(b, xs, n, thr, acc);
            else if (nk <= 2) entropy_sweep_m2<2>(b, xs, n, thr, acc);
            else if (nk <= 4) entropy_sweep_m2<4>(b, xs, n, thr, acc);
            else if (nk <= 6) entropy_sweep_m2<6>(b, xs, n, thr, acc);
            else entropy_sweep_m2<8>(b, xs, n, thr, acc);
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
