// Family SORT, second pass: the Langevin polynomial fit of friedrich_coefficients / max_langevin_fixed_point
// (fc.py:131-173, :2082, :2134) when the design is ill-conditioned.
//
// The reference fits the bin means with np.polyfit (numpy/lib/_polynomial_impl.py): Vandermonde matrix by running
// products, columns scaled to unit Euclidean norm, lstsq with rcond = len(x) * eps (LAPACK gelsd: singular values
// s_i <= rcond * s_max are treated as ZERO and the minimum-norm solution of the rest is returned), coefficients
// divided by the column scales.  Bin means of a series whose mean is far from zero relative to its spread
// (|mean| / std = q) give singular values 1, 1/q, 1/q^2, 1/q^3 of the scaled cubic design: from q ~ 3e4 on the last
// direction falls under the cut and the reference returns the minimum-norm cubic of a rank-3 (then rank-2) problem,
// which differs from the full-rank least-squares solution by orders of magnitude (50 Hz +- 5 mHz grid frequency,
// 101 325 +- 5 Pa, any float64 counter).
//
// k_sort solves the scaled design by Householder QR in float64, which is the same polynomial while the design is well
// conditioned (error ~ eps * cond^2 * |resid|).  When the diagonal of R spans more than TSFA_PF_FLAG (or there are fewer bins than
// coefficients) it hands the bin means to this pass: ONE LANE PER LISTED SERIES (k_langevin_dd), serial code in
// double-double arithmetic:
//   * the scaled design A exactly as numpy forms it (float64 running products, float64 column norms and divisions);
//   * G = A^T A and b = A^T y from exact products, accumulated in double-double;
//   * cyclic Jacobi eigen-decomposition G = V L V^T (<= 4 x 4) in double-double;
//   * keep  l_i > rcond^2 * l_max  (s_i > rcond * s_max),  c = sum_kept v_i (v_i . b) / l_i,  c / scale.
// Squaring the condition number costs nothing here: a kept direction has l_i / l_max >= 4e-29 against the 1e-32 of the
// arithmetic, so the smallest kept component still carries 3 digits where the reference's own float64 SVD carries
// none (its singular values have an ABSOLUTE error of ~eps * s_max), and 1e-8 relative or better for cond <= 1e12.
// Measured against 60-digit arithmetic the reference's coefficients deviate by ~eps * s_max / s_min(kept)
// (tests/polyfit_mp.py); this pass by < 1e-9 of that.
#ifndef TSFA_FAM_LANGEVIN_DD_H
#define TSFA_FAM_LANGEVIN_DD_H

#include "tsfa_dd.h"

#define TSFA_PF_MAXC 4          // coefficients of the largest fit (TSFA_FRIEDRICH_MAX_M + 1)
#define TSFA_PF_FLAG 1e-3       // min |R_kk| / max |R_kk| of the float64 QR below which a fit is redone here: the float64 solve
                                // errs by ~eps cond^2 |resid|, 2e-10 at the threshold (1e-5 left 2e-6 -- found by the fuzz)
#define TSFA_PF_SWEEPS 12

// (record layout of a deferred fit: TSFA_PF_HDR / tsfa_pf_slot_doubles in tsfa_specs.h)

// np.polyfit(x, y, deg = m) for k points (k >= 1, 1 <= m < TSFA_PF_MAXC): coef[0 .. m], highest power first.
template <class XF, class YF>
TSFA_DEV void polyfit_svd_dd(XF xm, YF ym, int k, int m, double *coef) {
    const int cols = m + 1;
    double sc[TSFA_PF_MAXC];
#pragma unroll
    for (int c = 0; c < TSFA_PF_MAXC; ++c) sc[c] = 1.0;
    {   // scale = sqrt((lhs * lhs).sum(axis = 0)): rows added in order
        double ss[TSFA_PF_MAXC] = {0.0, 0.0, 0.0, 0.0};
        for (int i = 0; i < k; ++i) {
            const double x = xm(i);
            double pw = 1.0;
#pragma unroll
            for (int e = 0; e < TSFA_PF_MAXC; ++e) {   // exponent e belongs to column m - e
                if (e <= m) ss[e] += pw * pw;
                pw *= x;
            }
        }
#pragma unroll
        for (int e = 0; e < TSFA_PF_MAXC; ++e)
            if (e <= m) sc[e] = sqrt(ss[e]);    // indexed by EXPONENT
    }
    // G (by exponent pair) and b, exact products accumulated in double-double
    dd G[TSFA_PF_MAXC][TSFA_PF_MAXC], bv[TSFA_PF_MAXC], V[TSFA_PF_MAXC][TSFA_PF_MAXC];
#pragma unroll
    for (int p = 0; p < TSFA_PF_MAXC; ++p) {
        bv[p] = dd_from(0.0);
#pragma unroll
        for (int q = 0; q < TSFA_PF_MAXC; ++q) {
            G[p][q] = dd_from(0.0);
            V[p][q] = dd_from(p == q ? 1.0 : 0.0);
        }
    }
    for (int i = 0; i < k; ++i) {
        const double x = xm(i), y = ym(i);
        double a[TSFA_PF_MAXC];
        double pw = 1.0;
#pragma unroll
        for (int e = 0; e < TSFA_PF_MAXC; ++e) {
            a[e] = (e <= m) ? pw / sc[e] : 0.0;
            pw *= x;
        }
#pragma unroll
        for (int p = 0; p < TSFA_PF_MAXC; ++p) {
            if (p > m) continue;
            bv[p] = dd_add_prod(bv[p], a[p], y);
#pragma unroll
            for (int q = 0; q < TSFA_PF_MAXC; ++q)
                if (q >= p && q <= m) G[p][q] = dd_add_prod(G[p][q], a[p], a[q]);
        }
    }
#pragma unroll
    for (int p = 0; p < TSFA_PF_MAXC; ++p)
#pragma unroll
        for (int q = 0; q < TSFA_PF_MAXC; ++q)
            if (q < p) G[p][q] = G[q][p];
    // cyclic Jacobi, G -> diag, V accumulates the rotations (columns = eigenvectors)
    for (int sweep = 0; sweep < TSFA_PF_SWEEPS; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int p = 0; p < TSFA_PF_MAXC - 1; ++p) {
#pragma unroll
            for (int q = p + 1; q < TSFA_PF_MAXC; ++q) {
                if (q > m) continue;
                const dd apq = G[p][q];
                const double lim = 3.0e-33 * sqrt(fabs(G[p][p].hi) * fabs(G[q][q].hi));
                if (!(fabs(apq.hi) > lim)) {
                    G[p][q] = G[q][p] = dd_from(0.0);
                    continue;
                }
                rotated = true;
                // theta = (a_qq - a_pp) / (2 a_pq);  t = sgn(theta) / (|theta| + sqrt(theta^2 + 1))
                const dd th = dd_div(dd_sub(G[q][q], G[p][p]), dd_mul_d(apq, 2.0));
                dd t;
                if (fabs(th.hi) > 1e150) {
                    t = dd_div(dd_from(0.5), th);
                } else {
                    const dd ath = (th.hi < 0.0) ? dd_neg(th) : th;
                    t = dd_div(dd_from(1.0), dd_add(ath, dd_sqrt(dd_add(dd_mul(th, th), dd_from(1.0)))));
                    if (th.hi < 0.0) t = dd_neg(t);
                }
                const dd c = dd_div(dd_from(1.0), dd_sqrt(dd_add(dd_mul(t, t), dd_from(1.0))));
                const dd s = dd_mul(t, c);
                const dd tap = dd_mul(t, apq);
                G[p][p] = dd_sub(G[p][p], tap);
                G[q][q] = dd_add(G[q][q], tap);
                G[p][q] = G[q][p] = dd_from(0.0);
#pragma unroll
                for (int r = 0; r < TSFA_PF_MAXC; ++r) {
                    if (r != p && r != q && r <= m) {
                        const dd arp = G[r][p], arq = G[r][q];
                        G[r][p] = G[p][r] = dd_sub(dd_mul(c, arp), dd_mul(s, arq));
                        G[r][q] = G[q][r] = dd_add(dd_mul(s, arp), dd_mul(c, arq));
                    }
                    if (r <= m) {
                        const dd vrp = V[r][p], vrq = V[r][q];
                        V[r][p] = dd_sub(dd_mul(c, vrp), dd_mul(s, vrq));
                        V[r][q] = dd_add(dd_mul(s, vrp), dd_mul(c, vrq));
                    }
                }
            }
        }
        if (!rotated) break;
    }
    double lmax = 0.0;
#pragma unroll
    for (int p = 0; p < TSFA_PF_MAXC; ++p)
        if (p <= m) lmax = fmax(lmax, G[p][p].hi);
    const double rcond = (double)k * 2.220446049250313e-16;
    const double cut = rcond * rcond * lmax;
    dd sol[TSFA_PF_MAXC];
#pragma unroll
    for (int r = 0; r < TSFA_PF_MAXC; ++r) sol[r] = dd_from(0.0);
    bool bad = !(lmax == lmax) || isinf(lmax);
#pragma unroll
    for (int p = 0; p < TSFA_PF_MAXC; ++p) {
        if (p > m) continue;
        if (!(G[p][p].hi > cut)) continue;         // truncated direction (also: zero / negative round-off eigenvalue)
        dd w = dd_from(0.0);
#pragma unroll
        for (int r = 0; r < TSFA_PF_MAXC; ++r)
            if (r <= m) w = dd_add(w, dd_mul(V[r][p], bv[r]));
        w = dd_div(w, G[p][p]);
#pragma unroll
        for (int r = 0; r < TSFA_PF_MAXC; ++r)
            if (r <= m) sol[r] = dd_add(sol[r], dd_mul(V[r][p], w));
    }
    // exponent e -> coefficient index m - e
#pragma unroll
    for (int e = 0; e < TSFA_PF_MAXC; ++e) {
        if (e > m) continue;
        const dd q = dd_div(sol[e], dd_from(sc[e]));
        coef[m - e] = bad ? TSFA_NAN : (q.hi + q.lo);
    }
    (void)cols;
}

#endif
