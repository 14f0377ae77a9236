// Family AR, second pass: rank-deficient and ill-conditioned regressions.
//
// statsmodels fits AutoReg (fc.py:1459-1508) and every OLS inside adfuller (fc.py:499-545) with the pseudo-inverse
// (regression/linear_model.py:300-338, tools/tools.py:398 pinv_extended, rcond = 1e-15): on a constant, linear,
// periodic, ... series the design is rank-deficient and the reference returns the MINIMUM-NORM solution, uses the
// RANK (not the column count) in the AIC and in the residual degrees of freedom, and -- tsatools.add_trend(...,
// has_constant="skip"), tsatools.py:112-136 -- leaves the constant out when the design already holds an exactly
// constant non-zero column.  k_ar solves the normal equations by Cholesky in float64, which is only trustworthy while
// every pivot keeps a fair share of its column (fam_ar.h: TSFA_AR_PIVOT_TOL); series that fail the test are listed and
// handled here, one workgroup per listed series, in double-double arithmetic (~32 digits):
//   * the Gram matrix of the RAW (uncentred) design from exact products (the minimum-norm solution is not invariant
//     under centring), lag products by the same diagonal recurrence as the first pass;
//   * a Cholesky factorization in natural column order that SKIPS dependent columns (pivot <= 1e-26 of its column:
//     the square of a singular-value ratio of 1e-13, the double-double image of pinv's 1e-15 cut-off): the kept
//     columns give X = Q R with R = L_kept^T of full row rank, every nested prefix fit (adfuller's lag search)
//     reads its rank and its residual sum from the same factor;
//   * minimum norm:  beta = R^T (R R^T)^-1 Q^T y,  (X^T X)^+ [0,0] = | (R R^T)^-1 R e_0 |^2.
// With 32 digits the normal equations resolve designs up to cond(X) ~ 1e12, so near-degenerate but full-rank series
// (a noiseless float32 sine, a ramp with 1e-9 noise) also agree with the reference's SVD.
// A residual sum that is zero in exact arithmetic (a perfect fit) gives AIC = -inf and a 0/0 or x/0 test statistic
// here; the reference divides two round-off numbers there (tests/parity.py documents that exclusion).
#ifndef TSFA_FAM_AR_DD_H
#define TSFA_FAM_AR_DD_H

#include "fam_ar.h"
#include "tsfa_dd.h"

// all-reduce of a double-double over the workgroup (every thread receives the same bits: dd_add is symmetric)
TSFA_DEV dd blk_sum_dd(const Blk &b, dd v) {
#if TSFA_GPU
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const dd u{__shfl_xor(v.hi, o), __shfl_xor(v.lo, o)};
        v = dd_add(v, u);
    }
    if (b.nt > 64) {
        const int nw = b.nt >> 6;
        blk_sync();
        if ((b.tid & 63) == 0) {
            b.red[2 * (b.tid >> 6)] = v.hi;
            b.red[2 * (b.tid >> 6) + 1] = v.lo;
        }
        blk_sync();
        v = dd{b.red[0], b.red[1]};
        for (int w = 1; w < nw; ++w) v = dd_add(v, dd{b.red[2 * w], b.red[2 * w + 1]});
    }
#endif
    return v;
}

#define TSFA_DD_SKIP_TOL 1e-26   // pivot / column norm^2 below which a column is dependent
#define TSFA_DD_ZERO_SSR 1e-24   // ssr / yy below which a fit is perfect (ssr = 0 in exact arithmetic)

// Cholesky in natural column order, skipping dependent columns (right-looking; rows of a column are dealt over the
// threads).  On exit kept[j] tells whether column j entered the factor; for a kept j, G[i + j*ld] (i >= j) = L(i, j)
// for EVERY row i (dependent rows included: they are the columns of R beyond its triangle).  `mask` (optional)
// pre-excludes columns; `skip_rel` (optional) receives |pivot| / diagonal of every skipped column.  Returns the rank, or -1 if a non-masked pivot is not positive although tol == 0.
TSFA_DEV int dd_chol_skip(const Blk &b, dd *G, int p, int ld, int *kept, const int *mask, double tol, dd *diag0,
                         double *skip_rel = nullptr) {
    blk_sync();
    for (int a = b.tid; a < p; a += b.nt) diag0[a] = G[a + a * ld];
    int rank = 0;
    bool bad = false;
    for (int j = 0; j < p; ++j) {
        blk_sync();
        const dd d = G[j + j * ld];
        const double d0 = diag0[j].hi;
        bool keep = !(mask && !mask[j]);
        if (keep) {
            if (tol > 0.0) keep = (d0 > 0.0) && (d.hi > tol * d0);
            else if (!(d.hi > 0.0)) { keep = false; bad = true; }
        }
        if (b.tid == 0) {
            kept[j] = keep ? 1 : 0;
            if (skip_rel) skip_rel[j] = (keep || !(d0 > 0.0)) ? 0.0 : fabs(d.hi) / d0;   // pivot a skipped column was left with
        }
        if (!keep) continue;
        ++rank;
        const dd sd = dd_sqrt(d);
        blk_sync();
        for (int i = j + b.tid; i < p; i += b.nt) G[i + j * ld] = (i == j) ? sd : dd_div(G[i + j * ld], sd);
        blk_sync();
        // trailing update: G(i, k) -= L(i, j) L(k, j), j < k <= i
        const int m = p - j - 1;
        for (int e = b.tid; e < m * m; e += b.nt) {
            const int i = j + 1 + e / m, k = j + 1 + e % m;
            if (k > i) continue;
            G[i + k * ld] = dd_sub(G[i + k * ld], dd_mul(G[i + j * ld], G[k + j * ld]));
        }
    }
    blk_sync();
    return bad ? -1 : rank;
}

// w = (L_kept)^-1 g over the kept columns, in place (w[j] = 0 for skipped columns)
TSFA_DEV void dd_forward_kept(const Blk &b, const dd *L, int p, int ld, const int *kept, dd *w) {
    for (int j = 0; j < p; ++j) {
        blk_sync();
        if (!kept[j]) {
            if (b.tid == 0) w[j] = dd_from(0.0);
            continue;
        }
        const dd wj = dd_div(w[j], L[j + j * ld]);
        blk_sync();
        if (b.tid == 0) w[j] = wj;
        for (int i = j + 1 + b.tid; i < p; i += b.nt) w[i] = dd_sub(w[i], dd_mul(L[i + j * ld], wj));
    }
    blk_sync();
}

// Minimum-norm least squares from the skipping factor:  beta = R^T (R R^T)^-1 w  with R = L_kept^T.
//   S: p x p scratch (receives R R^T and its factor); z, diag0: p scratch; kept2: p ints.
// If want_cov0, *cov0 = ((X^T X)^+)[0, 0].  Returns false if R R^T fails to factor (never for a consistent factor).
TSFA_DEV bool dd_min_norm(const Blk &b, const dd *L, int p, int ld, const int *kept, const dd *w, dd *S, dd *z, dd *diag0,
                          int *kept2, dd *beta, bool want_cov0, double *cov0) {
    blk_sync();
    for (int e = b.tid; e < p * p; e += b.nt) {
        const int a = e / p, c = e % p;
        if (c > a) continue;
        dd s = dd_from(0.0);
        if (kept[a] && kept[c])
            for (int i = a; i < p; ++i) s = dd_add(s, dd_mul(L[i + a * ld], L[i + c * ld]));
        S[a + c * ld] = s;
    }
    const int r2 = dd_chol_skip(b, S, p, ld, kept2, kept, 0.0, diag0);
    if (r2 < 0) return false;
    for (int pass = 0; pass < (want_cov0 ? 2 : 1); ++pass) {
        blk_sync();
        if (b.tid == 0) {
            // rhs: pass 0 = w, pass 1 = R e_0 (only its first entry, L(0, 0), is non-zero)
            for (int a = 0; a < p; ++a) z[a] = (pass == 0) ? w[a] : dd_from(0.0);
            if (pass == 1 && kept[0]) z[0] = L[0];
            for (int a = 0; a < p; ++a) {  // S = F F^T:  F y = rhs
                if (!kept[a]) continue;
                dd s = z[a];
                for (int c = 0; c < a; ++c)
                    if (kept[c]) s = dd_sub(s, dd_mul(S[a + c * ld], z[c]));
                z[a] = dd_div(s, S[a + a * ld]);
            }
            for (int a = p - 1; a >= 0; --a) {  // F^T x = y
                if (!kept[a]) continue;
                dd s = z[a];
                for (int c = a + 1; c < p; ++c)
                    if (kept[c]) s = dd_sub(s, dd_mul(S[c + a * ld], z[c]));
                z[a] = dd_div(s, S[a + a * ld]);
            }
            if (pass == 1) {
                dd s = dd_from(0.0);
                for (int a = 0; a < p; ++a)
                    if (kept[a]) s = dd_add(s, dd_mul(z[a], z[a]));
                diag0[0] = s;
            }
        }
        blk_sync();
        if (pass == 0) {
            for (int c = b.tid; c < p; c += b.nt) {
                dd s = dd_from(0.0);
                for (int a = 0; a <= c; ++a)
                    if (kept[a]) s = dd_add(s, dd_mul(L[c + a * ld], z[a]));
                beta[c] = s;
            }
        } else {
            *cov0 = diag0[0].hi;
        }
    }
    blk_sync();
    return true;
}

// ---------------------------------------------------------------------------------------------------------------
// The pseudo-inverse TRUNCATION regime.  statsmodels' pinv drops singular values <= 1e-15 s_max of the RAW design and
// counts rank = matrix_rank(diag(s)), tolerance s_max * p * eps.  A series with |mean| = mu >> spread gives the design
// [1, x_{t-1}, ...] a smallest singular value ~ s_max * sigma / (k mu^2): from mu / sigma ~ 1e7 on (epoch seconds, 1e8 +
// noise, any float64 counter) the reference solves a problem of lower rank although no column is dependent, and the
// skipping Cholesky above (which only recognises DEPENDENT columns) returns the full-rank least-squares solution
// instead -- 1e17 x off in the intercept.  Such designs take the SVD route:
//   * the Gram matrix of the design in SHIFTED coordinates x' = x - c (c = the series mean, so x' is the size of the
//     spread and nothing cancels): G' = X'^T X' = L L^T by the skipping Cholesky; the raw design is X = X' S with
//     S = I + c e_const d^T (d marks the shifted columns), hence X = Q (L^T S) and the singular values / right
//     singular vectors of X are those of the small matrix M = L^T S;
//   * one-sided (Hestenes) Jacobi on the columns of A = M^T = L + c d L[const, :] in double-double: on exit column i of
//     A is s_i v_i, and the same rotations turn w = L^-1 X'^T y into u_i . y.  One-sided Jacobi works on the factor,
//     not on a Gram matrix, so a singular value 1e-12 of the largest still carries ~20 digits (an eigen-decomposition
//     of X^T X would leave it 8, and none to the intercept of 6.6e8 +- 0.01);
//   * keep s_i > 1e-15 s_max:  beta = sum_kept v_i (u_i . y) / s_i,  ssr = ssr_full + sum_dropped (u_i . y)^2,
//     rank = #{s_i > p eps s_max},  (X^T X)^+[0, 0] = sum_kept v_i[0]^2 / s_i^2.
// Exactly dependent columns never enter the factor (zero singular values), so this route is a superset of the skipping
// one; it is taken when s_min / s_max of the raw design, estimated by inverse iteration on the raw factor, is within
// 20x of the cuts (dd_needs_svd).

#define TSFA_DD_PINV_RCOND 1e-15
#define TSFA_DD_SVD_TRIGGER 1e-13   // estimated s_min / s_max below which the SVD route decides
#define TSFA_DD_JACOBI_SWEEPS 20

// x = L^-T w over the kept columns, in place
TSFA_DEV void dd_backward_kept(const Blk &b, const dd *L, int p, int ld, const int *kept, dd *w) {
    for (int j = p - 1; j >= 0; --j) {
        blk_sync();
        if (!kept[j]) {
            if (b.tid == 0) w[j] = dd_from(0.0);
            continue;
        }
        const dd wj = dd_div(w[j], L[j + j * ld]);
        blk_sync();
        if (b.tid == 0) w[j] = wj;
        for (int i = b.tid; i < j; i += b.nt)
            if (kept[i]) w[i] = dd_sub(w[i], dd_mul(L[j + i * ld], wj));
    }
    blk_sync();
}

// Does the design behind the skipping factor L (diag0 = the Gram diagonal) need the SVD route?  Yes if a skipped
// column is not an EXACT dependency (its pivot is not at the arithmetic's noise floor), or if s_min / s_max of the kept
// block -- l_min by three steps of inverse iteration, l_max <= trace -- is below TSFA_DD_SVD_TRIGGER.
TSFA_DEV bool dd_needs_svd(const Blk &b, const dd *L, int p, int ld, const int *kept, const dd *diag0, const double *skipped_rel,
                           dd *v) {
    blk_sync();
    double trace = 0.0;
    bool soft_skip = false;
    int nk = 0;
    for (int a = 0; a < p; ++a) {
        if (kept[a]) { trace += diag0[a].hi; ++nk; }
        else if (skipped_rel[a] > 1e-29) soft_skip = true;
    }
    if (soft_skip) return true;
    if (nk == 0 || !(trace > 0.0)) return false;
    for (int a = b.tid; a < p; a += b.nt) v[a] = dd_from(kept[a] ? 1.0 / sqrt(diag0[a].hi) : 0.0);
    double lmin = trace;
    for (int it = 0; it < 3; ++it) {
        dd_forward_kept(b, L, p, ld, kept, v);
        dd_backward_kept(b, L, p, ld, kept, v);
        double nrm2 = 0.0;
        for (int a = 0; a < p; ++a) nrm2 += v[a].hi * v[a].hi;   // uniform: every thread reads the same LDS values
        const double nrm = sqrt(nrm2);
        blk_sync();
        if (!(nrm > 0.0) || isinf(nrm)) return true;
        lmin = 1.0 / nrm;   // |v| was 1 (first step: ~1): l_min ~ |v| / |G^-1 v|
        for (int a = b.tid; a < p; a += b.nt) v[a] = dd{v[a].hi / nrm, v[a].lo / nrm};
        blk_sync();
    }
    return lmin < TSFA_DD_SVD_TRIGGER * TSFA_DD_SVD_TRIGGER * trace;
}

// One-sided (Hestenes) Jacobi: plane rotations from the right orthogonalise the r columns (length p) of A (column-major,
// leading dimension ld); the row vector wv (r entries) takes the same rotations.  On exit A = V diag(s) up to column
// order: column i has norm s_i and direction v_i.
TSFA_DEV void dd_hestenes(const Blk &b, dd *A, int p, int r, int ld, dd *wv) {
    for (int sweep = 0; sweep < TSFA_DD_JACOBI_SWEEPS; ++sweep) {
        bool any = false;
        for (int i = 0; i < r - 1; ++i) {
            for (int j = i + 1; j < r; ++j) {
                blk_sync();
                dd al = dd_from(0.0), be = dd_from(0.0), ga = dd_from(0.0);
                for (int k = b.tid; k < p; k += b.nt) {
                    const dd ai = A[k + i * ld], aj = A[k + j * ld];
                    al = dd_add(al, dd_mul(ai, ai));
                    be = dd_add(be, dd_mul(aj, aj));
                    ga = dd_add(ga, dd_mul(ai, aj));
                }
                al = blk_sum_dd(b, al);
                be = blk_sum_dd(b, be);
                ga = blk_sum_dd(b, ga);
                if (!(fabs(ga.hi) > 1e-31 * sqrt(al.hi * be.hi))) continue;   // uniform (also: a zero column)
                any = true;
                const dd ze = dd_div(dd_sub(be, al), dd_mul_d(ga, 2.0));
                dd t;
                if (fabs(ze.hi) > 1e150) {
                    t = dd_div(dd_from(0.5), ze);
                } else {
                    const dd az = (ze.hi < 0.0) ? dd_neg(ze) : ze;
                    t = dd_div(dd_from(1.0), dd_add(az, dd_sqrt(dd_add(dd_mul(ze, ze), dd_from(1.0)))));
                    if (ze.hi < 0.0) t = dd_neg(t);
                }
                const dd c = dd_div(dd_from(1.0), dd_sqrt(dd_add(dd_mul(t, t), dd_from(1.0))));
                const dd sn = dd_mul(t, c);
                blk_sync();
                for (int k = b.tid; k < p; k += b.nt) {
                    const dd ai = A[k + i * ld], aj = A[k + j * ld];
                    A[k + i * ld] = dd_sub(dd_mul(c, ai), dd_mul(sn, aj));
                    A[k + j * ld] = dd_add(dd_mul(sn, ai), dd_mul(c, aj));
                }
                if (b.tid == 0 && wv) {
                    const dd wi = wv[i], wj = wv[j];
                    wv[i] = dd_sub(dd_mul(c, wi), dd_mul(sn, wj));
                    wv[j] = dd_add(dd_mul(sn, wi), dd_mul(c, wj));
                }
            }
        }
        if (!any) break;
    }
    blk_sync();
}

struct DdPinvFit {
    double dropped;  // sum over the dropped directions of (u_i . y)^2: what truncation adds to the residual sum
    int rank;        // matrix_rank(diag(s)): #{s_i > s_max pcols eps}
    double cov0;     // ((X^T X)^+)[row0, row0]
    dd beta0;        // coefficient of design column row0
};
// statsmodels OLS(...).fit(method="pinv") quantities from the orthogonalised columns A = V diag(s) (p x r) and the
// rotated wv = U^T y.  s2 (r entries of scratch) receives s_i^2.  If beta != nullptr it receives all p coefficients.
TSFA_DEV DdPinvFit dd_pinv_from_svd(const Blk &b, const dd *A, int p, int r, int ld, const dd *wv, int pcols, int row0,
                                    dd *s2, dd *beta) {
    blk_sync();
    for (int i = b.tid; i < r; i += b.nt) {
        dd acc = dd_from(0.0);
        for (int k = 0; k < p; ++k) acc = dd_add(acc, dd_mul(A[k + i * ld], A[k + i * ld]));
        s2[i] = acc;
    }
    blk_sync();
    double lmax = 0.0;
    for (int i = 0; i < r; ++i) lmax = fmax(lmax, s2[i].hi);
    const double cut = TSFA_DD_PINV_RCOND * TSFA_DD_PINV_RCOND * lmax;
    const double rtol = (double)pcols * 2.220446049250313e-16;
    const double rcut = rtol * rtol * lmax;
    DdPinvFit f;
    f.rank = 0;
    dd dropped = dd_from(0.0), c0 = dd_from(0.0), b0 = dd_from(0.0);
    for (int i = 0; i < r; ++i) {    // uniform
        const dd l = s2[i];
        if (l.hi > rcut) ++f.rank;
        if (!(l.hi > cut) || !(l.hi > 0.0)) {
            dropped = dd_add(dropped, dd_mul(wv[i], wv[i]));
            continue;
        }
        const dd a0 = A[row0 + i * ld];
        b0 = dd_add(b0, dd_div(dd_mul(a0, wv[i]), l));
        c0 = dd_add(c0, dd_div(dd_mul(a0, a0), dd_mul(l, l)));
    }
    f.dropped = dropped.hi;
    f.cov0 = c0.hi;
    f.beta0 = b0;
    if (beta) {
        for (int a = b.tid; a < p; a += b.nt) {
            dd acc = dd_from(0.0);
            for (int i = 0; i < r; ++i) {
                const dd l = s2[i];
                if (!(l.hi > cut) || !(l.hi > 0.0)) continue;
                acc = dd_add(acc, dd_div(dd_mul(A[a + i * ld], wv[i]), l));
            }
            beta[a] = acc;
        }
        blk_sync();
    }
    return f;
}

// A = M^T = L + c d L[crow, :] over the kept columns of the leading m x m block of the skipping factor L (see above):
// row a of A belongs to design column a, column i to the i-th KEPT factor column; shifted(a) tells whether design column
// a was shifted by c.  wv receives the kept entries of w.  Returns the number of columns r.
template <class SH>
TSFA_DEV int dd_build_mt(const Blk &b, const dd *L, int m, int ld, const int *kept, double c, int crow, SH shifted,
                         const dd *w, dd *A, dd *wv) {
    blk_sync();
    int r = 0;
    for (int i = 0; i < m; ++i) {
        if (!kept[i]) continue;
        for (int a = b.tid; a < m; a += b.nt) {
            dd v = (a >= i) ? L[a + i * ld] : dd_from(0.0);
            if (crow >= 0 && crow >= i && shifted(a)) v = dd_add(v, dd_mul_d(L[crow + i * ld], c));
            A[a + r * ld] = v;
        }
        if (b.tid == 0) wv[r] = w[i];
        ++r;
    }
    blk_sync();
    return r;
}

// Lag products of sequence s over rows t in [t0, t1) in double-double: T[i + j*ld] = sum_t s(t-i) s(t-j) (lower
// triangle, 0 <= j <= i <= Lg) and C[j] = sum_t s(t-j).  Requires t0 >= Lg.
template <class S>
TSFA_DEV void dd_lag_products(const Blk &b, S s, int Lg, int t0, int t1, dd *T, int ld, dd *C) {
    for (int j = 0; j <= Lg; ++j) {
        dd a = dd_from(0.0);
        for (int t = t0 + b.tid; t < t1; t += b.nt) a = dd_add_prod(a, s(t), s(t - j));
        a = blk_sum_dd(b, a);
        if (b.tid == 0) T[j] = a;
    }
    dd c0 = dd_from(0.0);
    for (int t = t0 + b.tid; t < t1; t += b.nt) c0 = dd_add(c0, dd_from(s(t)));
    c0 = blk_sum_dd(b, c0);
    if (b.tid == 0) {
        C[0] = c0;
        for (int j = 0; j < Lg; ++j) C[j + 1] = dd_add(dd_add(C[j], dd_from(s(t0 - 1 - j))), dd_from(-s(t1 - 1 - j)));
    }
    blk_sync();
    for (int dg = b.tid; dg <= Lg; dg += b.nt) {  // one lane per diagonal i - j = dg
        dd v = T[dg];
        for (int j = 0; dg + j + 1 <= Lg; ++j) {
            const int i = dg + j;
            v = dd_add_prod(v, s(t0 - 1 - i), s(t0 - 1 - j));
            v = dd_add_prod(v, -s(t1 - 1 - i), s(t1 - 1 - j));
            T[(i + 1) + (j + 1) * ld] = v;
        }
    }
    blk_sync();
}

// LDS scratch of the second pass, in doubles (ArDdLds::scratch_doubles): 2 matrices of P*P dd + 8 vectors of (P+1) dd
// + (P+1) doubles + 2 P ints, rounded up

// flags: bit 0 = ar_coefficient, bit 1 = augmented_dickey_fuller (which calculators the first pass gave up on)
template <class X>
TSFA_DEV void fam_ar_degenerate_series(const Blk &b, X xv, int n, const TsfaSpec *specs, int nspecs, double *out_row,
                                       double *scratch, int P, int flags, int adf_mode = TSFA_AUTOLAG_AIC) {
    dd *T = (dd *)(void *)scratch;        // lag products; later R R^T (skipping route) or A = M^T (SVD route)
    dd *G = T + P * P;                    // Gram matrix -> skipping factor
    dd *C = G + P * P;                    // column sums
    dd *V = C + (P + 1);                  // level products (ADF)
    dd *g = V + (P + 1);                  // rhs -> w
    dd *z = g + (P + 1);
    dd *beta = z + (P + 1);
    dd *diag0 = beta + (P + 1);
    dd *misc = diag0 + (P + 1);           // P + 1
    int *kept = (int *)(void *)(misc + (P + 1));
    int *kept2 = kept + P;
    dd *ev = (dd *)(void *)(kept2 + P + (P & 1));   // P + 1: inverse-iteration vector / rotated w of the SVD route
    double *skip_rel = (double *)(void *)(ev + (P + 1));   // P + 1

    // the series mean (any constant of that size would do): the shift of the SVD route
    double xmean;
    {
        dd acc = dd_from(0.0);
        for (int t = b.tid; t < n; t += b.nt) acc = dd_add(acc, dd_from(xv(t)));
        acc = blk_sum_dd(b, acc);
        xmean = acc.hi / (double)n;
        if (!(xmean == xmean) || isinf(xmean)) xmean = 0.0;
    }

    // ---------------------------------------------------------------------------------------------------------
    // augmented Dickey-Fuller, regression="c", autolag="AIC" (stattools.adfuller)
    // ---------------------------------------------------------------------------------------------------------
    if (flags & 2) {
        double r_stat = TSFA_NAN, r_p = TSFA_NAN, r_lag = TSFA_NAN;
        const int M = adf_maxlag_for(n);
        if (M >= 0 && M + 3 <= P) {
            auto dif = [=](int t) { return xv(t + 1) - xv(t); };  // np.diff(x) in float64
            const int t0 = M, t1 = n - 1;
            const int nobs = t1 - t0;
            // add_trend(..., has_constant="skip"): a column of [level, lag 1..U] over rows [r0, t1) that is exactly
            // constant and non-zero suppresses the constant.  Counts the offending columns.
            auto const_cols = [&](int U, int r0) {
                double cnt = 0.0;
                for (int c = b.tid; c <= U; c += b.nt) {  // c = 0: level; c >= 1: lag c
                    const double first = (c == 0) ? xv(r0) : dif(r0 - c);
                    bool same = (first != 0.0);
                    for (int t = r0 + 1; same && t < t1; ++t) same = (((c == 0) ? xv(t) : dif(t - c)) == first);
                    cnt += same ? 1.0 : 0.0;
                }
                return blk_sum(b, cnt) > 0.0;
            };
            const int hasc = const_cols(M, t0) ? 0 : 1;
            // sums over the rows [t0, t1) with the level column shifted by `shift` (0: the raw design)
            dd sx = dd_from(0.0), sxx = dd_from(0.0);
            auto assemble = [&](double shift) {
                blk_sync();
                dd_lag_products(b, dif, M, t0, t1, T, P, C);
                for (int j = 0; j <= M; ++j) {
                    dd a = dd_from(0.0);
                    for (int t = t0 + b.tid; t < t1; t += b.nt) a = dd_add_prod(a, xv(t) - shift, dif(t - j));
                    a = blk_sum_dd(b, a);
                    if (b.tid == 0) V[j] = a;
                }
                dd s1 = dd_from(0.0), s2 = dd_from(0.0);
                for (int t = t0 + b.tid; t < t1; t += b.nt) {
                    const double l = xv(t) - shift;
                    s1 = dd_add(s1, dd_from(l));
                    s2 = dd_add_prod(s2, l, l);
                }
                sx = blk_sum_dd(b, s1);
                sxx = blk_sum_dd(b, s2);
                blk_sync();
            };
            // column kinds of the lag-search design: -1 const, 0 level, j >= 1 lag j
            const int p1 = M + 1 + hasc;
            auto kind1 = [=](int a) { return hasc ? (a == 0 ? -1 : a - 1) : a; };
            auto gram = [&](int ka, int kc) -> dd {  // inner product of two column kinds over rows [t0, t1)
                if (ka < kc) { const int t = ka; ka = kc; kc = t; }
                if (ka == -1) return dd_from((double)nobs);
                if (kc == -1) return (ka == 0) ? sx : C[ka];
                if (ka == 0) return sxx;
                if (kc == 0) return V[ka];
                return T[ka + kc * P];
            };
            auto build_search = [&]() {
                for (int e = b.tid; e < p1 * p1; e += b.nt) {
                    const int a = e / p1, c = e % p1;
                    if (c > a) continue;
                    G[a + c * P] = gram(kind1(a), kind1(c));
                }
                for (int a = b.tid; a < p1; a += b.nt) {
                    const int ka = kind1(a);
                    g[a] = (ka == -1) ? C[0] : (ka == 0 ? V[0] : T[ka]);
                }
                blk_sync();
            };
            assemble(0.0);
            build_search();
            const dd yy = T[0];
            const int startlag = hasc + 1;
            const double dn = (double)nobs;
            dd_chol_skip(b, G, p1, P, kept, nullptr, TSFA_DD_SKIP_TOL, diag0, skip_rel);
            // ("t-stat" reads a coefficient and its variance of every nested fit: always from the SVD of its block, the
            // skipping factor only knows residual sums and ranks)
            const bool svd1 = dd_needs_svd(b, G, p1, P, kept, diag0, skip_rel, ev) || adf_mode == TSFA_AUTOLAG_TSTAT;
            // the shift needs the constant column inside the design (column 0 of the lag search, the last one of the
            // final regression); a design whose constant is another exactly constant column stays unshifted
            const double shift = hasc ? xmean : 0.0;
            if (svd1) {
                // pinv truncation regime: every nested fit from the SVD of its own leading block (uniform control flow)
                if (shift != 0.0) {
                    assemble(shift);
                    build_search();
                    dd_chol_skip(b, G, p1, P, kept, nullptr, TSFA_DD_SKIP_TOL, diag0, skip_rel);
                }
                dd_forward_kept(b, G, p1, P, kept, g);
                int best = -1;
                double best_aic = 0.0;
                bool stop = false;
                const int nfits = (adf_mode == TSFA_AUTOLAG_NONE) ? 0 : (p1 - startlag + 1);
                if (adf_mode == TSFA_AUTOLAG_NONE) best = p1;
                for (int q = 0; q < nfits && !stop; ++q) {
                    // AIC / BIC: every nested fit, smallest first; "t-stat": from the largest down to the first whose last
                    // coefficient is significant
                    const int m = (adf_mode == TSFA_AUTOLAG_TSTAT) ? p1 - q : startlag + q;
                    const int r = dd_build_mt(b, G, m, P, kept, shift, hasc ? 0 : -1, [=](int a) { return kind1(a) == 0; }, g, T, ev);
                    dd full = dd_from(0.0);
                    for (int i = 0; i < r; ++i) full = dd_add(full, dd_mul(ev[i], ev[i]));   // uniform: explained by the full fit
                    dd_hestenes(b, T, m, r, P, ev);
                    const DdPinvFit f = dd_pinv_from_svd(b, T, m, r, P, ev, m, (adf_mode == TSFA_AUTOLAG_TSTAT) ? m - 1 : 0, z, nullptr);
                    double ssr = dd_sub(yy, full).hi + f.dropped;
                    if (!(ssr > TSFA_DD_ZERO_SSR * yy.hi)) ssr = 0.0;
                    if (adf_mode == TSFA_AUTOLAG_TSTAT) {
                        const double tl = f.beta0.hi / sqrt(ssr / (dn - (double)f.rank) * f.cov0);
                        best = m;
                        if (fabs(tl) >= TSFA_AUTOLAG_TSTAT_STOP) stop = true;   // (uniform: every thread holds the same fit)
                        continue;
                    }
                    const double llf = -0.5 * dn * log(2.0 * M_PI) - 0.5 * dn * log(ssr / dn) - 0.5 * dn;
                    const double aic = -2.0 * llf + ((adf_mode == TSFA_AUTOLAG_BIC) ? log(dn) : 2.0) * (double)f.rank;
                    if (best < 0 || aic < best_aic) { best = m; best_aic = aic; }
                }
                blk_sync();
                if (b.tid == 0) misc[0] = dd_from((double)(best - startlag));
                assemble(0.0);   // T held the work matrices: the final regression starts from the raw sums again
            } else {
                dd_forward_kept(b, G, p1, P, kept, g);
                if (b.tid == 0) {
                    // nested fits: `lag` leading columns, lag = startlag .. startlag + M (stattools._autolag)
                    dd acc = dd_from(0.0);
                    int rank = 0, best = -1;
                    double best_aic = 0.0;
                    for (int m = 1; m <= p1; ++m) {
                        if (kept[m - 1]) { acc = dd_add(acc, dd_mul(g[m - 1], g[m - 1])); ++rank; }
                        if (m < startlag) continue;
                        double ssr = dd_sub(yy, acc).hi;
                        if (!(ssr > TSFA_DD_ZERO_SSR * yy.hi)) ssr = 0.0;
                        const double llf = -0.5 * dn * log(2.0 * M_PI) - 0.5 * dn * log(ssr / dn) - 0.5 * dn;
                        const double aic = -2.0 * llf + ((adf_mode == TSFA_AUTOLAG_BIC) ? log(dn) : 2.0) * (double)rank;
                        if (best < 0 || aic < best_aic) { best = m; best_aic = aic; }
                    }
                    if (adf_mode == TSFA_AUTOLAG_NONE) best = p1;
                    misc[0] = dd_from((double)(best - startlag));
                }
            }
            blk_sync();
            const int U = (int)misc[0].hi;
            blk_sync();
            // final regression: [level, lag 1..U] (+ const) over rows [U, n-1)
            const int u0 = U;
            const int nobs2 = t1 - u0;
            const int hasc2 = const_cols(U, u0) ? 0 : 1;
            const int p2 = U + 1 + hasc2;
            auto kind2 = [=](int a) { return (a <= U) ? a : -1; };
            // (T, C, V, sx, sxx hold the sums of the rows [t0, t1) for the level column shifted by `sh`)
            auto build_final = [&](double sh) {
                auto colv = [=](int k, int t) { return k == -1 ? 1.0 : (k == 0 ? xv(t) - sh : dif(t - k)); };
                for (int e = b.tid; e < p2 * p2 + p2 + 1; e += b.nt) {
                    const bool is_yy = (e == p2 * p2 + p2), is_rhs = (e >= p2 * p2) && !is_yy;
                    const int a = is_yy ? 0 : (is_rhs ? e - p2 * p2 : e / p2);
                    const int c = (is_rhs || is_yy) ? 0 : e % p2;
                    if (!is_rhs && !is_yy && c > a) continue;
                    const int ka = kind2(a), kc = kind2(c);
                    dd v;
                    if (is_yy) v = yy;
                    else if (is_rhs) v = (ka == -1) ? C[0] : (ka == 0 ? V[0] : T[ka]);
                    else v = gram(ka, kc);
                    for (int t = u0; t < t0; ++t) {  // the rows the lag search had trimmed
                        const double l = is_yy ? dif(t) : colv(ka, t);
                        const double r = (is_rhs || is_yy) ? dif(t) : colv(kc, t);
                        v = dd_add_prod(v, l, r);
                    }
                    if (is_yy) misc[1] = v;
                    else if (is_rhs) g[a] = v;
                    else G[a + c * P] = v;
                }
                blk_sync();
            };
            build_final(0.0);
            const dd yy2 = misc[1];
            const double lev2 = G[0].hi;  // squared norm of the level column
            const int rank2 = dd_chol_skip(b, G, p2, P, kept, nullptr, TSFA_DD_SKIP_TOL, diag0, skip_rel);
            if (dd_needs_svd(b, G, p2, P, kept, diag0, skip_rel, ev)) {
                const double sh2 = hasc2 ? xmean : 0.0;
                if (sh2 != 0.0) {
                    assemble(sh2);
                    build_final(sh2);
                    dd_chol_skip(b, G, p2, P, kept, nullptr, TSFA_DD_SKIP_TOL, diag0, skip_rel);
                }
                dd_forward_kept(b, G, p2, P, kept, g);
                const int r = dd_build_mt(b, G, p2, P, kept, sh2, hasc2 ? p2 - 1 : -1, [=](int a) { return a == 0; }, g, T, ev);
                dd full = dd_from(0.0);
                for (int i = 0; i < r; ++i) full = dd_add(full, dd_mul(ev[i], ev[i]));
                dd_hestenes(b, T, p2, r, P, ev);
                const DdPinvFit f = dd_pinv_from_svd(b, T, p2, r, P, ev, p2, 0, z, nullptr);
                double ssr = dd_sub(yy2, full).hi + f.dropped;
                if (!(ssr > TSFA_DD_ZERO_SSR * yy2.hi)) ssr = 0.0;
                const double sigma2 = ssr / ((double)nobs2 - (double)f.rank);
                double b0 = f.beta0.hi;
                if (ssr == 0.0 && fabs(b0) * sqrt(lev2) <= 1e-12 * sqrt(yy2.hi)) b0 = 0.0;
                r_stat = b0 / sqrt(sigma2 * f.cov0);
                r_p = (r_stat != r_stat) ? TSFA_NAN : mackinnon_p_c1(r_stat);
                r_lag = (double)U;
            } else {
                dd_forward_kept(b, G, p2, P, kept, g);
                double cov0 = 0.0;
                const bool ok = dd_min_norm(b, G, p2, P, kept, g, T, z, diag0, kept2, beta, true, &cov0);
                if (ok) {
                    dd acc = dd_from(0.0);
                    for (int a = 0; a < p2; ++a)
                        if (kept[a]) acc = dd_add(acc, dd_mul(g[a], g[a]));
                    double ssr = dd_sub(yy2, acc).hi;
                    if (!(ssr > TSFA_DD_ZERO_SSR * yy2.hi)) ssr = 0.0;
                    const double sigma2 = ssr / ((double)nobs2 - (double)rank2);
                    double b0 = beta[0].hi;
                    // a perfect fit whose level coefficient is zero in exact arithmetic: 0 / 0, not (round-off) / 0
                    if (ssr == 0.0 && fabs(b0) * sqrt(lev2) <= 1e-12 * sqrt(yy2.hi)) b0 = 0.0;
                    r_stat = b0 / sqrt(sigma2 * cov0);
                    r_p = (r_stat != r_stat) ? TSFA_NAN : mackinnon_p_c1(r_stat);
                    r_lag = (double)U;
                }
            }
        }
        for (int s = b.tid; s < nspecs; s += b.nt) {
            const TsfaSpec sp = specs[s];
            if (sp.calc != TSFA_C_AUGMENTED_DICKEY_FULLER) continue;
            const int attr = (int)sp.p[0];
            out_row[sp.col] = (attr == TSFA_ADF_TESTSTAT) ? r_stat : (attr == TSFA_ADF_PVALUE ? r_p : (attr == TSFA_ADF_USEDLAG ? r_lag : TSFA_NAN));
        }
        blk_sync();
    }

    // ---------------------------------------------------------------------------------------------------------
    // AutoReg(x, lags=k, trend="c").fit().params  (ar_model.py: OLS on [1, x[t-1..t-k]], rows t in [k, n))
    // ---------------------------------------------------------------------------------------------------------
    if (flags & 1) {
        int done_k = -1;
        bool ok = false;
        for (int s = 0; s < nspecs; ++s) {
            const TsfaSpec sp = specs[s];
            if (sp.calc != TSFA_C_AR_COEFFICIENT) continue;
            const int coeff = (int)sp.p[0], k = (int)sp.p[1];
            if (coeff > k || n < 2 * k + 2 || k + 2 > P || k < 1) continue;  // the first pass's answer stands
            if (done_k != k) {
                const int p = k + 1;
                auto build_ar = [&](double sh) {   // Gram matrix and rhs of [1, x'[t-1..t-k]] -> x'[t], x' = x - sh
                    blk_sync();
                    dd_lag_products(b, [=](int u) { return xv(u) - sh; }, k, k, n, T, P, C);
                    for (int e = b.tid; e < p * p; e += b.nt) {
                        const int a = e / p, c = e % p;
                        if (c > a) continue;
                        G[a + c * P] = (a == 0) ? dd_from((double)(n - k)) : (c == 0 ? C[a] : T[a + c * P]);
                    }
                    for (int a = b.tid; a < p; a += b.nt) g[a] = (a == 0) ? C[0] : T[a];
                    blk_sync();
                };
                build_ar(0.0);
                // The skipping Cholesky is not rank revealing in the natural column order: a design whose smallest singular
                // value is 5e-24 s_max WITHOUT any column depending exactly on its predecessors (32 noisy samples, then a
                // stuck sensor, AR(30): every exact pivot is above 6e-4 of its diagonal, the Gram matrix has condition 1e47)
                // breaks the factorisation down in double-double -- a pivot of -1.6e-3 -- and the fit came out as a rank-28
                // truncation (a random-parameter fuzz find; ~10 % of such series from AR(16) on, none at the AR(10) of
                // ComprehensiveFCParameters).  So the lag columns are first put in the order of a diagonally PIVOTED
                // Cholesky (float64 dry run, the constant stays first): whatever is dependent to working precision comes
                // last and meets a clean, tiny pivot.  perm[a] = design column at position a.
                int *perm = kept2;   // (kept2 is only used inside dd_min_norm, after the order is applied)
                auto order_columns = [&]() {
                    blk_sync();
                    double *S = (double *)(void *)T;   // T is free between build_ar and the routes: p x p doubles
                    for (int e = b.tid; e < p * p; e += b.nt) {
                        const int a = e / p, c = e % p;
                        S[a + c * p] = (c <= a) ? G[a + c * P].hi : G[c + a * P].hi;
                    }
                    blk_sync();
                    if (b.tid == 0) {
                        for (int a = 0; a < p; ++a) perm[a] = a;
                        double dmax = 0.0;
                        for (int a = 1; a < p; ++a) dmax = fmax(dmax, S[a + a * p]);
                        for (int j = 0; j < p; ++j) {
                            if (j >= 1) {   // largest remaining diagonal among the lag columns
                                int best = j;
                                for (int a = j + 1; a < p; ++a)
                                    if (S[a + a * p] > S[best + best * p]) best = a;
                                if (!(S[best + best * p] > 1e-13 * dmax)) break;   // float64 noise: the rest in the order it has
                                if (best != j) {
                                    for (int c = 0; c < p; ++c) { const double t = S[j + c * p]; S[j + c * p] = S[best + c * p]; S[best + c * p] = t; }
                                    for (int c = 0; c < p; ++c) { const double t = S[c + j * p]; S[c + j * p] = S[c + best * p]; S[c + best * p] = t; }
                                    const int t = perm[j]; perm[j] = perm[best]; perm[best] = t;
                                }
                            }
                            const double d = S[j + j * p];
                            if (!(d > 0.0)) break;
                            const double sd = sqrt(d);
                            for (int i = j + 1; i < p; ++i) S[i + j * p] /= sd;
                            for (int i = j + 1; i < p; ++i)
                                for (int c = j + 1; c <= i; ++c) { S[i + c * p] -= S[i + j * p] * S[c + j * p]; S[c + i * p] = S[i + c * p]; }
                        }
                    }
                    blk_sync();
                };
                auto apply_order = [&]() {   // G, g (natural order) -> the order of perm
                    blk_sync();
                    for (int e = b.tid; e < p * p; e += b.nt) {
                        const int a = e / p, c = e % p;
                        if (c <= a) T[a + c * P] = G[a + c * P];
                    }
                    for (int a = b.tid; a < p; a += b.nt) z[a] = g[a];
                    blk_sync();
                    for (int e = b.tid; e < p * p; e += b.nt) {
                        const int a = e / p, c = e % p;
                        if (c > a) continue;
                        const int pa = perm[a], pc = perm[c];
                        G[a + c * P] = (pa >= pc) ? T[pa + pc * P] : T[pc + pa * P];
                    }
                    for (int a = b.tid; a < p; a += b.nt) g[a] = z[perm[a]];
                    blk_sync();
                };
                order_columns();
                apply_order();
                dd_chol_skip(b, G, p, P, kept, nullptr, TSFA_DD_SKIP_TOL, diag0, skip_rel);
                if (dd_needs_svd(b, G, p, P, kept, diag0, skip_rel, ev)) {
                    // pinv truncation regime (see dd_hestenes): shifted design, target x[t] = x'[t] + c
                    build_ar(xmean);
                    apply_order();
                    dd_chol_skip(b, G, p, P, kept, nullptr, TSFA_DD_SKIP_TOL, diag0, skip_rel);
                    dd_forward_kept(b, G, p, P, kept, g);
                    blk_sync();
                    if (b.tid == 0 && kept[0]) g[0] = dd_add(g[0], dd_mul_d(G[0], xmean));   // + c Q^T 1 = c L[0, :] (row 0 of L: one entry)
                    blk_sync();
                    const int r = dd_build_mt(b, G, p, P, kept, xmean, 0, [=](int a) { return a >= 1; }, g, T, ev);
                    dd_hestenes(b, T, p, r, P, ev);
                    (void)dd_pinv_from_svd(b, T, p, r, P, ev, p, 0, z, beta);
                    ok = true;
                } else {
                    dd_forward_kept(b, G, p, P, kept, g);
                    double unused = 0.0;
                    // (dd_min_norm uses kept2 as scratch: the order moves to skip_rel's place first)
                    blk_sync();
                    if (b.tid == 0) for (int a = 0; a < p; ++a) skip_rel[a] = (double)perm[a];
                    blk_sync();
                    ok = dd_min_norm(b, G, p, P, kept, g, T, z, diag0, kept2, beta, false, &unused);
                    blk_sync();
                    if (b.tid == 0) for (int a = 0; a < p; ++a) perm[a] = (int)skip_rel[a];
                    blk_sync();
                }
                // coefficients back to the design's column order
                blk_sync();
                for (int a = b.tid; a < p; a += b.nt) z[a] = beta[a];
                blk_sync();
                for (int a = b.tid; a < p; a += b.nt) beta[perm[a]] = z[a];
                blk_sync();
                done_k = k;
            }
            if (b.tid == 0) out_row[sp.col] = ok ? beta[coeff].hi : TSFA_NAN;
        }
    }
}

#endif
