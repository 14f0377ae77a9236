// Family AR: calculators whose arithmetic lives in statsmodels in the reference (fc.py:387, 440, 499, 1459):
// adjusted autocovariance, Levinson-Durbin PACF, conditional-OLS AR(k), augmented Dickey-Fuller with AIC lag
// selection.  Restated from statsmodels/tsa/stattools.py (acovf, acf, pacf, levinson_durbin, adfuller, _autolag),
// statsmodels/tsa/ar_model.py (AutoReg = OLS on lagged regressors) and statsmodels/tsa/adfvalues.py (mackinnonp).
#ifndef TSFA_FAM_AR_H
#define TSFA_FAM_AR_H

#include "tsfa_common.h"

#define TSFA_AR_MAXP 32  // max regressors of any OLS solved here (ADF at n = 65536 needs 63: rejected at plan time via max_len)

// Cholesky factorization G = L L^T (lower, in place in the lower triangle of G, ld = p).  Serial.  Returns false
// if a pivot is not positive.
TSFA_DEV bool chol_factor(double *G, int p, int ld) {
    for (int j = 0; j < p; ++j) {
        double d = G[j + j * ld];
        for (int k = 0; k < j; ++k) d -= G[j + k * ld] * G[j + k * ld];
        if (!(d > 0.0)) return false;
        d = sqrt(d);
        G[j + j * ld] = d;
        for (int i = j + 1; i < p; ++i) {
            double s = G[i + j * ld];
            for (int k = 0; k < j; ++k) s -= G[i + k * ld] * G[j + k * ld];
            G[i + j * ld] = s / d;
        }
    }
    return true;
}
TSFA_DEV void chol_solve(const double *L, int p, int ld, const double *rhs, double *x) {
    for (int i = 0; i < p; ++i) {  // L w = rhs
        double s = rhs[i];
        for (int k = 0; k < i; ++k) s -= L[i + k * ld] * x[k];
        x[i] = s / L[i + i * ld];
    }
    for (int i = p - 1; i >= 0; --i) {  // L^T x = w
        double s = x[i];
        for (int k = i + 1; k < p; ++k) s -= L[k + i * ld] * x[k];
        x[i] = s / L[i + i * ld];
    }
}

// Regressor j of row t for the two designs used here.
//   AR   : z_0 = 1, z_j = xc[t - j] (j = 1..k), target xc[t], rows t = k .. n-1
//   ADF  : see adf_reg
struct ArDesign {
    const double *xc;
    TSFA_MEM double reg(int j, int t) const { return j == 0 ? 1.0 : xc[t - j]; }
};
// ADF rows are indexed by t in [lag0, n-2] with d[t] = x[t+1] - x[t]:
//   z_0 = 1, z_1 = xc[t] (level), z_j = d[t - (j-1)] for j >= 2, target d[t]
struct AdfDesign {
    const double *xc;
    TSFA_MEM double dif(int t) const { return xc[t + 1] - xc[t]; }
    TSFA_MEM double reg(int j, int t) const { return j == 0 ? 1.0 : (j == 1 ? xc[t] : dif(t - (j - 1))); }
};

// cooperative Gram matrix: G[a + b*ld] = sum_t z_a z_b (a >= b), g[a] = sum_t z_a y, yy = sum_t y^2
template <class D, class Y>
TSFA_DEV double blk_gram(const Blk &b, const D &dz, Y y, int p, int t0, int t1, double *G, int ld, double *g) {
    for (int a = 0; a < p; ++a) {
        for (int c = 0; c <= a; ++c) {
            double s = 0.0;
            for (int t = t0 + b.tid; t < t1; t += b.nt) s += dz.reg(a, t) * dz.reg(c, t);
            s = blk_sum(b, s);
            if (b.tid == 0) G[a + c * ld] = s;
        }
        double s = 0.0;
        for (int t = t0 + b.tid; t < t1; t += b.nt) s += dz.reg(a, t) * y(t);
        s = blk_sum(b, s);
        if (b.tid == 0) g[a] = s;
    }
    double s = 0.0;
    for (int t = t0 + b.tid; t < t1; t += b.nt) s += y(t) * y(t);
    s = blk_sum(b, s);
    blk_sync();
    return s;
}

// one step of iterative refinement of beta for the normal equations (L = chol factor), returns SSR of the refined fit
template <class D, class Y>
TSFA_DEV double blk_refine(const Blk &b, const D &dz, Y y, int p, int t0, int t1, const double *L, int ld,
                           double *beta, double *gr, double *delta, int nrefine) {
    for (int it = 0; it < nrefine; ++it) {
        for (int a = 0; a < p; ++a) {
            double s = 0.0;
            for (int t = t0 + b.tid; t < t1; t += b.nt) {
                double r = y(t);
                for (int c = 0; c < p; ++c) r -= dz.reg(c, t) * beta[c];
                s += dz.reg(a, t) * r;
            }
            s = blk_sum(b, s);
            if (b.tid == 0) gr[a] = s;
        }
        blk_sync();
        if (b.tid == 0) {
            chol_solve(L, p, ld, gr, delta);
            for (int a = 0; a < p; ++a) beta[a] += delta[a];
        }
        blk_sync();
    }
    double s = 0.0;
    for (int t = t0 + b.tid; t < t1; t += b.nt) {
        double r = y(t);
        for (int c = 0; c < p; ++c) r -= dz.reg(c, t) * beta[c];
        s += r * r;
    }
    return blk_sum(b, s);
}

// statsmodels.tsa.adfvalues.mackinnonp(teststat, regression="c", N=1)
TSFA_DEV double mackinnon_p_c1(double t) {
    if (t > 2.74) return 1.0;
    if (t < -18.83) return 0.0;
    double z;
    if (t <= -1.61) z = 2.1659 + t * (1.4412 + t * (3.8269 * 1e-2));
    else z = 1.7339 + t * (9.3202 * 1e-1 + t * (-1.2745 * 1e-1 + t * (-1.0368 * 1e-2)));
    return tsfa_norm_cdf(z);
}

// Evaluate the AR specs of one series.
//   xc : LDS, n doubles (will hold the mean-centred series)
//   aw : LDS, >= 2*TSFA_AR_MAXP*TSFA_AR_MAXP + 8*TSFA_AR_MAXP + 128 doubles
TSFA_DEV void fam_ar_series(const Blk &b, const double *xs, int n, const TsfaSpec *specs, int nspecs,
                            double *out_row, double *xc, double *aw) {
    const double dn = (double)n;
    double sm = 0.0;
    for (int i = b.tid; i < n; i += b.nt) sm += xs[i];
    const double mean = blk_sum(b, sm) / dn;
    blk_sync();
    for (int i = b.tid; i < n; i += b.nt) xc[i] = xs[i] - mean;
    blk_sync();
    double v0 = 0.0;
    for (int i = b.tid; i < n; i += b.nt) v0 += xc[i] * xc[i];
    const double ss0 = blk_sum(b, v0);
    const double var = ss0 / dn;

    const int P = TSFA_AR_MAXP;
    double *G = aw;                 // P*P
    double *G2 = G + P * P;         // P*P
    double *g = G2 + P * P;         // P
    double *beta = g + P;           // P
    double *tmp1 = beta + P;        // P
    double *tmp2 = tmp1 + P;        // P
    double *acv = tmp2 + P;         // 64: adjusted autocovariances
    double *res = acv + 64;         // 16: results broadcast
    double *pac = res + 16;         // 48

    // which shared pieces are needed
    int max_acf_lag = -1, max_pacf_lag = -1;
    bool need_adf = false;
    for (int s = 0; s < nspecs; ++s) {
        const int c = specs[s].calc;
        if (c == TSFA_C_AGG_AUTOCORRELATION) {
            const int ml = (int)specs[s].p[1];
            if (ml > max_acf_lag) max_acf_lag = ml;
        } else if (c == TSFA_C_PARTIAL_AUTOCORRELATION) {
            const int l = (int)specs[s].p[0];
            if (l > max_pacf_lag) max_pacf_lag = l;
        } else if (c == TSFA_C_AUGMENTED_DICKEY_FULLER) {
            need_adf = true;
        }
    }
    if (max_acf_lag > 60) max_acf_lag = 60;
    if (max_pacf_lag > 40) max_pacf_lag = 40;

    // ---- adjusted autocovariance acv[k] = sum_t xc[t] xc[t+k] / (n - k)  (stattools.acovf, adjusted=True) ----
    int pacf_maxlag = 0;  // lags actually computed by pacf
    if (max_pacf_lag >= 0 && n > 1) {
        pacf_maxlag = (max_pacf_lag >= n / 2) ? (n / 2 - 1) : max_pacf_lag;  // fc.py:472-475
    }
    int nacv = -1;
    if (max_acf_lag >= 0) nacv = (max_acf_lag < n - 1) ? max_acf_lag : (n - 1);
    if (pacf_maxlag > nacv) nacv = pacf_maxlag;
    for (int k = 0; k <= nacv; ++k) {
        double s = 0.0;
        for (int t = b.tid; t < n - k; t += b.nt) s += xc[t] * xc[t + k];
        s = blk_sum(b, s);
        if (b.tid == 0) acv[k] = s / (double)(n - k);
    }
    blk_sync();

    // ---- PACF by Levinson-Durbin (stattools.levinson_durbin, isacov=True) ----
    if (max_pacf_lag >= 0) {
        if (b.tid == 0) {
            for (int k = 0; k <= max_pacf_lag; ++k) pac[k] = TSFA_NAN;
            if (n > 1 && pacf_maxlag > 0) {
                const int ord = pacf_maxlag;
                double *phi = G;  // (ord+1) x (ord+1), phi[j + k*ld]
                const int ld = ord + 1;
                double *sig = tmp1;
                for (int i = 0; i < ld * ld; ++i) phi[i] = 0.0;
                phi[1 + 1 * ld] = acv[1] / acv[0];
                sig[1] = acv[0] - phi[1 + 1 * ld] * acv[1];
                for (int k = 2; k <= ord; ++k) {
                    double dot = 0.0;
                    for (int j = 1; j < k; ++j) dot += phi[j + (k - 1) * ld] * acv[k - j];
                    phi[k + k * ld] = (acv[k] - dot) / sig[k - 1];
                    for (int j = 1; j < k; ++j)
                        phi[j + k * ld] = phi[j + (k - 1) * ld] - phi[k + k * ld] * phi[(k - j) + (k - 1) * ld];
                    sig[k] = sig[k - 1] * (1.0 - phi[k + k * ld] * phi[k + k * ld]);
                }
                pac[0] = 1.0;
                for (int k = 1; k <= ord; ++k) pac[k] = phi[k + k * ld];
            }
        }
        blk_sync();
    }

    // ---- augmented Dickey-Fuller, regression="c", autolag="AIC" (stattools.adfuller) ----
    double adf_stat = TSFA_NAN, adf_p = TSFA_NAN, adf_lag = TSFA_NAN;
    if (need_adf) {
        int maxlag = (int)ceil(12.0 * pow(dn / 100.0, 0.25));
        const int cap = n / 2 - 1 - 1;
        if (cap < maxlag) maxlag = cap;
        if (maxlag >= 0 && maxlag + 2 <= P) {
            AdfDesign dz{xc};
            const double *xcc = xc;
            auto yv = [=](int t) { return xcc[t + 1] - xcc[t]; };
            // step 1: all nested fits on rows t = maxlag .. n-2 from one Cholesky factorization
            const int p1 = maxlag + 2;
            const int t0 = maxlag, t1 = n - 1;
            const double nobs = (double)(t1 - t0);
            const double yy = blk_gram(b, dz, yv, p1, t0, t1, G, P, g);
            if (b.tid == 0) {
                int best = -1;
                double best_aic = 0.0;
                const bool ok = chol_factor(G, p1, P);
                if (ok) {
                    // w = L^-1 g ;  SSR_p = yy - sum_{i<p} w_i^2
                    double acc = 0.0;
                    for (int i = 0; i < p1; ++i) {
                        double s = g[i];
                        for (int k = 0; k < i; ++k) s -= G[i + k * P] * tmp1[k];
                        tmp1[i] = s / G[i + i * P];
                        acc += tmp1[i] * tmp1[i];
                        const int pcols = i + 1;
                        if (pcols >= 2) {
                            const double ssr = yy - acc;
                            // OLS.aic = -2 llf + 2 k,  llf = -nobs/2 (log(2 pi) + log(ssr/nobs) + 1)
                            const double llf = -0.5 * nobs * log(2.0 * M_PI) - 0.5 * nobs * log(ssr / nobs) - 0.5 * nobs;
                            const double aic = -2.0 * llf + 2.0 * (double)pcols;
                            if (best < 0 || aic < best_aic) {
                                best = pcols;
                                best_aic = aic;
                            }
                        }
                    }
                }
                res[0] = (double)best;
            }
            blk_sync();
            const int bestcols = (int)res[0];
            blk_sync();
            if (bestcols >= 2) {
                const int usedlag = bestcols - 2;
                // step 2: final regression on rows t = usedlag .. n-2
                const int p2 = usedlag + 2;
                const int u0 = usedlag, u1 = n - 1;
                const double nobs2 = (double)(u1 - u0);
                blk_gram(b, dz, yv, p2, u0, u1, G2, P, g);
                if (b.tid == 0) {
                    const bool ok = chol_factor(G2, p2, P);
                    res[1] = ok ? 1.0 : 0.0;
                    if (ok) chol_solve(G2, p2, P, g, beta);
                }
                blk_sync();
                const bool ok2 = res[1] != 0.0;
                blk_sync();
                if (ok2) {
                    const double ssr = blk_refine(b, dz, yv, p2, u0, u1, G2, P, beta, tmp1, tmp2, 1);
                    if (b.tid == 0) {
                        // (X'X)^-1 [level, level]: solve G u = e_1
                        for (int i = 0; i < p2; ++i) tmp1[i] = (i == 1) ? 1.0 : 0.0;
                        chol_solve(G2, p2, P, tmp1, tmp2);
                        const double sigma2 = ssr / (nobs2 - (double)p2);
                        const double se = sqrt(sigma2 * tmp2[1]);
                        const double tstat = beta[1] / se;
                        res[2] = tstat;
                        res[3] = mackinnon_p_c1(tstat);
                        res[4] = (double)usedlag;
                    }
                    blk_sync();
                    adf_stat = res[2];
                    adf_p = res[3];
                    adf_lag = res[4];
                    if (adf_stat != adf_stat) adf_p = TSFA_NAN;
                    blk_sync();
                }
            }
        }
    }

    for (int s = 0; s < nspecs; ++s) {
        const TsfaSpec sp = specs[s];
        double v = TSFA_NAN;
        switch (sp.calc) {
        case TSFA_C_AGG_AUTOCORRELATION: {                               // fc.py:387
            const int agg = (int)sp.p[0];
            int ml = (int)sp.p[1];
            double r = TSFA_NAN;
            if (b.tid == 0) {
                if (fabs(var) < 1e-10 || n == 1) {
                    r = 0.0;  // f_agg over zeros
                } else {
                    int len = (ml < n - 1) ? ml : (n - 1);  // a = acf[1:], a[:maxlag]
                    if (len > 60) len = 60;
                    double *a = tmp1;  // <= 60 entries spill into tmp2.. (contiguous)
                    for (int k = 0; k < len; ++k) a[k] = acv[k + 1] / acv[0];
                    if (len <= 0) {
                        r = TSFA_NAN;
                    } else if (agg == TSFA_AGG_MEAN) {
                        r = np_leaf_sum(0, len, [=](int i) { return a[i]; }) / (double)len;
                    } else if (agg == TSFA_AGG_VAR) {
                        const double m = np_leaf_sum(0, len, [=](int i) { return a[i]; }) / (double)len;
                        r = np_leaf_sum(0, len, [=](int i) { const double d = a[i] - m; return d * d; }) / (double)len;
                    } else {  // median
                        for (int i = 1; i < len; ++i) {
                            const double key = a[i];
                            int j = i - 1;
                            while (j >= 0 && a[j] > key) {
                                a[j + 1] = a[j];
                                --j;
                            }
                            a[j + 1] = key;
                        }
                        r = (len & 1) ? a[(len - 1) / 2] : (0.0 + a[len / 2 - 1] + a[len / 2]) / 2.0;
                    }
                }
            }
            v = blk_bcast0(b, r);
        } break;
        case TSFA_C_PARTIAL_AUTOCORRELATION: {                           // fc.py:440
            const int l = (int)sp.p[0];
            v = (l >= 0 && l <= max_pacf_lag) ? pac[l] : TSFA_NAN;
        } break;
        case TSFA_C_AUGMENTED_DICKEY_FULLER: {                           // fc.py:499
            const int attr = (int)sp.p[0];
            v = (attr == TSFA_ADF_TESTSTAT) ? adf_stat : (attr == TSFA_ADF_PVALUE ? adf_p : adf_lag);
        } break;
        case TSFA_C_AR_COEFFICIENT: {                                    // fc.py:1459
            const int coeff = (int)sp.p[0], k = (int)sp.p[1];
            if (coeff > k) { v = TSFA_NAN; break; }
            // AutoReg(x, lags=k, trend="c") can only be estimated with n >= 2k + 2; on failure the reference
            // substitutes [nan]*k, so coeff < k -> NaN and coeff == k -> IndexError -> 0
            if (n < 2 * k + 2 || k + 1 > P || k < 1) {
                v = (coeff < k) ? TSFA_NAN : 0.0;
                break;
            }
            ArDesign dz{xc};
            const double *xcc = xc;
            auto yv = [=](int t) { return xcc[t]; };
            const int p = k + 1;
            blk_gram(b, dz, yv, p, k, n, G, P, g);
            if (b.tid == 0) {
                const bool ok = chol_factor(G, p, P);
                res[1] = ok ? 1.0 : 0.0;
                if (ok) chol_solve(G, p, P, g, beta);
            }
            blk_sync();
            const bool ok = res[1] != 0.0;
            blk_sync();
            if (!ok) { v = TSFA_NAN; break; }
            blk_refine(b, dz, yv, p, k, n, G, P, beta, tmp1, tmp2, 1);
            if (coeff == 0) {
                // undo the centring: const = c~ + mean (1 - sum phi)
                double sphi = 0.0;
                for (int j = 1; j <= k; ++j) sphi += beta[j];
                v = beta[0] + mean * (1.0 - sphi);
            } else {
                v = beta[coeff];
            }
            blk_sync();
        } break;
        default: break;
        }
        if (b.tid == 0) out_row[sp.col] = v;
    }
}

#endif
