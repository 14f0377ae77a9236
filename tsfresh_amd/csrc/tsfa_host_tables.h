// Host-side constant tables and spec bookkeeping shared by the C-ABI (tsfa_api.cpp) and the test-only CPU
// emulation (tests/emul/emul.cpp).  Plain C++17, no HIP.
#ifndef TSFA_HOST_TABLES_H
#define TSFA_HOST_TABLES_H

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "tsfa_specs.h"

#define TSFA_DEC_KMIN_H (-324)
#define TSFA_DEC_KMAX_H 308
#define TSFA_TW_N_H 65536

struct TsfaCalcInfo {
    const char *name;
    int family;
};

static const TsfaCalcInfo tsfa_calc_table[TSFA_N_CALCS] = {
#define X(id, name, fam) {name, fam},
    TSFA_CALC_LIST(X)
#undef X
};

// correctly rounded doubles of the decimals "d e k" (strtod is correctly rounded in glibc)
static inline void tsfa_build_dectab(std::vector<double> &tab) {
    tab.resize((size_t)(TSFA_DEC_KMAX_H - TSFA_DEC_KMIN_H + 1) * 9);
    char buf[32];
    for (int k = TSFA_DEC_KMIN_H; k <= TSFA_DEC_KMAX_H; ++k)
        for (int d = 1; d <= 9; ++d) {
            snprintf(buf, sizeof buf, "%de%d", d, k);
            tab[(size_t)(k - TSFA_DEC_KMIN_H) * 9 + (d - 1)] = strtod(buf, nullptr);
        }
}

// Per-plan constants the kernels used to recompute for every series (~4 % of k_spectral and of k_cwtpeaks):
//   [0, 256)                 the periodic Hann window of scipy.signal.welch's 256-sample segments
//                            (scipy.signal.windows.general_cosine(256, [0.5, 0.5], sym=False))
//   [256 + 5 w (w - 1) + k]  Ricker tap k of width w with 10 w points (fc.py:1307 _ricker), w = 1 .. TSFA_CONSTS_MAXW
//   (offsets: TSFA_CONSTS_* in tsfa_specs.h)
static inline void tsfa_build_consts(std::vector<double> &c) {
    c.assign(TSFA_CONSTS_N, 0.0);
    for (int j = 0; j < 256; ++j) {
        // fac = np.linspace(-pi, pi, 257)[j] = j * step + start (numpy's expression)
        const double step = (M_PI - (-M_PI)) / 256.0;
        const double fac = (double)j * step + (-M_PI);
        c[TSFA_CONSTS_HANN + j] = 0.5 + 0.5 * cos(fac);
    }
    for (int w = 1; w <= TSFA_CONSTS_MAXW; ++w) {
        const int points = 10 * w;
        const double a = (double)w;
        const double A = 2.0 / (sqrt(3.0 * a) * pow(M_PI, 0.25));
        const double wsq = a * a;
        for (int k = 0; k < points; ++k) {
            const double vec = (double)k - ((double)points - 1.0) / 2.0;
            const double xsq = vec * vec;
            c[TSFA_CONSTS_RICKER + 5 * w * (w - 1) + k] = A * (1.0 - xsq / wsq) * exp(-xsq / (2.0 * wsq));
        }
    }
}

// twc[j] = cos(2 pi j / N), tws[j] = -sin(2 pi j / N), j < N/2, N = 65536; octant symmetry keeps the table exact
// under the symmetries a radix-2 FFT relies on
static inline void tsfa_build_twiddles(std::vector<double> &twc, std::vector<double> &tws) {
    const int N = TSFA_TW_N_H;
    twc.resize(N / 2);
    tws.resize(N / 2);
    const long double PI_L = 3.141592653589793238462643383279502884L;
    for (int j = 0; j < N / 2; ++j) {
        // angle = 2 pi j / N in [0, pi); reduce to [0, pi/4] by symmetry
        int q = j;
        long double c, s;
        const int quarter = N / 4, eighth = N / 8;
        bool neg_c = false;
        if (q > quarter) {  // (pi/2, pi): cos(pi - a) = -cos(a), sin(pi - a) = sin(a)
            q = N / 2 - q;
            neg_c = true;
        }
        if (q <= eighth) {
            const long double a = 2.0L * PI_L * (long double)q / (long double)N;
            c = cosl(a);
            s = sinl(a);
        } else {  // (pi/4, pi/2]: cos(a) = sin(pi/2 - a)
            const long double a = 2.0L * PI_L * (long double)(quarter - q) / (long double)N;
            c = sinl(a);
            s = cosl(a);
        }
        if (neg_c) c = -c;
        twc[j] = (double)c;
        tws[j] = (double)(-s);
    }
}

// Per-family facts about a plan's spec list that the kernels would otherwise have to find by scanning the list
// (a serial scan costs one scalar-memory round trip per spec on the device).  May reorder `specs` (each spec
// carries its output column, so the order on the device is free).
#define TSFA_ALT_SLOTS 16
struct TsfaFamHints {
    int a = 0, b = 0, c = 0, d = -1, e = 0;
    TsfaAltPlan alt;  // BASIC
    TsfaCqPlan cq;    // SORT
};
// BASIC columns whose value is a closed form of the per-series statistics or a read of a cached result: they are
// moved behind the other columns and evaluated with lane = column (fam_basic.h: basic_epilogue) instead of one
// ~450-cycle trip through the column loop each.
static inline bool tsfa_basic_is_closed_form(int calc) {
    switch (calc) {
    case TSFA_C_SUM_VALUES: case TSFA_C_MEAN: case TSFA_C_LENGTH: case TSFA_C_STANDARD_DEVIATION: case TSFA_C_VARIANCE:
    case TSFA_C_ROOT_MEAN_SQUARE: case TSFA_C_MAXIMUM: case TSFA_C_ABSOLUTE_MAXIMUM: case TSFA_C_MINIMUM:
    case TSFA_C_ABS_ENERGY: case TSFA_C_VARIATION_COEFFICIENT: case TSFA_C_VAR_GT_STD: case TSFA_C_LARGE_STD:
    case TSFA_C_FIRST_LOCATION_OF_MAXIMUM: case TSFA_C_LAST_LOCATION_OF_MAXIMUM: case TSFA_C_FIRST_LOCATION_OF_MINIMUM:
    case TSFA_C_LAST_LOCATION_OF_MINIMUM: case TSFA_C_HAS_DUPLICATE_MAX: case TSFA_C_HAS_DUPLICATE_MIN:
    case TSFA_C_QUERY_SIMILARITY_COUNT:
        return true;
    default:
        return false;
    }
}

static inline void tsfa_prepare_family_impl(int fam, std::vector<TsfaSpec> &specs, TsfaFamHints &h);

static inline void tsfa_prepare_family(int fam, std::vector<TsfaSpec> &specs, TsfaFamHints &h) {
    tsfa_prepare_family_impl(fam, specs, h);
    if (fam == TSFA_FAM_BASIC) {  // bit 16 of a: no column reads the positions / multiplicities of the extrema
        bool want_loc = false;
        for (const auto &s : specs)
            if (s.calc == TSFA_C_FIRST_LOCATION_OF_MAXIMUM || s.calc == TSFA_C_LAST_LOCATION_OF_MAXIMUM ||
                s.calc == TSFA_C_FIRST_LOCATION_OF_MINIMUM || s.calc == TSFA_C_LAST_LOCATION_OF_MINIMUM ||
                s.calc == TSFA_C_HAS_DUPLICATE_MAX || s.calc == TSFA_C_HAS_DUPLICATE_MIN)
                want_loc = true;
        if (!want_loc) h.a |= (1 << 16);
    }
    if (fam == TSFA_FAM_AR) {
        // d = columns of the column loop; behind them the reads of results the prologue / an earlier column left in
        // LDS (pacf lags, the three ADF outputs, the other coefficients of an AR fit), evaluated with lane = column.
        // AR coefficients only when the plan fits a single order k (the cache holds one fit).
        int ar_k = -1;
        bool one_k = true;
        for (const auto &s : specs)
            if (s.calc == TSFA_C_AR_COEFFICIENT) {
                if (ar_k >= 0 && (int)s.p[1] != ar_k) one_k = false;
                ar_k = (int)s.p[1];
            }
        std::vector<TsfaSpec> loop, epi;
        bool seen_ar = false;
        for (const auto &s : specs) {
            bool e = (s.calc == TSFA_C_PARTIAL_AUTOCORRELATION || s.calc == TSFA_C_AUGMENTED_DICKEY_FULLER);
            // (the fit is made -- and cached for the epilogue -- by the first column that READS it: coeff > k is NaN without
            //  one, fc.py:1500; a plan whose first column was such a one left the cache empty for the others -- a fuzz find)
            if (s.calc == TSFA_C_AR_COEFFICIENT) { e = one_k && seen_ar; if ((int)s.p[0] <= (int)s.p[1]) seen_ar = true; }
            (e ? epi : loop).push_back(s);
        }
        h.d = (int)loop.size();
        specs = loop;
        specs.insert(specs.end(), epi.begin(), epi.end());
        return;
    }
    if (fam == TSFA_FAM_SORT) {
        // c = columns of the column loop; behind them the ones basic_epilogue's sibling sort_epilogue evaluates with
        // lane = column: order statistics read straight from the sorted copy, and reads of the corridor / run /
        // symmetry caches that an earlier loop column fills
        std::vector<TsfaSpec> loop, epi;
        bool seen_sym = false, seen_runs = false;
        for (const auto &s : specs) {
            bool e = (s.calc == TSFA_C_MEDIAN || s.calc == TSFA_C_QUANTILE);
            if (s.calc == TSFA_C_CHANGE_QUANTILES)
                e = (s.p[1] == -2.0) ? ((int)s.p[0] < 128) : (h.cq.n > 0 && s.p[0] >= s.p[1]);
            if (s.calc == TSFA_C_SYMMETRY_LOOKING) { e = seen_sym; seen_sym = true; }
            if (s.calc == TSFA_C_HAS_DUPLICATE || s.calc == TSFA_C_RATIO_VALUE_NUMBER || s.calc == TSFA_C_PCT_REOCC_VALUES ||
                s.calc == TSFA_C_PCT_REOCC_DATAPOINTS || s.calc == TSFA_C_SUM_REOCC_VALUES ||
                s.calc == TSFA_C_SUM_REOCC_DATA_POINTS) { e = seen_runs; seen_runs = true; }
            (e ? epi : loop).push_back(s);
        }
        // Langevin fits: fam_sort_series keeps the LAST fit's coefficients, so the columns of one (m, r) are made
        // neighbours (in their own slots of the column loop; a spec carries its output column) -- every distinct (m, r)
        // is then fitted once per series, and at most e = #distinct records per series reach the second pass
        {
            auto is_lv = [](const TsfaSpec &s) { return s.calc == TSFA_C_FRIEDRICH_COEFFICIENTS || s.calc == TSFA_C_MAX_LANGEVIN_FIXED_POINT; };
            auto key = [](const TsfaSpec &s) {
                return s.calc == TSFA_C_FRIEDRICH_COEFFICIENTS ? std::make_pair((int)s.p[1], (int)s.p[2]) : std::make_pair((int)s.p[0], (int)s.p[1]);
            };
            std::vector<std::pair<int, int>> order;   // distinct (m, r) by first appearance
            std::vector<TsfaSpec> lv;
            for (const auto &s : loop)
                if (is_lv(s)) {
                    lv.push_back(s);
                    if (std::find(order.begin(), order.end(), key(s)) == order.end()) order.push_back(key(s));
                }
            std::stable_sort(lv.begin(), lv.end(), [&](const TsfaSpec &x, const TsfaSpec &y) {
                return std::find(order.begin(), order.end(), key(x)) < std::find(order.begin(), order.end(), key(y));
            });
            size_t k = 0;
            for (auto &s : loop)
                if (is_lv(s)) s = lv[k++];
            h.e = (int)order.size();
        }
        // d = (stride << 8) | set of dimensions, when every permutation_entropy column of the plan has the same stride and
        // the dimensions are exactly 3 .. 7 (ComprehensiveFCParameters, fam_perm.h TSFA_PE_MASK): k_perm evaluates them all
        // from one sweep of the windows and k_sort skips them
        {
            int tau = -1, n_pe = 0;
            unsigned mask = 0;
            bool ok = true;
            for (const auto &s : loop)
                if (s.calc == TSFA_C_PERMUTATION_ENTROPY) {
                    const int t = (int)s.p[0], D = (int)s.p[1];
                    if (tau < 0) tau = t;
                    if (t != tau || D < 2 || D > 7 || t < 1 || t > 0x7FFF) ok = false;
                    else mask |= 1u << D;
                    ++n_pe;
                }
            h.d = (ok && n_pe >= 2 && mask == 0xF8u) ? (int)((unsigned)tau << 8 | mask) : 0;   // (fam_perm.h TSFA_PE_MASK)
        }
        h.c = (int)loop.size();
        // a = doubles of LDS scratch the plan needs (fam_sort.h friedrich_coeffs: 6 r + 16 + r (m + 1)), at least 320:
        // the ordinal-pattern histogram adapts to it
        h.a = 320;
        for (const auto &s : loop) {
            int m = 0, r = 0;
            if (s.calc == TSFA_C_FRIEDRICH_COEFFICIENTS) { m = (int)s.p[1]; r = (int)s.p[2]; }
            else if (s.calc == TSFA_C_MAX_LANGEVIN_FIXED_POINT) { m = (int)s.p[0]; r = (int)s.p[1]; }
            if (r > 0 && r <= 64 && m >= 1 && m <= 3) {
                h.a = std::max(h.a, 6 * r + 16 + r * (m + 1) + 8);
                h.b = std::max(h.b, r);   // b = largest bin count of a Langevin fit: sizes the records of the second pass
            }
        }
        specs = loop;
        specs.insert(specs.end(), epi.begin(), epi.end());
        return;
    }
    if (fam != TSFA_FAM_BASIC && fam != TSFA_FAM_TREND) return;
    // c = number of columns that stay in the column loop; the rest (closed forms, reads of the agg_linear_trend /
    // linear_trend / index_mass_quantile caches that an earlier loop column fills) follow and go to the epilogue
    std::vector<TsfaSpec> loop, epi;
    bool seen_lt = false, seen_ltt = false;
    for (const auto &s : specs) {
        bool e = tsfa_basic_is_closed_form(s.calc);
        if (s.calc == TSFA_C_AGG_LINEAR_TREND) e = (h.alt.nkeys > 0) && ((int)s.p[3] < 128);
        if (s.calc == TSFA_C_INDEX_MASS_QUANTILE) e = (h.alt.nq > 0) && (s.p[2] == 1.0) && ((int)s.p[1] < 128);
        if (s.calc == TSFA_C_LINEAR_TREND) { e = seen_lt; seen_lt = true; }
        if (s.calc == TSFA_C_LINEAR_TREND_TIMEWISE) { e = seen_ltt; seen_ltt = true; }
        (e ? epi : loop).push_back(s);
    }
    // d = count-type columns, moved to the front of the column loop: fam_basic.h (basic_count_pass) evaluates them
    // together from registers (at most 256: one LDS counter each)
    {
        std::vector<TsfaSpec> cnt, rest;
        for (const auto &s : loop) {
            const bool c = (s.calc == TSFA_C_RATIO_BEYOND_R_SIGMA || s.calc == TSFA_C_COUNT_ABOVE_MEAN ||
                            s.calc == TSFA_C_COUNT_BELOW_MEAN || s.calc == TSFA_C_COUNT_ABOVE || s.calc == TSFA_C_COUNT_BELOW ||
                            s.calc == TSFA_C_VALUE_COUNT || s.calc == TSFA_C_RANGE_COUNT || s.calc == TSFA_C_NUMBER_CROSSING_M);
            ((c && cnt.size() < 256) ? cnt : rest).push_back(s);
        }
        h.d = (int)cnt.size();
        // e = sum-type columns, next in line (basic_sum_pass)
        std::vector<TsfaSpec> sums, rest2;
        for (const auto &s : rest) {
            const bool c = (s.calc == TSFA_C_AUTOCORRELATION || s.calc == TSFA_C_C3 ||
                            s.calc == TSFA_C_TIME_REVERSAL_ASYMMETRY_STATISTIC || s.calc == TSFA_C_ENERGY_RATIO_BY_CHUNKS ||
                            s.calc == TSFA_C_CID_CE || s.calc == TSFA_C_MEAN_ABS_CHANGE ||
                            s.calc == TSFA_C_ABSOLUTE_SUM_OF_CHANGES || s.calc == TSFA_C_SKEWNESS || s.calc == TSFA_C_KURTOSIS);
            (c ? sums : rest2).push_back(s);
        }
        h.e = (int)sums.size();
        loop = cnt;
        loop.insert(loop.end(), sums.begin(), sums.end());
        loop.insert(loop.end(), rest2.begin(), rest2.end());
    }
    h.c = (int)loop.size();
    specs = loop;
    specs.insert(specs.end(), epi.begin(), epi.end());
}

static inline void tsfa_prepare_family_impl(int fam, std::vector<TsfaSpec> &specs, TsfaFamHints &h) {
    h = TsfaFamHints();
    memset(&h.alt, 0, sizeof h.alt);
    memset(&h.cq, 0, sizeof h.cq);
    if (fam == TSFA_FAM_AR) {
        // a: largest agg_autocorrelation maxlag, b: largest partial_autocorrelation lag (-1: none), c: ADF requested
        h.a = -1;
        h.b = -1;
        for (const auto &s : specs) {
            if (s.calc == TSFA_C_AGG_AUTOCORRELATION && (int)s.p[1] > h.a) h.a = (int)s.p[1];
            if (s.calc == TSFA_C_PARTIAL_AUTOCORRELATION && (int)s.p[0] > h.b) h.b = (int)s.p[0];
            // bit 0: requested; bits 1-2: TSFA_AUTOLAG_* (tsfa_validate_plan: one per plan).  A column whose attr code is 3
            // is NaN whatever the fit (an unknown attr, or an autolag value statsmodels rejects): it names no lag selection
            if (s.calc == TSFA_C_AUGMENTED_DICKEY_FULLER && (int)s.p[0] != 3) h.c = 1 + 2 * (int)s.p[1];
            if (s.calc == TSFA_C_AUGMENTED_DICKEY_FULLER && (int)s.p[0] == 3 && h.c == 0) h.c = 1;
        }
    }
    if (fam == TSFA_FAM_SORT) {
        std::vector<std::pair<double, double>> cor;
        for (const auto &s : specs)
            if (s.calc == TSFA_C_CHANGE_QUANTILES && s.p[0] < s.p[1]) {
                const std::pair<double, double> k(s.p[0], s.p[1]);
                if (std::find(cor.begin(), cor.end(), k) == cor.end()) cor.push_back(k);
            }
        if (!cor.empty() && cor.size() <= TSFA_CQ_MAX) {
            h.cq.n = (int)cor.size();
            for (size_t k = 0; k < cor.size(); ++k) { h.cq.ql[k] = cor[k].first; h.cq.qh[k] = cor[k].second; }
            bool first = true;
            for (auto &s : specs) {
                if (s.calc != TSFA_C_CHANGE_QUANTILES || !(s.p[0] < s.p[1])) continue;
                const std::pair<double, double> k(s.p[0], s.p[1]);
                const int idx = (int)(std::find(cor.begin(), cor.end(), k) - cor.begin());
                s.p[0] = (double)(idx + (first ? 128 : 0));
                s.p[1] = -2.0;
                first = false;
            }
        }
    }
    if (fam == TSFA_FAM_SPECTRAL) {
        // a: bit 0 = full-length rfft needed, bit 1 = Welch PSD needed;  b: number of leading Welch-based specs
        std::vector<TsfaSpec> lead, rest;
        for (const auto &s : specs) {
            if (s.calc == TSFA_C_SPKT_WELCH_DENSITY || s.calc == TSFA_C_FOURIER_ENTROPY) lead.push_back(s);
            else rest.push_back(s);
        }
        h.a = (rest.empty() ? 0 : 1) | (lead.empty() ? 0 : 2);
        h.b = (int)lead.size();
        specs = lead;
        specs.insert(specs.end(), rest.begin(), rest.end());
    } else if (fam == TSFA_FAM_BASIC || fam == TSFA_FAM_TREND) {  // BASIC: a;  TREND: b, alt
        // a: largest number_peaks support <= 254;  b: 1 if any agg_linear_trend column asks for the p-value
        // agg_linear_trend: p[3] = cache slot of the column's (f_agg, chunk_len) regression, + 64 if this column is
        // the one that has to compute it (the slots are simulated here, round-robin over TSFA_ALT_SLOTS)
        // index_mass_quantile: all distinct q at once (lane = q)
        {
            std::vector<double> qs;
            for (const auto &s : specs)
                if (s.calc == TSFA_C_INDEX_MASS_QUANTILE && std::find(qs.begin(), qs.end(), s.p[0]) == qs.end()) qs.push_back(s.p[0]);
            if (!qs.empty() && qs.size() <= TSFA_ALT_MAXKEYS) {
                h.alt.nq = (int)qs.size();
                for (size_t k = 0; k < qs.size(); ++k) h.alt.q[k] = qs[k];
                bool first = true;
                for (auto &s : specs) {
                    if (s.calc != TSFA_C_INDEX_MASS_QUANTILE) continue;
                    const int idx = (int)(std::find(qs.begin(), qs.end(), s.p[0]) - qs.begin());
                    s.p[1] = (double)(idx + (first ? 128 : 0));
                    s.p[2] = 1.0;
                    first = false;
                }
            }
        }
        {   // no column left that needs the n-double work array?  (per-column index_mass_quantile / agg_linear_trend do)
            bool imq_plain = false, has_alt = false;
            std::vector<std::pair<int, int>> akeys;
            for (const auto &s : specs) {
                if (s.calc == TSFA_C_INDEX_MASS_QUANTILE && s.p[2] != 1.0) imq_plain = true;
                if (s.calc == TSFA_C_AGG_LINEAR_TREND) {
                    has_alt = true;
                    const std::pair<int, int> k((int)s.p[1], (int)s.p[2]);
                    if (std::find(akeys.begin(), akeys.end(), k) == akeys.end()) akeys.push_back(k);
                }
            }
            h.alt.small_w = (!imq_plain && (!has_alt || akeys.size() <= TSFA_ALT_MAXKEYS)) ? 1 : 0;
        }
        // preferred: all distinct (chunk_len, f_agg) keys at once (TsfaAltPlan); p[3] = key index, + 128 on the first
        // agg_linear_trend column, which computes them all
        {
            std::vector<std::pair<int, int>> keys;
            for (const auto &s : specs)
                if (s.calc == TSFA_C_AGG_LINEAR_TREND) {
                    const std::pair<int, int> k((int)s.p[1], (int)s.p[2]);
                    if (std::find(keys.begin(), keys.end(), k) == keys.end()) keys.push_back(k);
                    if ((int)s.p[0] == TSFA_ATTR_PVALUE) h.alt.want_p = 1;
                }
            std::sort(keys.begin(), keys.end());
            if (!keys.empty() && keys.size() <= TSFA_ALT_MAXKEYS) {
                h.alt.nkeys = (int)keys.size();
                for (size_t k = 0; k < keys.size(); ++k) { h.alt.cl[k] = keys[k].first; h.alt.agg[k] = keys[k].second; }
                bool first = true;
                for (auto &s : specs) {
                    if (s.calc == TSFA_C_NUMBER_PEAKS && (int)s.p[0] <= 254 && (int)s.p[0] > h.a) h.a = (int)s.p[0];
                    if (s.calc != TSFA_C_AGG_LINEAR_TREND) continue;
                    const std::pair<int, int> k((int)s.p[1], (int)s.p[2]);
                    const int idx = (int)(std::find(keys.begin(), keys.end(), k) - keys.begin());
                    s.p[3] = (double)(idx + (first ? 128 : 0));
                    first = false;
                }
                h.b = h.alt.want_p;
                return;
            }
        }
        int slot_key[TSFA_ALT_SLOTS], next = 0;
        for (int k = 0; k < TSFA_ALT_SLOTS; ++k) slot_key[k] = -1;
        for (auto &s : specs) {
            if (s.calc == TSFA_C_NUMBER_PEAKS && (int)s.p[0] <= 254 && (int)s.p[0] > h.a) h.a = (int)s.p[0];
            if (s.calc == TSFA_C_AGG_LINEAR_TREND) {
                if ((int)s.p[0] == TSFA_ATTR_PVALUE) h.b = 1;
                const int key = ((int)s.p[2] << 20) | (int)s.p[1];
                int slot = -1;
                for (int k = 0; k < TSFA_ALT_SLOTS; ++k)
                    if (slot_key[k] == key) slot = k;
                if (slot >= 0) {
                    s.p[3] = (double)slot;
                } else {
                    slot = next;
                    next = (next + 1) % TSFA_ALT_SLOTS;
                    slot_key[slot] = key;
                    s.p[3] = (double)(slot + 64);
                }
            }
        }
    }
}

// pywt.cwt(x, scales, "mexh") restated (pywt/_cwt.py:125-197, pywt/_functions.py:29-32,104-109; pywt 1.1.1):
//   int_psi = cumsum(mexh(linspace(-8, 8, 1024))) * step
//   taps(scale) = int_psi[floor(arange(scale*16 + 1) / (scale*step))][::-1]
//   coef = -sqrt(scale) * diff(convolve(x, taps)), centre-cropped to len(x)
// Only coef[t] for small t is consumed, which is a dot product of x[0 .. t+f+1] with one filter column:
//   W_{scale,t}[n] = -sqrt(scale) * (taps[t+f+1-n] - taps[t+f-n]),   f = floor((T-2)/2)
struct TsfaMexh {
    std::vector<double> int_psi;
    double step;
    TsfaMexh() {
        const int P = 1024;
        std::vector<double> x(P), psi(P);
        const double start = -8.0, stop = 8.0;
        const double st = (stop - start) / (double)(P - 1);
        for (int i = 0; i < P; ++i) x[i] = (double)i * st + start;
        x[P - 1] = stop;
        step = x[1] - x[0];
        for (int i = 0; i < P; ++i)
            psi[i] = (1.0 - x[i] * x[i]) * exp(-(x[i] * x[i]) / 2.0) * 2.0 / (sqrt(3.0) * sqrt(sqrt(M_PI)));
        int_psi.resize(P);
        double acc = 0.0;
        for (int i = 0; i < P; ++i) {
            acc += psi[i];
            int_psi[i] = acc;
        }
        for (int i = 0; i < P; ++i) int_psi[i] *= step;
    }
    void taps(double scale, std::vector<double> &out) const {
        const int cnt = (int)ceil(scale * 16.0 + 1.0);
        std::vector<int> j;
        for (int k = 0; k < cnt; ++k) {
            const int jj = (int)((double)k / (scale * step));
            if (jj < (int)int_psi.size()) j.push_back(jj);
        }
        out.resize(j.size());
        for (size_t i = 0; i < j.size(); ++i) out[i] = int_psi[j[j.size() - 1 - i]];
    }
    // filter column for (scale, t): w[n], n < support;  returns the support length (t + f + 2)
    int column(double scale, int t, std::vector<double> &w) const {
        std::vector<double> tp;
        taps(scale, tp);
        const int T = (int)tp.size();
        if (T < 2) return -1;
        const int f = (T - 2) / 2;
        const int sup = t + f + 2;
        w.assign(sup, 0.0);
        const double sq = -sqrt(scale);
        for (int n = 0; n < sup; ++n) {
            const int i1 = t + f + 1 - n, i0 = t + f - n;
            const double a = (i1 >= 0 && i1 < T) ? tp[i1] : 0.0;
            const double c = (i0 >= 0 && i0 < T) ? tp[i0] : 0.0;
            w[n] = sq * (a - c);
        }
        return sup;
    }
};

// The dense filter bank for the cwt_coefficients specs of a plan: W[c][k], c < C (padded to Cpad rows), k < S4.
struct TsfaCwtBank {
    std::vector<double> W;
    std::vector<int> cols, coeff_idx;
    int S4 = 0, C = 0, Cpad = 0;
    // returns empty string on success
    std::string build(const std::vector<TsfaSpec> &specs) {
        TsfaMexh mexh;
        C = (int)specs.size();
        Cpad = ((C + 15) / 16) * 16;
        std::vector<std::vector<double>> colsW(C);
        int S = 0;
        cols.assign(Cpad, 0);
        coeff_idx.assign(Cpad, 0);
        for (int c = 0; c < C; ++c) {
            const double w = specs[c].p[0];
            const int t = (int)specs[c].p[1];
            if (!(w > 0.0) || t < 0) return "cwt_coefficients: width must be > 0 and coeff >= 0";
            const int sup = mexh.column(w, t, colsW[c]);
            if (sup < 0) return "cwt_coefficients: selected scale too small";
            if (sup > S) S = sup;
            cols[c] = specs[c].col;
            coeff_idx[c] = t;
        }
        S4 = ((S + 3) / 4) * 4;
        W.assign((size_t)Cpad * S4, 0.0);
        for (int c = 0; c < C; ++c)
            for (size_t k = 0; k < colsW[c].size(); ++k) W[(size_t)c * S4 + k] = colsW[c][k];
        return "";
    }
};

// what holds for the plan as a whole; returns "" or the reason
static inline std::string tsfa_validate_plan(const TsfaSpec *specs, int n) {
    int adf_mode = -1;
    for (int i = 0; i < n; ++i)
        if (specs[i].calc == TSFA_C_AUGMENTED_DICKEY_FULLER && (int)specs[i].p[0] != 3) {   // code 3: a NaN column, no fit
            if (adf_mode >= 0 && (int)specs[i].p[1] != adf_mode)
                return "augmented_dickey_fuller: one autolag value per plan (the kernels hold one fit per series)";
            adf_mode = (int)specs[i].p[1];
        }
    return "";
}

// validate one spec; returns "" if it can be evaluated natively, else the reason
// ---- parameter values beyond the tables of the tuned kernels (fam_general.h) ----
#define TSFA_GEN_MAX_M 60      // friedrich / Langevin polynomial degree (x^m of a float64 design; the reference's own fit is noise long before)
#define TSFA_AR_MAX_K 1024     // ar_coefficient order (the normal equations of the double-double pass: (k + 2)^2 entries)
#define TSFA_CWTP_TABLE_N 16   // number_cwt_peaks widths 1 .. n of k_cwtpeaks
// does this column ask the tuned kernel of its calculator for more than its tables hold?
static inline bool tsfa_spec_beyond_tables(const TsfaSpec &s) {
    const double *p = s.p;
    switch (s.calc) {
    case TSFA_C_AGG_AUTOCORRELATION: return p[1] > 60;
    case TSFA_C_PARTIAL_AUTOCORRELATION: return p[0] > 40;
    case TSFA_C_FRIEDRICH_COEFFICIENTS: return p[1] > 3 || p[2] > 64;
    case TSFA_C_MAX_LANGEVIN_FIXED_POINT: return p[0] > 3 || p[1] > 64;
    case TSFA_C_LEMPEL_ZIV_COMPLEXITY: return p[0] > 255;
    case TSFA_C_NUMBER_CWT_PEAKS: return p[0] > TSFA_CWTP_TABLE_N;
    case TSFA_C_QUERY_SIMILARITY_COUNT: return p[3] > 0;   // a query (query=None: the constant NaN column of the BASIC family)
    default: return false;
    }
}
// general[c] = calculator c of this plan goes to k_general -- ALL its columns: the reference's combiners read the largest value
// of the dict (agg_autocorrelation fc.py:428, partial_autocorrelation fc.py:470), and one route per calculator keeps a column's
// value independent of which other columns of OTHER calculators the plan holds.  friedrich_coefficients and
// max_langevin_fixed_point share their fits and move together.
template <class SPEC>
static inline void tsfa_general_calcs(const SPEC *specs, int n, bool *general) {
    for (int c = 0; c < TSFA_N_CALCS; ++c) general[c] = false;
    for (int i = 0; i < n; ++i) {
        TsfaSpec s;
        s.calc = specs[i].calc;
        s.col = 0;
        for (int k = 0; k < 4; ++k) s.p[k] = specs[i].p[k];
        if (s.calc >= 0 && s.calc < TSFA_N_CALCS && tsfa_spec_beyond_tables(s)) general[s.calc] = true;
    }
    if (general[TSFA_C_FRIEDRICH_COEFFICIENTS] || general[TSFA_C_MAX_LANGEVIN_FIXED_POINT])
        general[TSFA_C_FRIEDRICH_COEFFICIENTS] = general[TSFA_C_MAX_LANGEVIN_FIXED_POINT] = true;
}

static inline TsfaGenPlan tsfa_prepare_general(const std::vector<TsfaSpec> &specs) {
    TsfaGenPlan g;
    g.acf_maxlag = -1; g.pacf_maxlag = -1; g.fr_maxr = 0; g.fr_maxm = 0; g.lz = 0; g.cwt_maxw = 0; g.query = 0;
    for (const auto &s : specs) {
        switch (s.calc) {
        case TSFA_C_AGG_AUTOCORRELATION: g.acf_maxlag = std::max(g.acf_maxlag, (int)s.p[1]); break;
        case TSFA_C_PARTIAL_AUTOCORRELATION: g.pacf_maxlag = std::max(g.pacf_maxlag, (int)s.p[0]); break;
        case TSFA_C_FRIEDRICH_COEFFICIENTS: g.fr_maxm = std::max(g.fr_maxm, (int)s.p[1]); g.fr_maxr = std::max(g.fr_maxr, (int)s.p[2]); break;
        case TSFA_C_MAX_LANGEVIN_FIXED_POINT: g.fr_maxm = std::max(g.fr_maxm, (int)s.p[0]); g.fr_maxr = std::max(g.fr_maxr, (int)s.p[1]); break;
        case TSFA_C_LEMPEL_ZIV_COMPLEXITY: g.lz = 1; break;
        case TSFA_C_NUMBER_CWT_PEAKS: g.cwt_maxw = std::max(g.cwt_maxw, (int)s.p[0]); break;
        case TSFA_C_QUERY_SIMILARITY_COUNT: g.query += 1; break;
        default: break;
        }
    }
    return g;
}

static inline std::string tsfa_validate_spec(const TsfaSpec &s) {
    const double *p = s.p;
    auto is_int = [](double v) { return v == floor(v); };
    switch (s.calc) {
    case TSFA_C_NUMBER_PEAKS: if (!(is_int(p[0]) && p[0] >= 1)) return "number_peaks: n must be an integer >= 1"; break;
    case TSFA_C_BINNED_ENTROPY: if (!(is_int(p[0]) && p[0] >= 1 && p[0] <= 1048576)) return "binned_entropy: max_bins must be in [1, 1048576]"; break;   // (beyond 256 bins: counted in rounds of 256)
    case TSFA_C_FOURIER_ENTROPY: if (!(is_int(p[0]) && p[0] >= 1 && p[0] <= 1048576)) return "fourier_entropy: bins must be in [1, 1048576]"; break;   // (beyond 128 bins: rounds of 128)
    case TSFA_C_LEMPEL_ZIV_COMPLEXITY: if (!(is_int(p[0]) && p[0] >= 1 && p[0] <= 1073741824.0)) return "lempel_ziv_complexity: bins must be in [1, 2^30]"; break;   // (beyond 255: fam_general.h)
    case TSFA_C_ENERGY_RATIO_BY_CHUNKS:
        if (!(is_int(p[0]) && is_int(p[1]) && p[0] > 0 && p[1] >= 0 && p[1] < p[0])) return "energy_ratio_by_chunks: need 0 <= segment_focus < num_segments";
        break;
    case TSFA_C_C3: case TSFA_C_TIME_REVERSAL_ASYMMETRY_STATISTIC: case TSFA_C_AUTOCORRELATION:
        if (!(is_int(p[0]) && p[0] >= 0)) return "lag must be a non-negative integer";
        break;
    case TSFA_C_LINEAR_TREND: if (!(p[0] >= 0 && p[0] <= 4)) return "linear_trend: unknown attr"; break;
    case TSFA_C_LINEAR_TREND_TIMEWISE: if (!(p[0] >= 0 && p[0] <= 4)) return "linear_trend_timewise: unknown attr"; break;
    case TSFA_C_AGG_LINEAR_TREND:
        if (!(p[0] >= 0 && p[0] <= 4)) return "agg_linear_trend: unknown attr";
        if (!(is_int(p[1]) && p[1] >= 1)) return "agg_linear_trend: chunk_len must be a positive integer";
        if (!(p[2] >= 0 && p[2] <= 3)) return "agg_linear_trend: f_agg must be max/min/mean/var";
        break;
    case TSFA_C_MEAN_N_ABSOLUTE_MAX: if (!(is_int(p[0]) && p[0] >= 1)) return "mean_n_absolute_max: number_of_maxima must be >= 1"; break;
    case TSFA_C_CHANGE_QUANTILES:
        if (!(p[3] == TSFA_AGG_MEAN || p[3] == TSFA_AGG_VAR)) return "change_quantiles: f_agg must be mean or var";
        if (!(p[0] >= 0 && p[0] <= 1 && p[1] >= 0 && p[1] <= 1)) return "change_quantiles: ql, qh must be in [0, 1]";
        break;
    case TSFA_C_QUANTILE: if (!(p[0] >= 0 && p[0] <= 1)) return "quantile: q must be in [0, 1]"; break;
    case TSFA_C_PERMUTATION_ENTROPY:
        if (!(is_int(p[0]) && p[0] >= 1)) return "permutation_entropy: tau must be >= 1";
        if (!(is_int(p[1]) && p[1] >= 2 && p[1] <= 10)) return "permutation_entropy: dimension must be in [2, 10]";   // (8 .. 10: sort-and-count of the pattern codes, fam_sort.h)
        break;
    case TSFA_C_FRIEDRICH_COEFFICIENTS:
        if (!(is_int(p[1]) && p[1] >= 1 && p[1] <= TSFA_GEN_MAX_M)) return "friedrich_coefficients: m must be in [1, 60]";   // (m > 3, r > 64: fam_general.h; x^61 overflows float64 long before)
        if (!(is_int(p[2]) && p[2] >= 1 && p[2] <= 1048576)) return "friedrich_coefficients: r must be in [1, 1048576]";
        if (!(is_int(p[0]) && p[0] >= 0)) return "friedrich_coefficients: coeff must be >= 0";
        break;
    case TSFA_C_MAX_LANGEVIN_FIXED_POINT:
        if (!(is_int(p[0]) && p[0] >= 1 && p[0] <= TSFA_GEN_MAX_M)) return "max_langevin_fixed_point: m must be in [1, 60]";
        if (!(is_int(p[1]) && p[1] >= 1 && p[1] <= 1048576)) return "max_langevin_fixed_point: r must be in [1, 1048576]";
        break;
    case TSFA_C_FFT_COEFFICIENT:
        if (!(is_int(p[0]) && p[0] >= 0)) return "fft_coefficient: coeff must be >= 0";
        if (!(p[1] >= 0 && p[1] <= 3)) return "fft_coefficient: unknown attr";
        break;
    case TSFA_C_FFT_AGGREGATED: if (!(p[0] >= 0 && p[0] <= 3)) return "fft_aggregated: unknown aggtype"; break;
    case TSFA_C_SPKT_WELCH_DENSITY: if (!(is_int(p[0]) && p[0] >= 0)) return "spkt_welch_density: coeff must be >= 0"; break;
    case TSFA_C_AGG_AUTOCORRELATION:
        if (!(p[0] == TSFA_AGG_MEAN || p[0] == TSFA_AGG_MEDIAN || p[0] == TSFA_AGG_VAR)) return "agg_autocorrelation: f_agg must be mean/median/var";
        if (!(is_int(p[1]) && p[1] >= 1 && p[1] <= 2147483647.0)) return "agg_autocorrelation: maxlag must be a positive integer";   // (beyond 60: fam_general.h)
        break;
    case TSFA_C_PARTIAL_AUTOCORRELATION: if (!(is_int(p[0]) && p[0] >= 0 && p[0] <= 2147483647.0)) return "partial_autocorrelation: lag must be a non-negative integer"; break;   // (beyond 40: fam_general.h)
    case TSFA_C_AR_COEFFICIENT:
        if (!(is_int(p[0]) && p[0] >= 0)) return "ar_coefficient: coeff must be >= 0";
        if (!(is_int(p[1]) && p[1] >= 1 && p[1] <= TSFA_AR_MAX_K)) return "ar_coefficient: k must be in [1, 1024]";   // (beyond 31: every series through the double-double pass, fam_ar_dd.h)
        break;
    case TSFA_C_AUGMENTED_DICKEY_FULLER:
        if (!(is_int(p[0]) && p[0] >= 0 && p[0] <= 3)) return "augmented_dickey_fuller: unknown attr code";   // (3: a name the reference answers with NaN, fc.py:543)
        if (!(is_int(p[1]) && p[1] >= TSFA_AUTOLAG_AIC && p[1] <= TSFA_AUTOLAG_NONE)) return "augmented_dickey_fuller: autolag must be AIC, BIC, t-stat or None";
        break;
    case TSFA_C_APPROXIMATE_ENTROPY:
        if (!(is_int(p[0]) && p[0] >= 1)) return "approximate_entropy: m must be >= 1";
        if (!(p[1] >= 0)) return "approximate_entropy: Parameter r must be positive.";
        break;
    case TSFA_C_NUMBER_CWT_PEAKS: if (!(is_int(p[0]) && p[0] >= 1 && p[0] <= 65536)) return "number_cwt_peaks: n must be in [1, 65536]"; break;   // (beyond 16: fam_general.h)
    case TSFA_C_QUERY_SIMILARITY_COUNT:   // (threshold, normalize, offset of the query in the plan's pool, its length; length 0: query=None)
        if (!(is_int(p[2]) && p[2] >= 0 && is_int(p[3]) && p[3] >= 0 && p[3] <= 16777216.0)) return "query_similarity_count: bad query reference";
        break;
    default: break;
    }
    return "";
}

#endif
