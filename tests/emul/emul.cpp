// TEST INFRASTRUCTURE ONLY -- never loaded by the tsfresh_amd package.
//
// Compiles the per-series kernel sources (tsfresh_amd/csrc/fam_*.h) with g++ as a single-thread (nt = 1) program so
// that the kernel LOGIC can be checked against the oracle on a box without a GPU (this container has none; GPU
// minutes are scarce).  It does not exercise wavefront shuffles, LDS races, barriers or the MFMA fragment layout --
// those are covered by the `-m gpu` tests, which call the real library through the C-ABI.
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/tsfresh_amd.h"
#include "../../tsfresh_amd/csrc/fam_ar.h"
#include "../../tsfresh_amd/csrc/fam_ar_dd.h"
#include "../../tsfresh_amd/csrc/fam_basic.h"
#include "../../tsfresh_amd/csrc/fam_cwt.h"
#include "../../tsfresh_amd/csrc/fam_entropy.h"
#include "../../tsfresh_amd/csrc/fam_perm.h"
#include "../../tsfresh_amd/csrc/fam_seq.h"
#include "../../tsfresh_amd/csrc/fam_sort.h"
#include "../../tsfresh_amd/csrc/fam_general.h"
#include "../../tsfresh_amd/csrc/fam_spectral.h"
#include "../../tsfresh_amd/csrc/tsfa_host_tables.h"
#include "../../tsfresh_amd/csrc/tsfa_layout.h"

// TSFA_EMUL_POISON=<1|2|3>: every scratch buffer of the emulation (the stand-ins for LDS, which the device never
// clears between series) starts out as NaN / a large finite value / a small negative one instead of zeros.  The outputs
// must not depend on it (tests/test_oracle_golden.py): a kernel that reads scratch it has not written reads the PREVIOUS
// series' values on the device.
// non-power-of-two lengths from here on take the chirp-z transform on even series (TSFA_EMUL_BLUESTEIN_MIN: the tests
// lower it to reach every tile / cross-pass shape on short series)
static int emul_bluestein_min() {
    const char *e = getenv("TSFA_EMUL_BLUESTEIN_MIN");
    return e ? std::max(atoi(e), 3) : TSFA_BLUESTEIN_MIN_EVEN;   // (the smaller of the two crossovers: scratch for both)
}
static int emul_bluestein_pack() {
    const char *e = getenv("TSFA_EMUL_BLUESTEIN_MIN");
    const int v = e ? std::max(atoi(e), 3) : 0;
    return e ? TSFA_BLUESTEIN_PACK(v, v) : TSFA_BLUESTEIN_PACK(TSFA_BLUESTEIN_MIN, TSFA_BLUESTEIN_MIN_EVEN);
}

static int emul_poison_mode() {
    const char *e = getenv("TSFA_EMUL_POISON");
    return e ? atoi(e) : 0;
}
static void poison(std::vector<double> &v, size_t from = 0) {
    const int m = emul_poison_mode();
    if (!m) return;
    const double val = (m == 1) ? TSFA_NAN : (m == 2) ? 1.0e30 : -0.75;
    for (size_t i = from; i < v.size(); ++i) v[i] = val;
}
template <class I>
static void poison_int(std::vector<I> &v) {
    const int m = emul_poison_mode();
    if (!m) return;
    const unsigned long long pat = (m == 1) ? 0x7ff800007ff80000ull : (m == 2) ? 0x3fe000003fe00000ull : 0xfffffff9fffffff9ull;
    for (size_t i = 0; i < v.size(); ++i) v[i] = (I)pat;
}

extern "C" int tsfa_emul_calc_id(const char *name) {
    for (int i = 0; i < TSFA_N_CALCS; ++i)
        if (strcmp(tsfa_calc_table[i].name, name) == 0) return i;
    return -1;
}

extern "C" int tsfa_emul_extract_timed(const tsfa_feature_spec *specs, int n_specs, const double *values,
                                       const double *times, const int64_t *offsets, int64_t n_series, double *out,
                                       int64_t ld, char *err, int errlen);

// the float64 pool of array-valued parameters (tsfa_plan_create_with_data): set before the extract that uses it
static std::vector<double> g_pool;
extern "C" void tsfa_emul_set_pool(const double *data, int64_t n) { g_pool.assign(data, data + (n > 0 ? n : 0)); }

extern "C" int tsfa_emul_extract(const tsfa_feature_spec *specs, int n_specs, const double *values,
                                 const int64_t *offsets, int64_t n_series, double *out, int64_t ld, char *err,
                                 int errlen) {
    return tsfa_emul_extract_timed(specs, n_specs, values, nullptr, offsets, n_series, out, ld, err, errlen);
}

extern "C" int tsfa_emul_extract_timed(const tsfa_feature_spec *specs, int n_specs, const double *values,
                                       const double *times, const int64_t *offsets, int64_t n_series, double *out,
                                       int64_t ld, char *err, int errlen) {
    std::vector<TsfaSpec> fam[TSFA_N_FAMILIES], cwt_coef, gen;
    bool general[TSFA_N_CALCS];
    tsfa_general_calcs(specs, n_specs, general);
    for (int i = 0; i < n_specs; ++i) {
        TsfaSpec s;
        s.calc = specs[i].calc;
        s.col = i;
        for (int k = 0; k < 4; ++k) s.p[k] = specs[i].p[k];
        if (s.calc < 0 || s.calc >= TSFA_N_CALCS) {
            snprintf(err, errlen, "spec %d: unknown calculator", i);
            return TSFA_ERR_UNSUPPORTED;
        }
        const std::string why = tsfa_validate_spec(s);
        if (!why.empty()) {
            snprintf(err, errlen, "spec %d (%s): %s", i, tsfa_calc_table[s.calc].name, why.c_str());
            return TSFA_ERR_UNSUPPORTED;
        }
        if (s.calc == TSFA_C_CWT_COEFFICIENTS) cwt_coef.push_back(s);
        else if (general[s.calc]) gen.push_back(s);
        else fam[tsfa_calc_table[s.calc].family].push_back(s);
    }
    {
        std::vector<TsfaSpec> all;
        for (int f = 0; f < TSFA_N_FAMILIES; ++f) all.insert(all.end(), fam[f].begin(), fam[f].end());
        const std::string why = tsfa_validate_plan(all.data(), (int)all.size());
        if (!why.empty()) {
            snprintf(err, errlen, "%s", why.c_str());
            return TSFA_ERR_UNSUPPORTED;
        }
    }
    TsfaFamHints hints[TSFA_N_FAMILIES];
    for (int f = 0; f < TSFA_N_FAMILIES; ++f) tsfa_prepare_family(f, fam[f], hints[f]);
    TsfaCwtBank bank;
    if (!cwt_coef.empty()) {
        const std::string why = bank.build(cwt_coef);
        if (!why.empty()) {
            snprintf(err, errlen, "%s", why.c_str());
            return TSFA_ERR_UNSUPPORTED;
        }
    }
    std::vector<double> dectab, twc, tws;
    tsfa_build_dectab(dectab);
    tsfa_build_twiddles(twc, tws);
    std::vector<double> consts;
    tsfa_build_consts(consts);

    // the device sizes a launch's LDS layout, table plans and matrix dimensions for the LONGEST series of the launch
    // (a length class); the emulation does the same with the longest series of the call
    int call_maxn = 1;
    for (int64_t s = 0; s < n_series; ++s) call_maxn = std::max(call_maxn, (int)(offsets[s + 1] - offsets[s]));
    for (int64_t s = 0; s < n_series; ++s) {
        const int n = (int)(offsets[s + 1] - offsets[s]);
        const double *x = values + offsets[s];
        double *row = out + s * ld;
        for (int c = 0; c < n_specs; ++c) row[c] = TSFA_NAN;
        if (n < 1) {
            snprintf(err, errlen, "empty series");
            return TSFA_ERR_INVALID;
        }
        const int maxn = (s % 3 == 1) ? n : call_maxn;   // every third series: a launch of its own
        std::vector<double> red(TSFA_RED_DOUBLES);
        poison(red);
        NpScratch nps;
        if (emul_poison_mode()) memset(&nps, 0x7f, sizeof nps);
        Blk b{0, 1, red.data(), &nps};
        std::vector<double> xs(x, x + n);
        xs.resize(n + 8, 0.0);
        poison(xs, (size_t)n);
        if (!fam[TSFA_FAM_BASIC].empty()) {
            std::vector<double> w(maxn + 8), cum(maxn + 8), altc(8 * 16), ctx(32);
            std::vector<int> iw(512);
            poison(w); poison(cum); poison(altc); poison(ctx); poison_int(iw);
            fam_basic_series<1>(b, xs.data(), n, fam[TSFA_FAM_BASIC].data(), (int)fam[TSFA_FAM_BASIC].size(), row, w.data(),
                                (s % 2) ? w.data() : cum.data(), altc.data(), iw.data(), dectab.data(),
                                hints[TSFA_FAM_BASIC].a, 0, hints[TSFA_FAM_BASIC].alt, nullptr, nullptr,
                                (s % 2) ? -1 : hints[TSFA_FAM_BASIC].c, ctx.data(),
                                (s % 4 == 1) ? 0 : hints[TSFA_FAM_BASIC].d);
        }
        if (!fam[TSFA_FAM_TREND].empty()) {
            // (BasicLds: with TsfaAltPlan::small_w the work array is nt + 2 * 16 + 8 doubles whatever the length)
            std::vector<double> w(maxn + 48, TSFA_NAN), cum(maxn + 48, TSFA_NAN), altc(8 * 16), ctx(32);
            std::vector<int> iw(512, 0x3fe00000);   // poisoned with finite values: LDS keeps the previous series' contents
            poison(w); poison(cum); poison(altc); poison(ctx); poison_int(iw);
            fam_basic_series<2>(b, xs.data(), n, fam[TSFA_FAM_TREND].data(), (int)fam[TSFA_FAM_TREND].size(), row, w.data(),
                                (s % 2) ? w.data() : cum.data(), altc.data(), iw.data(), dectab.data(), 0,
                                hints[TSFA_FAM_TREND].b, hints[TSFA_FAM_TREND].alt, nullptr,
                                times ? times + offsets[s] : nullptr, (s % 2) ? -1 : hints[TSFA_FAM_TREND].c, ctx.data());
        }
        if (!fam[TSFA_FAM_SORT].empty()) {
            std::vector<double> srt(tsfa_pow2_ceil(maxn) + 8), w(1280), cq(5 * TSFA_CQ_MAX), sctx(8);
            poison(srt); poison(w); poison(cq); poison(sctx);
            // every third series: the sorted copy as a gather through the sample order another family left behind
            // (k_entropy_bits -> k_sort on the device)
            std::vector<unsigned short> order;
            if (s % 3 == 0 && n >= 3 && n <= 65535) {
                order.resize(n);
                for (int i = 0; i < n; ++i) order[i] = (unsigned short)i;
                std::stable_sort(order.begin(), order.end(), [&](unsigned short a, unsigned short c) { return xs[a] < xs[c]; });
            }
            fam_sort_series(b, xs.data(), n, fam[TSFA_FAM_SORT].data(), (int)fam[TSFA_FAM_SORT].size(), row, srt.data(),
                            w.data(), (int *)w.data(), hints[TSFA_FAM_SORT].cq, cq.data(), nullptr,
                            (s % 2) ? -1 : hints[TSFA_FAM_SORT].c, sctx.data(),
                            (s % 4 >= 2) ? hints[TSFA_FAM_SORT].a : 1280,  // plan-sized scratch: multi-pass pattern histogram
                            FrDefer{nullptr, nullptr, TSFA_PF_HDR + 2 * TSFA_FRIEDRICH_MAX_R, 0, 0},
                            order.empty() ? nullptr : order.data(), nullptr,
                            (s % 2 == 0 && n <= 40960) ? hints[TSFA_FAM_SORT].d : 0);   // every other series: permutation_entropy left to k_perm's code ...
            if (s % 2 == 0 && n <= 40960 && hints[TSFA_FAM_SORT].d != 0) {   // (the device admits k_perm only where the series fits LDS)                    // ... all dimensions from one sweep (fam_perm.h)
                std::vector<int> hist(TSFA_PE_HIST_WORDS + 4);
                std::vector<double> ltab(TSFA_PE_LOGS + TSFA_PE_MAXD + 1);
                for (auto &v : hist) v = 0x5a5a5a5a;
                poison(ltab);
                fam_perm_series(b, xs.data(), n, fam[TSFA_FAM_SORT].data(), (int)fam[TSFA_FAM_SORT].size(), row,
                                hints[TSFA_FAM_SORT].d >> 8, hist.data(), ltab.data());
            }
        }
        if (!fam[TSFA_FAM_SPECTRAL].empty()) {
            const int nx = (maxn / 2 + 2 > 260) ? maxn / 2 + 2 : 260;
            const int nd = (maxn > 256) ? maxn : 256;
            std::vector<double> Xr(nx), Xi(nx), tc(nd), ts(nd), win(256), pxx(132), chirp_tab(1024);
            poison(chirp_tab);
            std::vector<int> iw(128);
            poison(Xr); poison(Xi); poison(tc); poison(ts); poison(win); poison(pxx); poison_int(iw);
            // even series take the Bluestein route where it applies (HBM scratch on the device), odd ones the Goertzel sweep
            std::vector<double> gsv((s % 2 == 0 && n >= emul_bluestein_min() && bluestein_m(n) <= TSFA_BLUESTEIN_MAXM)
                                        ? 4 * (size_t)bluestein_m(n) : 0);
            poison(gsv);
            fam_spectral_series(b, xs.data(), n, fam[TSFA_FAM_SPECTRAL].data(), (int)fam[TSFA_FAM_SPECTRAL].size(), row,
                                Xr.data(), Xi.data(), tc.data(), ts.data(), win.data(), pxx.data(), iw.data(),
                                twc.data(), tws.data(), hints[TSFA_FAM_SPECTRAL].a, hints[TSFA_FAM_SPECTRAL].b,
                                gsv.empty() ? nullptr : gsv.data(), (s % 2 == 0) ? consts.data() + TSFA_CONSTS_HANN : nullptr,
                                chirp_tab.data(), emul_bluestein_pack());
        }
        if (!fam[TSFA_FAM_AR].empty()) {
            int P = 8, Pdd = 8;
            for (const auto &sp : fam[TSFA_FAM_AR]) {
                if (sp.calc == TSFA_C_AUGMENTED_DICKEY_FULLER) P = std::max(P, adf_maxlag_for(maxn) + 3);
                else if (sp.calc == TSFA_C_AR_COEFFICIENT) {
                    if ((int)sp.p[1] <= TSFA_AR_TABLE_K) P = std::max(P, (int)sp.p[1] + 2);
                    Pdd = std::max(Pdd, std::min((int)sp.p[1], std::max(1, (maxn - 2) / 2)) + 2);
                }
            }
            Pdd = std::max(Pdd, P);
            std::vector<double> xc(maxn + TSFA_AR_PADL + TSFA_AR_PADR, TSFA_NAN), aw(ArLds::scratch_doubles(P));
            poison(xc); poison(aw);
            const double *xp = xs.data();
            const int flags = fam_ar_series<double>(b, [=](int i) { return xp[i]; }, n, fam[TSFA_FAM_AR].data(),
                          (int)fam[TSFA_FAM_AR].size(), row, (void *)xc.data(), aw.data(), P, hints[TSFA_FAM_AR].a,
                          hints[TSFA_FAM_AR].b, hints[TSFA_FAM_AR].c, (s % 2) ? -1 : hints[TSFA_FAM_AR].d);
            if (flags) {  // the second pass of the family (k_ar_degenerate)
                std::vector<double> sc(ArDdLds::scratch_doubles(Pdd), TSFA_NAN);  // poisoned: LDS is not zero-initialised on the device
                fam_ar_degenerate_series(b, [=](int i) { return xp[i]; }, n, fam[TSFA_FAM_AR].data(),
                                         (int)fam[TSFA_FAM_AR].size(), row, sc.data(), Pdd, flags, (hints[TSFA_FAM_AR].c >> 1) & 3);
            }
        }
        if (!fam[TSFA_FAM_ENTROPY].empty()) {
            std::vector<double> thr(56), xe(xs.begin(), xs.begin() + n);
            xe.resize(n + 8, 0.0);
            std::vector<unsigned short> perm(tsfa_pow2_ceil(maxn) + 96);
            std::vector<unsigned int> cnt((size_t)(maxn + 16) * 4), refs(perm.size());
            poison(thr); poison(xe, (size_t)n); poison_int(perm); poison_int(cnt); poison_int(refs);
            // odd series exercise the ordered-pair sweep (no LDS counters), every fourth the grouped symmetric sweep
            // instead of the staged one
            bool all_m2 = true;
            for (const auto &sp : fam[TSFA_FAM_ENTROPY])
                if (sp.calc == TSFA_C_APPROXIMATE_ENTROPY && (int)sp.p[0] != 2) all_m2 = false;
            const char *force = getenv("TSFA_EMUL_ENTROPY");  // "bits" / "pairs": one sweep for every series
            const bool bits = all_m2 && n <= TSFA_ENTB_MAXN_LONG && (force ? !strcmp(force, "bits") : (s % 4 == 2 || n > TSFA_ENTB_MAXN));
            if (bits && n <= TSFA_ENTB_MAXN_WIDE) {  // the bit-matrix sweep (k_entropy_bits), 48-byte table entries
                std::vector<unsigned int> work(entb_work_words(n) + 64);
                poison_int(work);
                fam_entropy_series_bits<false>(b, xe.data(), n, fam[TSFA_FAM_ENTROPY].data(), (int)fam[TSFA_FAM_ENTROPY].size(),
                                               row, thr.data(), perm.data(), work.data());
            } else if (bits) {  // ... with 16-byte table entries and tolerance rounds (series of 1025 .. 4096 samples)
                // the work region as the DEVICE sizes it: for the longest series of the launch (here: of the call), which may
                // hold fewer tolerances per round than this series' registers would; canary words behind it
                int lmax = n;
                for (int64_t q = 0; q < n_series; ++q) {
                    const int nq_ = (int)(offsets[q + 1] - offsets[q]);
                    if (nq_ <= TSFA_ENTB_MAXN_LONG && nq_ > lmax) lmax = nq_;
                }
                const int kcap = entb_kround(lmax, TSFA_ENTB_MAXK, TSFA_ENTB_MAXWAVES);
                const size_t words = entb_work_words(lmax, TSFA_ENTB_QW_LONG + 1, kcap);
                std::vector<unsigned int> work(words + 64, 0xFFFFFFFFu);
                for (size_t q = words; q < words + 64; ++q) work[q] = 0xC0FFEE00u + (unsigned int)(q - words);
                fam_entropy_series_bits<false, TSFA_ENTB_QW_LONG>(b, xe.data(), n, fam[TSFA_FAM_ENTROPY].data(),
                                                                 (int)fam[TSFA_FAM_ENTROPY].size(), row, thr.data(), perm.data(),
                                                                 work.data(), nullptr, kcap);
                for (size_t q = words; q < words + 64; ++q)
                    if (work[q] != 0xC0FFEE00u + (unsigned int)(q - words)) {
                        snprintf(err, errlen, "entropy work region overrun (series of %d samples in a launch of up to %d)", n, lmax);
                        return TSFA_ERR_INVALID;
                    }
            } else
            fam_entropy_series<double>(b, xe.data(), n, fam[TSFA_FAM_ENTROPY].data(), (int)fam[TSFA_FAM_ENTROPY].size(),
                                       row, thr.data(), perm.data(), refs.data(), (s % 2) ? nullptr : cnt.data(),
                                       (s % 4 == 0) ? 1 : 0);
        }
        if (!fam[TSFA_FAM_SEQ].empty()) {
            const int group = 2;  // exercise the multi-launch path
            const int nsq = (int)fam[TSFA_FAM_SEQ].size();
            const double *xp = xs.data();
            for (int s0 = 0; s0 < nsq; s0 += group) {
                TsfaSeqGroup g;
                const int grows = (s % 2 == 1) ? 1 : 0;  // exercise both homes of the symbol rows (LDS: packed; HBM: a byte per symbol)
                lz_build_group(fam[TSFA_FAM_SEQ].data() + s0, std::min(group, nsq - s0), maxn, &g, grows);
                std::vector<uint32_t> seqw(((size_t)g.stride + 16) / 4 + 8), tab((size_t)g.ttotal + 4);
                std::vector<double> edges(g.etotal + 4);
                poison_int(seqw); poison_int(tab); poison(edges);
                unsigned char *seqp = (unsigned char *)seqw.data();
                seqp += (16 - ((uintptr_t)seqp & 15)) & 15;
                if (grows)
                    fam_seq_series<true>(b, [=](int i) { return xp[i]; }, n, g, row, seqp, tab.data(), edges.data());
                else
                    fam_seq_series<false>(b, [=](int i) { return xp[i]; }, n, g, row, seqp, tab.data(), edges.data());
            }
        }
        if (!fam[TSFA_FAM_CWT].empty()) {
            const int with_rowv = (s % 2 == 0) ? 1 : 0;  // exercise both variants
            std::vector<unsigned char> lds(CwtPeaksLayout().carve(nullptr, maxn, with_rowv) + 64);
            poison_int(lds);
            CwtPeaksLayout L;
            unsigned char *basep = lds.data();
            basep += (16 - ((uintptr_t)basep & 15)) & 15;
            L.carve(basep, maxn, with_rowv);
            L.p.rk = (s % 2 == 0) ? consts.data() + TSFA_CONSTS_RICKER : nullptr;   // every other series from the plan's table
            const double *xp = xs.data();
            fam_cwtpeaks_series<double>(b, [=](int i) { return xp[i]; }, n, fam[TSFA_FAM_CWT].data(), (int)fam[TSFA_FAM_CWT].size(),
                                row, L.p);
        }
        if (!gen.empty()) {   // k_general: a slot sized for the longest series of the call, poisoned
            const TsfaGenPlan gp = tsfa_prepare_general(gen);
            GenSlot S;
            std::vector<double> slot(S.carve(nullptr, call_maxn, gp) + 8);
            poison(slot);
            S.carve(slot.data(), call_maxn, gp);
            const double *xp = xs.data();
            fam_general_series(b, [=](int i) { return xp[i]; }, n, gen.data(), (int)gen.size(), row, S, gp, g_pool.data());
        }
        for (int c = 0; c < bank.C; ++c) {  // plain-loop stand-in for k_cwt_gemm
            double acc = 0.0;
            const int kmax = (n < bank.S4) ? n : bank.S4;
            for (int k = 0; k < kmax; ++k) acc += x[k] * bank.W[(size_t)c * bank.S4 + k];
            row[bank.cols[c]] = (bank.coeff_idx[c] < n) ? acc : TSFA_NAN;
        }
    }
    return TSFA_OK;
}
