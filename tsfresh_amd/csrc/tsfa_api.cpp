// C-ABI of libtsfresh_amd.so (see include/tsfresh_amd.h).  Host code: plan compilation, HBM staging, launches.
// There is deliberately no CPU compute path here: without a HIP device every entry point that would compute
// returns TSFA_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/tsfresh_amd.h"
#include "tsfa_host_tables.h"
#include "tsfa_launch.h"
#include "fam_seq.h"
#include "tsfa_layout.h"

static thread_local std::string g_last_error;

static int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

// for the other translation units of the library (tsfa_relevance.hip)
int tsfa_fail(int code, const char *msg) { return fail(code, std::string(msg ? msg : "")); }
#define HIP_TRY(expr)                                                                                    \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return fail(TSFA_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 4096;
        if (hipMalloc(&p, want) != hipSuccess) {
            p = nullptr;
            return -1;
        }
        cap = want;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

#define TSFA_MAX_AUX 3
#define TSFA_DEFAULT_STREAMS 1

struct Timing {
    std::string name;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms = 0.f;
};

struct tsfa_plan {
    int device = 0;
    hipStream_t stream = nullptr;
    int n_cols = 0;
    std::vector<TsfaSpec> fam_specs[TSFA_N_FAMILIES];  // CWT slot: number_cwt_peaks specs only
    TsfaSpec *d_specs[TSFA_N_FAMILIES] = {nullptr};
    TsfaFamHints hints[TSFA_N_FAMILIES];
    TsfaCwtBank bank;  // cwt_coefficients
    double *d_W = nullptr;
    int *d_cols = nullptr, *d_coeff = nullptr;
    double *d_dectab = nullptr, *d_twc = nullptr, *d_tws = nullptr;
    long long *d_stats = nullptr;
    DevBuf values, offsets, out, gscratch, times, deg_list;
    int *d_deg_count = nullptr;
    bool needs_times = false;  // the plan holds linear_trend_timewise columns
    // side streams: the family kernels are independent (each writes its own columns), so they may overlap
    int n_streams = 1;
    hipStream_t aux[TSFA_MAX_AUX] = {nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[TSFA_MAX_AUX] = {nullptr};
    bool profiling = false;
    std::vector<Timing> timings;
    long long hint_min_len = 0, hint_max_len = 0;  // tsfa_plan_set_length_hint: skip the length scan (and its host sync)
};

static const char *fam_names[TSFA_N_FAMILIES] = {"k_basic", "k_sort", "k_spectral", "k_ar", "k_entropy", "k_cwtpeaks", "k_seq", "k_trend"};

template <class T>
static int upload(const std::vector<T> &h, T **d) {
    *d = nullptr;
    if (h.empty()) return 0;
    if (hipMalloc((void **)d, h.size() * sizeof(T)) != hipSuccess) return -1;
    if (hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return -1;
    return 0;
}

static int record(tsfa_plan *plan, hipStream_t st, size_t slot, const char *name, bool begin) {
    if (!plan->profiling) return 0;
    if (plan->timings.size() <= slot) plan->timings.resize(slot + 1);
    Timing &t = plan->timings[slot];
    if (!t.e0) {
        if (hipEventCreate(&t.e0) != hipSuccess || hipEventCreate(&t.e1) != hipSuccess) return -1;
    }
    t.name = name;
    return hipEventRecord(begin ? t.e0 : t.e1, st) == hipSuccess ? 0 : -1;
}

extern "C" {

int tsfa_version(void) { return TSFA_VERSION; }

int tsfa_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *tsfa_last_error(void) { return g_last_error.c_str(); }

int tsfa_calc_id(const char *name) {
    if (!name) return -1;
    for (int i = 0; i < TSFA_N_CALCS; ++i)
        if (strcmp(tsfa_calc_table[i].name, name) == 0) return i;
    return -1;
}
const char *tsfa_calc_name(int calc) { return (calc >= 0 && calc < TSFA_N_CALCS) ? tsfa_calc_table[calc].name : nullptr; }
int tsfa_calc_count(void) { return TSFA_N_CALCS; }

void tsfa_plan_destroy(tsfa_plan *plan) {
    if (!plan) return;
    (void)hipSetDevice(plan->device);
    for (int f = 0; f < TSFA_N_FAMILIES; ++f)
        if (plan->d_specs[f]) (void)hipFree(plan->d_specs[f]);
    if (plan->d_W) (void)hipFree(plan->d_W);
    if (plan->d_cols) (void)hipFree(plan->d_cols);
    if (plan->d_coeff) (void)hipFree(plan->d_coeff);
    if (plan->d_dectab) (void)hipFree(plan->d_dectab);
    if (plan->d_twc) (void)hipFree(plan->d_twc);
    if (plan->d_tws) (void)hipFree(plan->d_tws);
    if (plan->d_stats) (void)hipFree(plan->d_stats);
    if (plan->d_deg_count) (void)hipFree(plan->d_deg_count);
    plan->deg_list.release();
    plan->times.release();
    plan->values.release();
    plan->offsets.release();
    plan->out.release();
    plan->gscratch.release();
    for (auto &t : plan->timings) {
        if (t.e0) (void)hipEventDestroy(t.e0);
        if (t.e1) (void)hipEventDestroy(t.e1);
    }
    for (int i = 0; i < TSFA_MAX_AUX; ++i) {
        if (plan->aux[i]) (void)hipStreamDestroy(plan->aux[i]);
        if (plan->ev_join[i]) (void)hipEventDestroy(plan->ev_join[i]);
    }
    if (plan->ev_fork) (void)hipEventDestroy(plan->ev_fork);
    if (plan->stream) (void)hipStreamDestroy(plan->stream);
    delete plan;
}

int tsfa_plan_create(const tsfa_feature_spec *specs, int32_t n_specs, int32_t device, tsfa_plan **out_plan) {
    if (!out_plan) return fail(TSFA_ERR_INVALID, "out_plan is NULL");
    *out_plan = nullptr;
    if (n_specs < 0 || (n_specs > 0 && !specs)) return fail(TSFA_ERR_INVALID, "bad specs");
    const int ndev = tsfa_device_count();
    if (ndev <= 0) return fail(TSFA_ERR_NO_DEVICE, "no HIP device visible: tsfresh_amd has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(TSFA_ERR_INVALID, "device ordinal out of range");

    tsfa_plan *plan = new tsfa_plan();
    plan->device = device;
    plan->n_cols = n_specs;
    std::vector<TsfaSpec> cwt_coef;
    for (int i = 0; i < n_specs; ++i) {
        TsfaSpec s;
        s.calc = specs[i].calc;
        s.col = i;
        for (int k = 0; k < 4; ++k) s.p[k] = specs[i].p[k];
        if (s.calc < 0 || s.calc >= TSFA_N_CALCS) {
            delete plan;
            return fail(TSFA_ERR_UNSUPPORTED, "spec " + std::to_string(i) + ": unknown calculator id " + std::to_string(s.calc));
        }
        const std::string why = tsfa_validate_spec(s);
        if (!why.empty()) {
            delete plan;
            return fail(TSFA_ERR_UNSUPPORTED, std::string("spec ") + std::to_string(i) + " (" + tsfa_calc_table[s.calc].name + "): " + why);
        }
        if (s.calc == TSFA_C_LINEAR_TREND_TIMEWISE) plan->needs_times = true;
        if (s.calc == TSFA_C_CWT_COEFFICIENTS) cwt_coef.push_back(s);
        else plan->fam_specs[tsfa_calc_table[s.calc].family].push_back(s);
    }
    if (cwt_coef.size() > 128) {
        delete plan;
        return fail(TSFA_ERR_UNSUPPORTED, "more than 128 cwt_coefficients columns in one plan");
    }
    if (!cwt_coef.empty()) {
        const std::string why = plan->bank.build(cwt_coef);
        if (!why.empty()) {
            delete plan;
            return fail(TSFA_ERR_UNSUPPORTED, why);
        }
    }
    if (hipSetDevice(device) != hipSuccess) {
        delete plan;
        return fail(TSFA_ERR_HIP, "hipSetDevice failed");
    }
    for (int f = 0; f < TSFA_N_FAMILIES; ++f) tsfa_prepare_family(f, plan->fam_specs[f], plan->hints[f]);
    bool ok = hipStreamCreateWithFlags(&plan->stream, hipStreamNonBlocking) == hipSuccess;
    for (int f = 0; ok && f < TSFA_N_FAMILIES; ++f) ok = upload(plan->fam_specs[f], &plan->d_specs[f]) == 0;
    if (ok && !cwt_coef.empty()) {
        ok = upload(plan->bank.W, &plan->d_W) == 0 && upload(plan->bank.cols, &plan->d_cols) == 0 &&
             upload(plan->bank.coeff_idx, &plan->d_coeff) == 0;
    }
    if (ok) {
        std::vector<double> dt, twc, tws;
        tsfa_build_dectab(dt);
        tsfa_build_twiddles(twc, tws);
        ok = upload(dt, &plan->d_dectab) == 0 && upload(twc, &plan->d_twc) == 0 && upload(tws, &plan->d_tws) == 0;
    }
    if (ok) ok = hipMalloc((void **)&plan->d_stats, 4 * sizeof(long long)) == hipSuccess;
    if (ok) ok = hipMalloc((void **)&plan->d_deg_count, sizeof(int)) == hipSuccess;
    {
        const char *e = getenv("TSFA_STREAMS");
        plan->n_streams = e ? std::min(std::max(atoi(e), 1), TSFA_MAX_AUX + 1) : TSFA_DEFAULT_STREAMS;
        for (int i = 0; ok && i + 1 < plan->n_streams; ++i)
            ok = hipStreamCreateWithFlags(&plan->aux[i], hipStreamNonBlocking) == hipSuccess &&
                 hipEventCreateWithFlags(&plan->ev_join[i], hipEventDisableTiming) == hipSuccess;
        if (ok && plan->n_streams > 1) ok = hipEventCreateWithFlags(&plan->ev_fork, hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) {
        tsfa_plan_destroy(plan);
        return fail(TSFA_ERR_HIP, "device allocation/upload failed while creating the plan");
    }
    *out_plan = plan;
    return TSFA_OK;
}

int32_t tsfa_plan_n_cols(const tsfa_plan *plan) { return plan ? plan->n_cols : -1; }

int tsfa_plan_set_length_hint(tsfa_plan *plan, int64_t min_len, int64_t max_len) {
    if (!plan) return fail(TSFA_ERR_INVALID, "null plan");
    if (max_len == 0 && min_len == 0) { plan->hint_min_len = plan->hint_max_len = 0; return TSFA_OK; }
    if (min_len < 1 || max_len < min_len || max_len > 65535) return fail(TSFA_ERR_INVALID, "length hint must satisfy 1 <= min <= max <= 65535");
    plan->hint_min_len = min_len;
    plan->hint_max_len = max_len;
    return TSFA_OK;
}

int tsfa_plan_set_profiling(tsfa_plan *plan, int32_t enable) {
    if (!plan) return fail(TSFA_ERR_INVALID, "plan is NULL");
    plan->profiling = enable != 0;
    return TSFA_OK;
}

int32_t tsfa_plan_last_timings(const tsfa_plan *plan, const char **names, float *ms, int32_t cap) {
    if (!plan) return 0;
    int32_t n = 0;
    for (const auto &t : plan->timings) {
        if (n >= cap) break;
        if (names) names[n] = t.name.c_str();
        if (ms) ms[n] = t.ms;
        ++n;
    }
    return n;
}

int tsfa_extract(tsfa_plan *plan, const void *values, int32_t dtype, const int64_t *offsets, int64_t n_series,
                 double *out, int64_t ld_out, int32_t space, void *stream) {
    return tsfa_extract_timed(plan, values, dtype, nullptr, offsets, n_series, out, ld_out, space, stream);
}

int tsfa_extract_timed(tsfa_plan *plan, const void *values, int32_t dtype, const double *times, const int64_t *offsets,
                       int64_t n_series, double *out, int64_t ld_out, int32_t space, void *stream) {
    // a ragged batch is the special case ends = starts + 1 of the window form
    return tsfa_extract_windows(plan, values, dtype, times, offsets, offsets ? offsets + 1 : nullptr, n_series, out,
                                ld_out, space, stream);
}

int tsfa_extract_windows(tsfa_plan *plan, const void *values, int32_t dtype, const double *times, const int64_t *starts,
                         const int64_t *ends, int64_t n_series, double *out, int64_t ld_out, int32_t space,
                         void *stream) {
    if (!plan) return fail(TSFA_ERR_INVALID, "plan is NULL");
    if (plan->needs_times && !times)
        return fail(TSFA_ERR_INVALID, "the plan holds linear_trend_timewise columns: call tsfa_extract_timed with the "
                                      "per-sample times (the reference skips the calculator without a DatetimeIndex)");
    if (dtype != TSFA_F32 && dtype != TSFA_F64) return fail(TSFA_ERR_INVALID, "dtype must be TSFA_F32 or TSFA_F64");
    if (space != TSFA_HOST && space != TSFA_DEVICE) return fail(TSFA_ERR_INVALID, "space must be TSFA_HOST or TSFA_DEVICE");
    if (n_series < 0) return fail(TSFA_ERR_INVALID, "n_series < 0");
    if (n_series == 0 || plan->n_cols == 0) return TSFA_OK;
    if (!values || !starts || !ends || !out) return fail(TSFA_ERR_INVALID, "NULL buffer");
    if (ld_out < plan->n_cols) return fail(TSFA_ERR_INVALID, "ld_out < n_cols");
    if (n_series > 2147483647LL) return fail(TSFA_ERR_INVALID, "n_series exceeds the grid limit (2^31 - 1)");
    HIP_TRY(hipSetDevice(plan->device));
    hipStream_t st = (space == TSFA_DEVICE && stream) ? (hipStream_t)stream : plan->stream;
    const size_t esz = (dtype == TSFA_F32) ? 4 : 8;

    const void *d_values = values;
    const double *d_times = plan->needs_times ? times : nullptr;
    const int64_t *d_starts = starts, *d_ends = ends;
    double *d_out = out;
    int64_t ld = ld_out;
    if (space == TSFA_HOST) {
        // stage the span of `values` the windows touch; window bounds become relative to its first sample
        const bool ragged = (ends == starts + 1);
        int64_t base = starts[0], top = ends[0];
        for (int64_t i = 0; i < n_series; ++i) {
            if (ends[i] < starts[i]) return fail(TSFA_ERR_INVALID, "a series ends before it starts (offsets must be non-decreasing)");
            base = std::min(base, starts[i]);
            top = std::max(top, ends[i]);
        }
        const int64_t total = top - base;
        std::vector<int64_t> rel;
        if (ragged) {
            rel.resize((size_t)n_series + 1);
            for (int64_t i = 0; i <= n_series; ++i) rel[(size_t)i] = starts[i] - base;
        } else {
            rel.resize(2 * (size_t)n_series);
            for (int64_t i = 0; i < n_series; ++i) {
                rel[(size_t)i] = starts[i] - base;
                rel[(size_t)(n_series + i)] = ends[i] - base;
            }
        }
        if (plan->values.ensure((size_t)total * esz + 16) || plan->offsets.ensure(rel.size() * sizeof(int64_t)) ||
            plan->out.ensure((size_t)n_series * plan->n_cols * sizeof(double)))
            return fail(TSFA_ERR_HIP, "hipMalloc failed for the staging buffers");
        HIP_TRY(hipMemcpyAsync(plan->values.p, (const char *)values + (size_t)base * esz, (size_t)total * esz,
                               hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(plan->offsets.p, rel.data(), rel.size() * sizeof(int64_t), hipMemcpyHostToDevice, st));
        if (d_times) {
            if (plan->times.ensure((size_t)total * sizeof(double) + 16)) return fail(TSFA_ERR_HIP, "hipMalloc failed for the times buffer");
            HIP_TRY(hipMemcpyAsync(plan->times.p, times + base, (size_t)total * sizeof(double), hipMemcpyHostToDevice, st));
            d_times = (const double *)plan->times.p;
        }
        HIP_TRY(hipStreamSynchronize(st));  // rel goes out of scope below
        d_values = plan->values.p;
        d_starts = (const int64_t *)plan->offsets.p;
        d_ends = d_starts + (ragged ? 1 : n_series);
        d_out = (double *)plan->out.p;
        ld = plan->n_cols;
    }

    // ---- batch length statistics (decides workgroup size and the LDS carve) ----
    long long h_stats[3] = {0, (1LL << 62), 0};
    if (plan->hint_max_len > 0) {
        // the caller vouches for the length range of every batch (tsfa_plan_set_length_hint): no scan, no host sync --
        // back-to-back calls on one stream (chunks of a shard) are enqueued without waiting for each other
        h_stats[0] = plan->hint_max_len;
        h_stats[1] = plan->hint_min_len;
        h_stats[2] = (plan->hint_min_len == plan->hint_max_len && (plan->hint_max_len & (plan->hint_max_len - 1)) == 0)
                         ? 0 : plan->hint_max_len;  // longest length that may not be a power of two
    } else {
        HIP_TRY(hipMemcpyAsync(plan->d_stats, h_stats, sizeof h_stats, hipMemcpyHostToDevice, st));
        if (tsfa_launch_len_stats(d_starts, d_ends, n_series, plan->d_stats, st)) return fail(TSFA_ERR_HIP, "len_stats launch failed");
        HIP_TRY(hipMemcpyAsync(h_stats, plan->d_stats, sizeof h_stats, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    const long long max_len = h_stats[0], min_len = h_stats[1], max_np2 = h_stats[2];
    if (min_len < 1) return fail(TSFA_ERR_INVALID, "every series must hold at least one sample");
    if (max_len > 65535) return fail(TSFA_ERR_TOO_LONG, "series longer than 65535 samples are not supported");
    const int maxn = (int)max_len;
    const int nt = (maxn <= 2048) ? 64 : 256;

    if (tsfa_launch_fill_nan(d_out, n_series * ld, st)) return fail(TSFA_ERR_HIP, "fill launch failed");

    // Launch order: longest kernels first.  With side streams (and no per-kernel timing requested) the families are
    // dealt round-robin over the streams after a fork event; the join events bring them back to `st`.
    static const int order[TSFA_N_FAMILIES] = {TSFA_FAM_ENTROPY, TSFA_FAM_AR, TSFA_FAM_SORT, TSFA_FAM_CWT, TSFA_FAM_BASIC,
                                               TSFA_FAM_SEQ, TSFA_FAM_SPECTRAL, TSFA_FAM_TREND};
    const bool overlap = plan->n_streams > 1 && !plan->profiling;
    if (overlap) {
        HIP_TRY(hipEventRecord(plan->ev_fork, st));
        for (int i = 0; i + 1 < plan->n_streams; ++i) HIP_TRY(hipStreamWaitEvent(plan->aux[i], plan->ev_fork, 0));
    }
    size_t slot = 0;
    int dealt = 0;
    for (int fi = 0; fi < TSFA_N_FAMILIES; ++fi) {
        const int f = order[fi];
        if (plan->fam_specs[f].empty()) continue;
        hipStream_t fst = st;
        if (overlap) {
            const int k = dealt++ % plan->n_streams;
            if (k > 0) fst = plan->aux[k - 1];
        }
        TsfaLaunch a;
        memset(&a, 0, sizeof a);
        a.fam = f;
        a.dtype = dtype;
        a.values = d_values;
        a.starts = d_starts;
        a.ends = d_ends;
        a.n_series = n_series;
        a.specs = plan->d_specs[f];
        a.nspecs = (int)plan->fam_specs[f].size();
        a.out = d_out;
        a.ld = ld;
        a.maxn = maxn;
        a.nt = nt;
        if (maxn <= 2048) {
            // Wavefronts per series, measured on MI355X at n = 1024 (profiles/r01_*): the LDS footprint of a series
            // caps the workgroups per CU, so the latency-bound families gain from more wavefronts per workgroup,
            // while k_basic's many short reductions lose to the extra barriers.  At least 4 samples per thread.
            static const int pref[TSFA_N_FAMILIES] = {64, 128, 128, 128, 256, 256, 128, 64};
            const int cap = std::max(64, ((maxn / 4 + 63) / 64) * 64);
            a.nt = std::min(pref[f], cap);
        }
        {   // experiment hook: TSFA_NT_<family index>=<threads>
            char key[32];
            snprintf(key, sizeof key, "TSFA_NT_%d", f);
            const char *e = getenv(key);
            if (e && atoi(e) >= 64) a.nt = atoi(e);
        }
        a.stream = fst;
        a.dectab = plan->d_dectab;
        a.times = d_times;
        a.twc = plan->d_twc;
        a.tws = plan->d_tws;
        a.hint_a = plan->hints[f].a;
        a.hint_b = plan->hints[f].b;
        a.hint_c = plan->hints[f].c;
        a.hint_d = plan->hints[f].d;
        a.hint_e = plan->hints[f].e;
        a.alt = plan->hints[f].alt;
        a.cq = plan->hints[f].cq;
        int aux = 0;
        if (f == TSFA_FAM_SPECTRAL) {
            // only non-power-of-two lengths <= 256 use the table-driven DFT (longer ones: Goertzel, no table)
            a.dft_n = (int)std::min<long long>(max_np2, 256);
            aux = a.dft_n;
        } else if (f == TSFA_FAM_CWT) {
            a.cwt_rowv = tsfa_family_lds_bytes(f, maxn, a.nt, 1) <= 96 * 1024 ? 1 : 0;
            aux = a.cwt_rowv;
        } else if (f == TSFA_FAM_AR) {
            // leading dimension of the normal matrices: ADF needs maxlag(n) + 3, AR(k) needs k + 2
            int P = 8;
            for (const auto &s : plan->fam_specs[f]) {
                if (s.calc == TSFA_C_AUGMENTED_DICKEY_FULLER) {
                    int ml = (int)ceil(12.0 * pow((double)maxn / 100.0, 0.25));
                    if (maxn / 2 - 2 < ml) ml = maxn / 2 - 2;
                    P = std::max(P, ml + 3);
                } else if (s.calc == TSFA_C_AR_COEFFICIENT) {
                    P = std::max(P, (int)s.p[1] + 2);
                }
            }
            a.ar_P = P;
            aux = P;
            for (const auto &s : plan->fam_specs[f])
                if (s.calc == TSFA_C_AR_COEFFICIENT) a.ar_has_coef = 1;
            if (plan->deg_list.ensure((size_t)n_series * sizeof(long long))) return fail(TSFA_ERR_HIP, "hipMalloc failed for the k_ar_degenerate list");
            a.deg_list = (long long *)plan->deg_list.p;
            a.deg_count = plan->d_deg_count;
            HIP_TRY(hipMemsetAsync(plan->d_deg_count, 0, sizeof(int), fst));
        } else if (f == TSFA_FAM_ENTROPY) {
            // one wavefront per 64-template row block, up to four per series; the symmetric sweep needs 12 B of
            // LDS counters per sample
            const int waves = std::min(4, std::max(1, (maxn - 1 + 63) / 64));
            a.nt = std::max(a.nt, 64 * waves);
            a.ent_cnt = tsfa_entropy_lds_bytes(maxn, 1) <= TSFA_LDS_LIMIT ? 1 : 0;
            a.ent_fast = a.ent_cnt;
            for (const auto &s : plan->fam_specs[f])
                if (s.calc == TSFA_C_APPROXIMATE_ENTROPY && (int)s.p[0] != 2) a.ent_fast = 0;
            if (getenv("TSFA_ENT_SLOW")) a.ent_fast = 0;  // experiment / test hook: the general kernel
        }
        // SEQ: one launch parses up to TSFA_LZ_MAX_GROUP `bins` values side by side -- as many as LDS allows
        int seq_group = 0;
        if (f == TSFA_FAM_SEQ) {
            for (seq_group = std::min(a.nspecs, TSFA_LZ_MAX_GROUP); seq_group > 1; --seq_group) {
                bool fits = true;
                for (int s0 = 0; fits && s0 < a.nspecs; s0 += seq_group) {
                    lz_build_group(plan->fam_specs[f].data() + s0, std::min(seq_group, a.nspecs - s0), maxn, &a.seq);
                    fits = tsfa_seq_lds_bytes(a.seq) <= TSFA_LDS_LIMIT;
                }
                if (fits) break;
            }
            lz_build_group(plan->fam_specs[f].data(), std::min(seq_group, a.nspecs), maxn, &a.seq);
        }
        const size_t lds = (f == TSFA_FAM_SEQ) ? tsfa_seq_lds_bytes(a.seq)
                           : (f == TSFA_FAM_ENTROPY) ? tsfa_entropy_lds_bytes(maxn, a.ent_cnt)
                                                     : tsfa_family_lds_bytes(f, maxn, a.nt, aux);
        if (lds > TSFA_LDS_LIMIT)
            return fail(TSFA_ERR_TOO_LONG, std::string(fam_names[f]) + ": a series of " + std::to_string(maxn) +
                                               " samples needs " + std::to_string(lds) + " B of LDS (limit 163840)");
        if (record(plan, fst, slot, fam_names[f], true)) return fail(TSFA_ERR_HIP, "event record failed");
        int rc = 0;
        if (f == TSFA_FAM_SEQ) {
            for (int s0 = 0; rc == 0 && s0 < a.nspecs; s0 += seq_group) {
                lz_build_group(plan->fam_specs[f].data() + s0, std::min(seq_group, a.nspecs - s0), maxn, &a.seq);
                if (tsfa_seq_lds_bytes(a.seq) > TSFA_LDS_LIMIT)
                    return fail(TSFA_ERR_TOO_LONG, "k_seq: a series of " + std::to_string(maxn) + " samples does not fit LDS");
                rc = tsfa_launch_family(a);
            }
        } else {
            rc = tsfa_launch_family(a);
        }
        if (rc) return fail(TSFA_ERR_HIP, std::string(fam_names[f]) + " launch failed: " + hipGetErrorString((hipError_t)rc));
        if (record(plan, fst, slot, fam_names[f], false)) return fail(TSFA_ERR_HIP, "event record failed");
        ++slot;
    }
    if (overlap) {
        for (int i = 0; i + 1 < plan->n_streams; ++i) {
            HIP_TRY(hipEventRecord(plan->ev_join[i], plan->aux[i]));
            HIP_TRY(hipStreamWaitEvent(st, plan->ev_join[i], 0));
        }
    }
    if (plan->bank.C > 0) {
        TsfaCwtLaunch c;
        memset(&c, 0, sizeof c);
        c.dtype = dtype;
        c.values = d_values;
        c.starts = d_starts;
        c.ends = d_ends;
        c.n_series = n_series;
        c.W = plan->d_W;
        c.S4 = plan->bank.S4;
        c.C = plan->bank.C;
        c.cols = plan->d_cols;
        c.coeff_idx = plan->d_coeff;
        c.out = d_out;
        c.ld = ld;
        c.stream = st;
        if (record(plan, st, slot, "k_cwt_gemm", true)) return fail(TSFA_ERR_HIP, "event record failed");
        const int rc = tsfa_launch_cwt(c);
        if (rc) return fail(TSFA_ERR_HIP, "k_cwt_gemm launch failed");
        if (record(plan, st, slot, "k_cwt_gemm", false)) return fail(TSFA_ERR_HIP, "event record failed");
        ++slot;
    }
    if (plan->profiling) plan->timings.resize(slot);

    if (space == TSFA_HOST) {
        HIP_TRY(hipMemcpy2DAsync(out, (size_t)ld_out * sizeof(double), d_out, (size_t)ld * sizeof(double),
                                 (size_t)plan->n_cols * sizeof(double), (size_t)n_series, hipMemcpyDeviceToHost, st));
    }
    const bool async = (space == TSFA_DEVICE && stream != nullptr);
    if (!async || plan->profiling) {
        HIP_TRY(hipStreamSynchronize(st));
        if (plan->profiling)
            for (auto &t : plan->timings) (void)hipEventElapsedTime(&t.ms, t.e0, t.e1);
    }
    return TSFA_OK;
}

}  // extern "C"
