// Block-cooperative primitives for the per-series feature kernels.
//
// Execution model: ONE workgroup (64 or 256 threads = 1 or 4 wavefronts) owns ONE series, which it
// stages once from HBM into LDS as float64 and then sweeps repeatedly.  Every function below is
// "block-uniform": all threads of the workgroup call it with the same control flow.
//
// The same source is compiled two ways:
//   * hipcc --offload-arch=gfx950  -> the product (device code; wave64 shuffles + LDS)
//   * g++ -DTSFA_EMUL              -> a single-thread (nt = 1) emulation used ONLY by tests/ to check
//                                     the kernel logic against the oracle on a box without a GPU.
//                                     It is never loaded by the tsfresh_amd package.
#ifndef TSFA_COMMON_H
#define TSFA_COMMON_H

#include <math.h>
#include <stdint.h>

#include "tsfa_specs.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TSFA_GPU 1
#define TSFA_DEV __device__ __forceinline__
#define TSFA_DEVN __device__ __noinline__
#define TSFA_MEM __device__ __forceinline__
#else
#define TSFA_GPU 0
#define TSFA_DEV static inline
#define TSFA_DEVN static
#define TSFA_MEM inline
#endif
// p points into LDS (for out-of-line device functions, whose pointer arguments are generic: InferAddressSpaces turns the
// accesses behind such an assumption into ds_* instructions)
#if defined(__HIP_DEVICE_COMPILE__)
#define TSFA_ASSUME_LDS(p) __builtin_assume(__builtin_amdgcn_is_shared((const void *)(p)))
#else
#define TSFA_ASSUME_LDS(p) ((void)0)
#endif

#define TSFA_NAN (__builtin_nan(""))
#define TSFA_INF (__builtin_inf())

// LDS scratch sizes (in doubles)
#define TSFA_RED_DOUBLES 64
#define TSFA_NP_MAXLEAF 168

struct NpScratch {  // scratch for the numpy-order pairwise sum
    double leaf_sum[TSFA_NP_MAXLEAF];
    double vst[32];
    int leaf_off[TSFA_NP_MAXLEAF];
    int leaf_len[TSFA_NP_MAXLEAF];
    int st_o[32], st_l[32], st_v[32];
    int nleaf;
    int pad;
    double result;
};

// The series as the kernels see it: LDS-resident in its INPUT precision (float32 samples take half the LDS of their
// float64 image, and LDS capacity is what limits the resident series per CU), read as float64 -- the conversion is
// exact, so every expression downstream is the float64 arithmetic of the reference on x.astype(float64).
template <typename ST>
struct XsView {
    typedef ST elem;
    const ST *p;
    TSFA_MEM double operator[](int i) const { return (double)p[i]; }
};

struct Blk {
    int tid;        // thread index in the workgroup
    int nt;         // workgroup size
    double *red;    // LDS, TSFA_RED_DOUBLES doubles: cross-wave reduction scratch + broadcast slot
    NpScratch *np;  // LDS
};

// ROW FORM (round 6): a series of at most TSFA_ROW_MAXN samples is owned by ONE 16-lane DPP row of a wavefront, four series
// per wavefront.  At 256 samples a 64-lane workgroup holds four samples per lane and spends its time on what is per COLUMN --
// the scalar spec fetch and dispatch, a cross-lane reduction with its v_readlane stage, the closing float64 divisions, the
// store (k_basic: 3.14 ms per 125 000 x 256 against 3.09 ms per 100 000 x 1024).  In the row form those
// instructions are issued once for four series; a reduction is the four DPP steps inside the row and ends there.
// A BlkRow is a Blk whose `tid` / `nt` are the lane within the row and 16; the overloads below (chosen by the static type,
// so every function between the kernel and a primitive is a template on the block type) never leave the row.  Which form
// evaluates a series depends on ITS length alone (n <= TSFA_ROW_MAXN: always the row form), never on what else the launch
// group or the shard holds: a series gives the same bits wherever it lands.
#define TSFA_ROW_MAXN 256
#define TSFA_ROW_LANES 16
struct BlkRow : Blk {};
template <class BT> struct BlkLanes { static constexpr int n = 64; };
template <> struct BlkLanes<BlkRow> { static constexpr int n = TSFA_ROW_LANES; };
template <class BT> struct BlkIsRow { static constexpr bool v = false; };
template <> struct BlkIsRow<BlkRow> { static constexpr bool v = true; };
TSFA_DEV Blk blk_rebind(const Blk &b, int tid) { return Blk{tid, b.nt, b.red, b.np}; }
TSFA_DEV BlkRow blk_rebind(const BlkRow &b, int tid) { BlkRow r; r.tid = tid; r.nt = b.nt; r.red = b.red; r.np = b.np; return r; }

// Phase clocks (diagnostics build only: make ticks -> libtsfresh_amd_ticks.so, read with tsfa_debug_ticks).
// Thread 0 of every workgroup adds the shader-clock cycles since its previous mark to a global counter per phase id,
// so one bench pass yields the per-phase latency breakdown of every kernel (rocprofv3 only sees whole kernels).
#if TSFA_GPU && defined(TSFA_TICKS)
static __device__ unsigned long long tsfa_ticks[256];
static __shared__ unsigned long long tsfa_ticks_lds[256];  // per-workgroup accumulation; flushed once at kernel end
struct Ticker {
    unsigned long long t;
    int base;
    __device__ __forceinline__ explicit Ticker(int base_) : t(__builtin_readcyclecounter()), base(base_) {}
    __device__ __forceinline__ void mark(int tid, int id) {
        const unsigned long long now = __builtin_readcyclecounter();
        if (tid == 0) tsfa_ticks_lds[base + id] += now - t;
        t = __builtin_readcyclecounter();
    }
};
#define TSFA_TICKER(name, base) Ticker name(base)
#define TSFA_TICK(name, b, id) name.mark((b).tid, id)
#define TSFA_TICKS_BEGIN()                                                                  \
    do {                                                                                    \
        for (int i_ = threadIdx.x; i_ < 256; i_ += blockDim.x) tsfa_ticks_lds[i_] = 0ull;   \
        __syncthreads();                                                                    \
    } while (0)
#define TSFA_TICKS_END()                                                                    \
    do {                                                                                    \
        __syncthreads();                                                                    \
        for (int i_ = threadIdx.x; i_ < 256; i_ += blockDim.x)                              \
            if (tsfa_ticks_lds[i_]) atomicAdd(&tsfa_ticks[i_], tsfa_ticks_lds[i_]);         \
    } while (0)
#else
#define TSFA_TICKER(name, base)
#define TSFA_TICK(name, b, id)
#define TSFA_TICKS_BEGIN()
#define TSFA_TICKS_END()
#endif

// Workgroup barrier for data exchanged through LDS.  __syncthreads() also drains the wave's outstanding GLOBAL
// stores (s_waitcnt vmcnt(0)): every feature ends with a store of its value to the HBM output row, so the next
// barrier would expose that store's round trip (~600 cycles per column, measured).  LDS traffic only needs lgkmcnt.
TSFA_DEV void blk_sync() {
#if TSFA_GPU
#if defined(TSFA_LONG)
    // the working set lives in HBM scratch: the barrier must order global memory too.  The explicit wait is for the
    // one-wavefront workgroups (k_general): there the compiler drops __syncthreads() to a "wave barrier" with no
    // s_waitcnt at all, and a lane's load can pass the store of another lane of the same wavefront.
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
#endif
}
// Barrier that also orders global memory within the workgroup (scratch in HBM).
TSFA_DEV void blk_sync_all() {
#if TSFA_GPU
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
#endif
}

// ---------------------------------------------------------------------------------------------
// reductions: every thread receives the result
// ---------------------------------------------------------------------------------------------
#if TSFA_GPU
// Wave64 all-reduce without LDS traffic.  __shfl_xor on a double lowers to two ds_bpermute_b32 (LDS crossbar, ~64
// cycles of latency each) per step; the first four butterfly steps stay inside a 16-lane DPP row, so they are done
// with DPP modifiers (quad_perm / row_half_mirror / row_mirror: VALU latency), and the four row totals are then read
// with v_readlane and added -- the result is wave-uniform (SGPR) by construction.
template <int CTRL>
TSFA_DEV double dpp_mov_f64(double v) {
    union { double d; int i[2]; } a, r;
    a.d = v;
    r.i[0] = __builtin_amdgcn_update_dpp(a.i[0], a.i[0], CTRL, 0xf, 0xf, false);
    r.i[1] = __builtin_amdgcn_update_dpp(a.i[1], a.i[1], CTRL, 0xf, 0xf, false);
    return r.d;
}
TSFA_DEV double readlane_f64(double v, int lane) {
    union { double d; int i[2]; } a, r;
    a.d = v;
    r.i[0] = __builtin_amdgcn_readlane(a.i[0], lane);
    r.i[1] = __builtin_amdgcn_readlane(a.i[1], lane);
    return r.d;
}
#define TSFA_DPP_QUAD_XOR1 0xB1       /* quad_perm:[1,0,3,2] */
#define TSFA_DPP_QUAD_XOR2 0x4E       /* quad_perm:[2,3,0,1] */
#define TSFA_DPP_ROW_HALF_MIRROR 0x141
#define TSFA_DPP_ROW_MIRROR 0x140
TSFA_DEV double wave_sum(double v) {
    v += dpp_mov_f64<TSFA_DPP_QUAD_XOR1>(v);
    v += dpp_mov_f64<TSFA_DPP_QUAD_XOR2>(v);
    v += dpp_mov_f64<TSFA_DPP_ROW_HALF_MIRROR>(v);
    v += dpp_mov_f64<TSFA_DPP_ROW_MIRROR>(v);
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
TSFA_DEV double wave_min(double v) {
    v = fmin(v, dpp_mov_f64<TSFA_DPP_QUAD_XOR1>(v));
    v = fmin(v, dpp_mov_f64<TSFA_DPP_QUAD_XOR2>(v));
    v = fmin(v, dpp_mov_f64<TSFA_DPP_ROW_HALF_MIRROR>(v));
    v = fmin(v, dpp_mov_f64<TSFA_DPP_ROW_MIRROR>(v));
    return fmin(fmin(readlane_f64(v, 0), readlane_f64(v, 16)), fmin(readlane_f64(v, 32), readlane_f64(v, 48)));
}
TSFA_DEV double wave_max(double v) {
    v = fmax(v, dpp_mov_f64<TSFA_DPP_QUAD_XOR1>(v));
    v = fmax(v, dpp_mov_f64<TSFA_DPP_QUAD_XOR2>(v));
    v = fmax(v, dpp_mov_f64<TSFA_DPP_ROW_HALF_MIRROR>(v));
    v = fmax(v, dpp_mov_f64<TSFA_DPP_ROW_MIRROR>(v));
    return fmax(fmax(readlane_f64(v, 0), readlane_f64(v, 16)), fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}
#endif

TSFA_DEV double blk_sum(const Blk &b, double v) {
#if TSFA_GPU
    v = wave_sum(v);
    if (b.nt > 64) {
        const int nw = b.nt >> 6;
        blk_sync();
        if ((b.tid & 63) == 0) b.red[b.tid >> 6] = v;
        blk_sync();
        v = b.red[0];
        for (int w = 1; w < nw; ++w) v += b.red[w];
    }
#endif
    return v;
}
// Column specs: a scalar (SMEM) load per column, or -- with TSFA_SPEC_LDS -- staged through LDS TSFA_SPEC_BATCH at a
// time.  Measured on MI355X (profiles/phase_ticks.py): the staged variant is ~20% slower per column (two barriers per
// batch, ten readfirstlanes per column), so the scalar load is the default.  Block-uniform: every thread calls it.
#define TSFA_SPEC_BATCH 16
TSFA_DEV TsfaSpec spec_fetch(const Blk &b, const TsfaSpec *specs, int nspecs, int s, TsfaSpec *stage) {
#if TSFA_GPU && defined(TSFA_SPEC_LDS)
    if ((s % TSFA_SPEC_BATCH) == 0) {
        blk_sync();  // every thread is done with the previous batch
        const int cnt = (nspecs - s < TSFA_SPEC_BATCH) ? (nspecs - s) : TSFA_SPEC_BATCH;
        const double *src = (const double *)(const void *)(specs + s);
        double *dst = (double *)(void *)stage;
        for (int i = b.tid; i < cnt * (int)(sizeof(TsfaSpec) / sizeof(double)); i += b.nt) dst[i] = src[i];
        blk_sync();
    }
    const TsfaSpec *e = stage + (s % TSFA_SPEC_BATCH);
    TsfaSpec r;
    r.calc = __builtin_amdgcn_readfirstlane(e->calc);  // wave-uniform by construction: keep the dispatch scalar
    r.col = __builtin_amdgcn_readfirstlane(e->col);
#pragma unroll
    for (int k = 0; k < 4; ++k) r.p[k] = readlane_f64(e->p[k], 0);
    return r;
#else
    (void)b; (void)nspecs; (void)stage;
    return specs[s];
#endif
}

// N sums at once: the wave reductions are independent chains (they overlap), and the cross-wave exchange costs one
// barrier pair for all N instead of one per value.  Same summation order as N calls of blk_sum.  N * waves <= 64.
#if TSFA_GPU
template <int LX>
TSFA_DEV double lane_xor_f64(double v);  // defined with the sorting networks below
#endif
#if TSFA_GPU
// G (4 or 8) per-lane values -> their G wavefront totals, one per lane: a reduce-scatter inside the wavefront.  At each
// of the first log2(G) steps a lane keeps one half of its values and hands the other half to its partner (lane xor 1, 2,
// 4), so the values per lane halve while the lanes per sum double: G - 1 additions + G - 1 exchanges instead of G full
// butterflies (6 G of each).  The lane then holds the partial sum of quantity rev(lane mod G) over its G-lane group
// (rev = bit reversal), and log2(64 / G) more steps add the groups.  Returns that total.
template <int G>
TSFA_DEV double wave_reduce_scatter(const double *v) {
    static_assert(G == 4 || G == 8, "groups of four or eight values");
    const int lane = (int)(threadIdx.x & 63);
    double t[G / 2];
    {
        const bool hi = (lane & 1) != 0;
#pragma unroll
        for (int i = 0; i < G / 2; ++i) {
            const double keep = hi ? v[G / 2 + i] : v[i], send = hi ? v[i] : v[G / 2 + i];
            t[i] = keep + lane_xor_f64<1>(send);
        }
    }
    {
        const bool hi = (lane & 2) != 0;
#pragma unroll
        for (int i = 0; i < G / 4; ++i) {
            const double keep = hi ? t[G / 4 + i] : t[i], send = hi ? t[i] : t[G / 4 + i];
            t[i] = keep + lane_xor_f64<2>(send);
        }
    }
    double r = t[0];
    if (G == 8) {
        const bool hi = (lane & 4) != 0;
        const double keep = hi ? t[1] : t[0], send = hi ? t[0] : t[1];
        r = keep + lane_xor_f64<4>(send);
    } else {
        r += lane_xor_f64<4>(r);
    }
    r += lane_xor_f64<8>(r);
    r += lane_xor_f64<16>(r);
    r += lane_xor_f64<32>(r);
    return r;
}
// quantity held by lane l (l < G) after wave_reduce_scatter<G>
template <int G>
TSFA_DEV int wave_reduce_scatter_index(int l) {
    return (G == 8) ? (((l & 1) << 2) | (l & 2) | ((l >> 2) & 1)) : (((l & 1) << 1) | ((l >> 1) & 1));
}
#endif

template <int N>
TSFA_DEV void blk_sum_multi(const Blk &b, double *v) {
#if TSFA_GPU
    // groups of eight (and one of four) values by the wavefront reduce-scatter, the remaining ones by full butterflies
    constexpr int N8 = (N / 8) * 8, N4 = N8 + (((N - N8) >= 4) ? 4 : 0);
    const int lane = b.tid & 63;
    double held[(N4 > 0) ? (N4 / 4) : 1];
    (void)held;
#pragma unroll
    for (int g = 0; g < N8; g += 8) held[g / 8] = wave_reduce_scatter<8>(v + g);
    if (N4 > N8) held[N8 / 8] = wave_reduce_scatter<4>(v + N8);
#pragma unroll
    for (int k = N4; k < N; ++k) v[k] = wave_sum(v[k]);
    if (b.nt <= 64) {
#pragma unroll
        for (int g = 0; g < N8; g += 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[g + k] = readlane_f64(held[g / 8], wave_reduce_scatter_index<8>(k));
        }
        if (N4 > N8) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[N8 + k] = readlane_f64(held[N8 / 8], wave_reduce_scatter_index<4>(k));
        }
    }
    if (b.nt > 64) {
        const int nw = b.nt >> 6;
        blk_sync();
        {
            double *slot = b.red + (b.tid >> 6) * N;
#pragma unroll
            for (int g = 0; g < N8; g += 8)
                if (lane < 8) slot[g + wave_reduce_scatter_index<8>(lane)] = held[g / 8];
            if (N4 > N8 && lane < 4) slot[N8 + wave_reduce_scatter_index<4>(lane)] = held[N8 / 8];
            if (lane == 0) {
#pragma unroll
                for (int k = N4; k < N; ++k) slot[k] = v[k];
            }
        }
        blk_sync();
#pragma unroll
        for (int k = 0; k < N; ++k) {
            double t = b.red[k];
            for (int w = 1; w < nw; ++w) t += b.red[w * N + k];
            v[k] = t;
        }
    }
#endif
}

TSFA_DEV double blk_min(const Blk &b, double v) {
#if TSFA_GPU
    v = wave_min(v);
    if (b.nt > 64) {
        const int nw = b.nt >> 6;
        blk_sync();
        if ((b.tid & 63) == 0) b.red[b.tid >> 6] = v;
        blk_sync();
        v = b.red[0];
        for (int w = 1; w < nw; ++w) v = fmin(v, b.red[w]);
    }
#endif
    return v;
}
TSFA_DEV double blk_max(const Blk &b, double v) {
#if TSFA_GPU
    v = wave_max(v);
    if (b.nt > 64) {
        const int nw = b.nt >> 6;
        blk_sync();
        if ((b.tid & 63) == 0) b.red[b.tid >> 6] = v;
        blk_sync();
        v = b.red[0];
        for (int w = 1; w < nw; ++w) v = fmax(v, b.red[w]);
    }
#endif
    return v;
}

// Exclusive count of `flag` over the lower-numbered threads of the workgroup, and the workgroup total.
// (thread order = ascending tid; used to hand out indices in a deterministic order)
TSFA_DEV int blk_excl_count(const Blk &b, bool flag, int *total) {
#if TSFA_GPU
    const unsigned long long m = __ballot(flag);
    const int lane = b.tid & 63;
    int pre = __popcll(m & ((1ull << lane) - 1ull));
    int tot = __popcll(m);
    if (b.nt > 64) {
        const int nw = b.nt >> 6, w = b.tid >> 6;
        int *ir = (int *)(b.red + 32);
        blk_sync();
        if (lane == 0) ir[w] = tot;
        blk_sync();
        int base = 0, all = 0;
        for (int k = 0; k < nw; ++k) {
            if (k < w) base += ir[k];
            all += ir[k];
        }
        pre += base;
        tot = all;
    }
    *total = tot;
    return pre;
#else
    *total = flag ? 1 : 0;
    return 0;
#endif
}

// Exclusive prefix of a small per-thread count (0 <= cnt < 2^bits) over the lower-numbered threads, and the workgroup
// total: a thread that owns several consecutive items hands out indices for all of them with ONE scan.
TSFA_DEV int blk_excl_sum_small(const Blk &b, int cnt, int bits, int *total) {
#if TSFA_GPU
    const int lane = b.tid & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    int pre = 0, tot = 0;
    for (int k = 0; k < bits; ++k) {
        const unsigned long long m = __ballot((cnt >> k) & 1);
        pre += __popcll(m & below) << k;
        tot += __popcll(m) << k;
    }
    if (b.nt > 64) {
        const int nw = b.nt >> 6, w = b.tid >> 6;
        int *ir = (int *)(b.red + 32);
        blk_sync();
        if (lane == 0) ir[w] = tot;
        blk_sync();
        int base = 0, all = 0;
        for (int k = 0; k < nw; ++k) {
            if (k < w) base += ir[k];
            all += ir[k];
        }
        pre += base;
        tot = all;
    }
    *total = tot;
    return pre;
#else
    *total = cnt;
    return 0;
#endif
}

// bitwise OR of a 16-bit value over the workgroup (every thread receives it)
TSFA_DEV unsigned blk_or16(const Blk &b, unsigned v) {
#if TSFA_GPU
    unsigned r = 0;
    for (int k = 0; k < 16; ++k) r |= (__ballot((v >> k) & 1u) != 0ull) ? (1u << k) : 0u;
    if (b.nt > 64) {
        const int nw = b.nt >> 6;
        int *ir = (int *)(b.red + 32);
        blk_sync();
        if ((b.tid & 63) == 0) ir[b.tid >> 6] = (int)r;
        blk_sync();
        r = 0;
        for (int k = 0; k < nw; ++k) r |= (unsigned)ir[k];
    }
    return r;
#else
    return v;
#endif
}

// compare-exchange: a <- min, b <- max.  The hardware minimum / maximum directly: the compiler's fmin / fmax spend a
// third instruction per pair on canonicalising an operand
TSFA_DEV void ce_f64(double &a, double &b2) {
#if TSFA_GPU
    double lo, hi;
    asm("v_min_f64 %0, %2, %3\n\tv_max_f64 %1, %2, %3" : "=&v"(lo), "=&v"(hi) : "v"(a), "v"(b2));
    a = lo;
    b2 = hi;
#else
    const double lo = fmin(a, b2), hi = fmax(a, b2);
    a = lo;
    b2 = hi;
#endif
}
TSFA_DEV double min_f64(double a, double b2) {
#if TSFA_GPU
    double lo;
    asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(a), "v"(b2));
    return lo;
#else
    return fmin(a, b2);
#endif
}

// broadcast a value computed by thread 0 to the whole workgroup
TSFA_DEV double blk_bcast0(const Blk &b, double v) {
#if TSFA_GPU
    if (b.nt == 64) return readlane_f64(v, 0);
    blk_sync();
    if (b.tid == 0) b.red[TSFA_RED_DOUBLES - 1] = v;
    blk_sync();
    v = b.red[TSFA_RED_DOUBLES - 1];
#endif
    return v;
}

#if TSFA_GPU
// ---------------------------------------------------------------------------------------------
// row form: every lane of the 16-lane row receives the result; nothing crosses a row
// ---------------------------------------------------------------------------------------------
TSFA_DEV double row_sum_f64(double v) {
    v += dpp_mov_f64<TSFA_DPP_QUAD_XOR1>(v);
    v += dpp_mov_f64<TSFA_DPP_QUAD_XOR2>(v);
    v += dpp_mov_f64<TSFA_DPP_ROW_HALF_MIRROR>(v);
    v += dpp_mov_f64<TSFA_DPP_ROW_MIRROR>(v);
    return v;
}
template <int CTRL>
TSFA_DEV int dpp_mov_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
TSFA_DEV int row_sum_i32(int v) {
    v += dpp_mov_i32<TSFA_DPP_QUAD_XOR1>(v);
    v += dpp_mov_i32<TSFA_DPP_QUAD_XOR2>(v);
    v += dpp_mov_i32<TSFA_DPP_ROW_HALF_MIRROR>(v);
    v += dpp_mov_i32<TSFA_DPP_ROW_MIRROR>(v);
    return v;
}
TSFA_DEV double blk_sum(const BlkRow &, double v) { return row_sum_f64(v); }
TSFA_DEV double blk_min(const BlkRow &, double v) {
    v = fmin(v, dpp_mov_f64<TSFA_DPP_QUAD_XOR1>(v));
    v = fmin(v, dpp_mov_f64<TSFA_DPP_QUAD_XOR2>(v));
    v = fmin(v, dpp_mov_f64<TSFA_DPP_ROW_HALF_MIRROR>(v));
    return fmin(v, dpp_mov_f64<TSFA_DPP_ROW_MIRROR>(v));
}
TSFA_DEV double blk_max(const BlkRow &, double v) {
    v = fmax(v, dpp_mov_f64<TSFA_DPP_QUAD_XOR1>(v));
    v = fmax(v, dpp_mov_f64<TSFA_DPP_QUAD_XOR2>(v));
    v = fmax(v, dpp_mov_f64<TSFA_DPP_ROW_HALF_MIRROR>(v));
    return fmax(v, dpp_mov_f64<TSFA_DPP_ROW_MIRROR>(v));
}
template <int N>
TSFA_DEV void blk_sum_multi(const BlkRow &, double *v) {
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = row_sum_f64(v[k]);
}
TSFA_DEV unsigned blk_or16(const BlkRow &, unsigned v) {
    int r = (int)v;
    r |= dpp_mov_i32<TSFA_DPP_QUAD_XOR1>(r);
    r |= dpp_mov_i32<TSFA_DPP_QUAD_XOR2>(r);
    r |= dpp_mov_i32<TSFA_DPP_ROW_HALF_MIRROR>(r);
    r |= dpp_mov_i32<TSFA_DPP_ROW_MIRROR>(r);
    return (unsigned)r;
}
// the value of the row's lane `src` (0 .. 15) in every lane of the row
TSFA_DEV double row_bcast_f64(double v, int src) { return __shfl(v, src, TSFA_ROW_LANES); }
TSFA_DEV double blk_bcast0(const BlkRow &, double v) { return row_bcast_f64(v, 0); }
#endif

// ---------------------------------------------------------------------------------------------
// numpy-order summation.
//
// np.sum / np.mean / np.var on a contiguous float64 array (numpy/_core/src/umath/loops_utils.h.src,
// DOUBLE_pairwise_sum; verified against numpy 2.2.6 in tests/test_oracle_numpy_order.py) add in a fixed
// order: blocks of 8192 elements are accumulated serially; inside a block the array is halved
// recursively (left half rounded down to a multiple of 8) until <= 128 elements remain, which are
// summed with 8 strided accumulators.  Features that COMPARE against mean/std (count_above_mean,
// longest_strike_*, large_standard_deviation, symmetry_looking, ratio_beyond_r_sigma, ...) flip on a
// 1-ulp difference for "nice" data (constant series of 0.1, small decimals), so mean and variance are
// summed in exactly this order.  F(i) returns element i.
// ---------------------------------------------------------------------------------------------
template <class F>
TSFA_DEV double np_leaf_sum(int o, int l, F f) {
    if (l < 8) {
        double r = 0.0;
        for (int i = 0; i < l; ++i) r += f(o + i);
        return r;
    }
    double r0 = f(o), r1 = f(o + 1), r2 = f(o + 2), r3 = f(o + 3);
    double r4 = f(o + 4), r5 = f(o + 5), r6 = f(o + 6), r7 = f(o + 7);
    int i = 8;
    const int lim = l - (l % 8);
    for (; i < lim; i += 8) {
        r0 += f(o + i);
        r1 += f(o + i + 1);
        r2 += f(o + i + 2);
        r3 += f(o + i + 3);
        r4 += f(o + i + 4);
        r5 += f(o + i + 5);
        r6 += f(o + i + 6);
        r7 += f(o + i + 7);
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < l; ++i) res += f(o + i);
    return res;
}

template <class F>
TSFA_DEV double np_sum(const Blk &b, int n, F f) {
    NpScratch *s = b.np;
    double total = 0.0;
    if (n <= 0) return 0.0;
    for (int c0 = 0; c0 < n; c0 += 8192) {
        const int clen = (n - c0 < 8192) ? (n - c0) : 8192;
        blk_sync();
#if TSFA_GPU
        {
            // Fast path (every length whose pairwise tree is COMPLETE, e.g. 1024 = 8 leaves of 128): lane j of every
            // wavefront walks from the root to leaf j with index arithmetic only -- no LDS stack, no serial thread --
            // and the leaf sums are combined by log2(leaves) butterfly steps, which is the recursion's own order
            // (sum(left half) + sum(right half) at every level).  All wavefronts do this redundantly, so the result
            // is uniform without a broadcast.  Incomplete trees (leaf depths differ) take the general path below.
            int dl = 0, dr = 0;
            for (int l = clen; l > 128; ++dl) { int n2 = l / 2; n2 -= n2 % 8; l = n2; }
            for (int l = clen; l > 128; ++dr) { int n2 = l / 2; n2 -= n2 % 8; l = l - n2; }
            bool complete = (dl == dr) && (dl <= 6);
            const int d = dl, nl = 1 << d, lane = b.tid & 63;
            int lo = c0, ll = clen, lp = 1 << 30;  // my leaf's offset / length, its parent's length
            if (complete) {
                for (int lev = d - 1; lev >= 0; --lev) {
                    int n2 = ll / 2;
                    n2 -= n2 % 8;
                    lp = ll;
                    if ((lane >> lev) & 1) { lo += n2; ll -= n2; } else { ll = n2; }
                }
                const bool ok = (lane >= nl) || (ll <= 128 && (d == 0 || lp > 128));
                complete = (__ballot(ok) == ~0ull);
            }
            if (complete && nl <= 8) {
                // up to 8 leaves (n <= 1024): 8 leaves x 8 accumulators = the 64 lanes of ONE wavefront.  Every wavefront
                // evaluates the whole tree on its own -- no LDS exchange, no barrier (three per call otherwise); the
                // additions and their order are those of the general path.
                const int leaf = lane >> 3, k = lane & 7;
                const bool live = leaf < nl;
                const int o = __shfl(lo, leaf), l = live ? __shfl(ll, leaf) : 0;
                double r = 0.0;
                if (l >= 8) {
                    const int lim = l - (l % 8);
                    r = f(o + k);
                    for (int i = 8; i < lim; i += 8) r += f(o + i + k);
                }
                r += dpp_mov_f64<TSFA_DPP_QUAD_XOR1>(r);
                r += dpp_mov_f64<TSFA_DPP_QUAD_XOR2>(r);
                r += dpp_mov_f64<TSFA_DPP_ROW_HALF_MIRROR>(r);
                if (live && k == 0) {
                    if (l < 8) {
                        r = 0.0;
                        for (int i = 0; i < l; ++i) r += f(o + i);
                    } else {
                        for (int i = l - (l % 8); i < l; ++i) r += f(o + i);
                    }
                }
                // leaf sums sit in lanes 0, 8, 16, ...: the recursion's pairwise order is a butterfly over lane bits 3, 4, 5
                double v = live ? r : 0.0;
                if (nl > 1) v = v + lane_xor_f64<8>(v);
                if (nl > 2) v = v + lane_xor_f64<16>(v);
                if (nl > 4) v = v + lane_xor_f64<32>(v);
                const double chunk = readlane_f64(v, 0);
                total = (c0 == 0) ? chunk : (total + chunk);
                if (c0 + 8192 >= n) return total;
                continue;
            }
            if (complete) {
                for (int u0 = 0; u0 < nl * 8; u0 += b.nt) {
                    const int u = u0 + b.tid;
                    const int leaf = u >> 3, k = u & 7;
                    const bool live = leaf < nl;
                    const int o = __shfl(lo, leaf & 63), l = live ? __shfl(ll, leaf & 63) : 0;
                    double r = 0.0;
                    if (l >= 8) {
                        const int lim = l - (l % 8);
                        r = f(o + k);
                        for (int i = 8; i < lim; i += 8) r += f(o + i + k);
                    }
                    r += dpp_mov_f64<TSFA_DPP_QUAD_XOR1>(r);
                    r += dpp_mov_f64<TSFA_DPP_QUAD_XOR2>(r);
                    r += dpp_mov_f64<TSFA_DPP_ROW_HALF_MIRROR>(r);
                    if (live && k == 0) {
                        if (l < 8) {
                            r = 0.0;
                            for (int i = 0; i < l; ++i) r += f(o + i);
                        } else {
                            for (int i = l - (l % 8); i < l; ++i) r += f(o + i);
                        }
                        s->leaf_sum[leaf] = r;
                    }
                }
                blk_sync();
                double v = (lane < nl) ? s->leaf_sum[lane] : 0.0;
                for (int bit = 1; bit < nl; bit <<= 1) v = v + __shfl_xor(v, bit);
                const double chunk = readlane_f64(v, 0);
                total = (c0 == 0) ? chunk : (total + chunk);
                if (c0 + 8192 >= n) {
                    blk_sync();  // leaf_sum may be reused by the next call
                    return total;
                }
                continue;
            }
        }
#endif
        if (b.tid == 0) {  // enumerate the leaves of the pairwise tree, left to right
            int sp = 0, nl = 0;
            s->st_o[0] = c0;
            s->st_l[0] = clen;
            sp = 1;
            while (sp > 0) {
                --sp;
                const int o = s->st_o[sp], l = s->st_l[sp];
                if (l <= 128) {
                    s->leaf_off[nl] = o;
                    s->leaf_len[nl] = l;
                    ++nl;
                } else {
                    int n2 = l / 2;
                    n2 -= n2 % 8;
                    s->st_o[sp] = o + n2;
                    s->st_l[sp] = l - n2;
                    ++sp;
                    s->st_o[sp] = o;
                    s->st_l[sp] = n2;
                    ++sp;
                }
            }
            s->nleaf = nl;
        }
        blk_sync();
        const int nl = s->nleaf;
#if TSFA_GPU
        // 8 lanes per leaf: lane k of a group owns numpy's accumulator r[k]; the fixed combination tree
        // ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) is three DPP butterfly steps inside the 8-lane group
        for (int u0 = 0; u0 < nl * 8; u0 += b.nt) {
            const int u = u0 + b.tid;
            const int leaf = u >> 3, k = u & 7;
            const bool live = leaf < nl;
            const int o = live ? s->leaf_off[leaf] : 0, l = live ? s->leaf_len[leaf] : 0;
            double r = 0.0;
            if (l >= 8) {
                const int lim = l - (l % 8);
                r = f(o + k);
                for (int i = 8; i < lim; i += 8) r += f(o + i + k);
            }
            r += dpp_mov_f64<TSFA_DPP_QUAD_XOR1>(r);
            r += dpp_mov_f64<TSFA_DPP_QUAD_XOR2>(r);
            r += dpp_mov_f64<TSFA_DPP_ROW_HALF_MIRROR>(r);
            if (live && k == 0) {
                if (l < 8) {
                    r = 0.0;
                    for (int i = 0; i < l; ++i) r += f(o + i);
                } else {
                    for (int i = l - (l % 8); i < l; ++i) r += f(o + i);
                }
                s->leaf_sum[leaf] = r;
            }
        }
#else
        for (int k = b.tid; k < nl; k += b.nt) s->leaf_sum[k] = np_leaf_sum(s->leaf_off[k], s->leaf_len[k], f);
#endif
        blk_sync();
        if (b.tid == 0) {  // combine in recursion (post-)order
            int sp = 0, vp = 0, li = 0;
            s->st_l[0] = clen;
            s->st_v[0] = 0;
            sp = 1;
            while (sp > 0) {
                const int l = s->st_l[sp - 1];
                if (l <= 128) {
                    --sp;
                    s->vst[vp++] = s->leaf_sum[li++];
                } else if (s->st_v[sp - 1] == 0) {
                    s->st_v[sp - 1] = 1;
                    int n2 = l / 2;
                    n2 -= n2 % 8;
                    s->st_l[sp] = l - n2;
                    s->st_v[sp] = 0;
                    ++sp;
                    s->st_l[sp] = n2;
                    s->st_v[sp] = 0;
                    ++sp;
                } else {
                    --sp;
                    const double r = s->vst[--vp];
                    const double lft = s->vst[--vp];
                    s->vst[vp++] = lft + r;
                }
            }
            const double chunk = s->vst[0];
            s->result = chunk;
        }
        blk_sync();
        total = (c0 == 0) ? s->result : (total + s->result);
    }
    blk_sync();
    return total;
}

#if TSFA_GPU
// np.sum's order for a series of at most TSFA_ROW_MAXN = 256 samples on ONE 16-lane row: the pairwise tree of such a length
// has at most three leaves (129 .. 256 samples: a left half rounded down to a multiple of eight, <= 128, and a right part
// that is split once more when it exceeds 128 -- 255 samples are 120 + (64 + 71)), a leaf is eight strided accumulators
// combined as ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)) plus a serial tail: eight lanes per leaf, two leaves side by
// side, a second turn for the third.  The same additions in the same order as the general path above; no LDS, no barrier.
template <class F>
TSFA_DEV double np_row_leaves(int lane, int o, int l, F f) {   // lanes 0-7: leaf (o, l) -> its sum in lane 0 of the group of eight
    const int k = lane & 7;
    double r = 0.0;
    if (l >= 8) {
        const int lim = l - (l % 8);
        r = f(o + k);
        for (int i = 8; i < lim; i += 8) r += f(o + i + k);
    }
    r += dpp_mov_f64<TSFA_DPP_QUAD_XOR1>(r);
    r += dpp_mov_f64<TSFA_DPP_QUAD_XOR2>(r);
    r += dpp_mov_f64<TSFA_DPP_ROW_HALF_MIRROR>(r);
    if (k == 0) {
        if (l < 8) {
            r = 0.0;
            for (int i = 0; i < l; ++i) r += f(o + i);
        } else {
            for (int i = l - (l % 8); i < l; ++i) r += f(o + i);
        }
    }
    return r;
}
template <class F>
TSFA_DEV double np_sum(const BlkRow &b, int n, F f) {
    if (n <= 0) return 0.0;
    const int lane = b.tid;
    // leaves, left to right
    int o1 = 0, l1 = 0, o2 = 0, l2 = 0, l0 = n, nleaf = 1;
    if (n > 128) {
        int h = n / 2;
        h -= h % 8;
        l0 = h;
        o1 = h;
        l1 = n - h;
        nleaf = 2;
        if (l1 > 128) {
            int h2 = l1 / 2;
            h2 -= h2 % 8;
            o2 = o1 + h2;
            l2 = l1 - h2;
            l1 = h2;
            nleaf = 3;
        }
    }
    const bool second = lane >= 8;
    const double a = np_row_leaves(lane, second ? o1 : 0, second ? l1 : l0, f);
    const double s0 = row_bcast_f64(a, 0), s1 = row_bcast_f64(a, 8);
    double s2 = 0.0;
    // (rows of one wavefront may differ in their leaf count: the third turn is taken by the rows that have one)
    if (nleaf == 3) s2 = row_bcast_f64(np_row_leaves(lane, o2, second ? 0 : l2, f), 0);
    return (nleaf == 1) ? s0 : (nleaf == 2) ? (s0 + s1) : (s0 + (s1 + s2));
}
#endif

// ---------------------------------------------------------------------------------------------
// in-LDS bitonic sort (ascending) of a power-of-two padded array
// ---------------------------------------------------------------------------------------------
TSFA_DEV int next_pow2(int n) {
    int p = 1;
    while (p < n) p <<= 1;
    return p;
}

template <typename K>
TSFA_DEV void blk_bitonic_sort(const Blk &b, K *a, int npow2) {
    for (int k = 2; k <= npow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            blk_sync();
            for (int t = b.tid; t < (npow2 >> 1); t += b.nt) {
                // index of the lower element of pair t for stride j
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const bool up = ((i & k) == 0);
                const K x = a[i], y = a[l];
                if ((x > y) == up) {
                    a[i] = y;
                    a[l] = x;
                }
            }
        }
    }
    blk_sync();
}

// perm[0 .. np2) = indices 0 .. n-1 sorted ascending by key[index] (ties by index), padded with the index type's largest
// value (key +inf).  In-LDS bitonic network on the indices; np2 = power of two >= n.  IDX: unsigned short (n <= 65 535) or
// unsigned int (the long-series build).
template <class IDX>
TSFA_DEV void blk_argsort_idx(const Blk &b, const double *key, int n, IDX *perm, int np2) {
    const IDX none = (IDX)~(IDX)0;
    blk_sync();
    for (int i = b.tid; i < np2; i += b.nt) perm[i] = (i < n) ? (IDX)i : none;
    for (int k = 2; k <= np2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            blk_sync();
            for (int t = b.tid; t < (np2 >> 1); t += b.nt) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const bool up = ((i & k) == 0);
                const IDX a = perm[i], c = perm[l];
                const double ka = (a == none) ? TSFA_INF : key[a], kc = (c == none) ? TSFA_INF : key[c];
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
TSFA_DEV void blk_argsort_u16(const Blk &b, const double *key, int n, unsigned short *perm, int np2) {
    blk_argsort_idx<unsigned short>(b, key, n, perm, np2);
}

#if TSFA_GPU
// ---------------------------------------------------------------------------------------------
// Register-blocked bitonic sort of np2 = E * nt (key, index) pairs, ascending by key, ties by index.
// Thread t holds elements [t*E, (t+1)*E).  Of the log2(np2)(log2(np2)+1)/2 compare-exchange stages
//   * stride < E        : both elements live in this thread's registers,
//   * stride < 64 * E   : the partner lives in another lane of the wavefront -> ds_bpermute (no LDS memory, no barrier),
//   * stride >= 64 * E  : the partner lives in another wavefront -> exchanged through LDS with a barrier pair.
// For 1024 elements on 256 threads that is 19 + 33 + 3 stages: three barrier pairs instead of 55.
// xchg_idx: LDS exchange buffer of np2 indices, only touched by the cross-wavefront stages; the partner's key is
// looked up again from its index (keyof), so no second buffer is needed.
// ---------------------------------------------------------------------------------------------
TSFA_DEV bool sort_pair_gt(double ka, int ia, double kb, int ib) { return (ka > kb) || (ka == kb && ia > ib); }

template <int E, int J>
TSFA_DEV void sort_stage_regs(double (&key)[E], int (&idx)[E], int g0, int k) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if ((e & J) != 0) continue;
        const int f = e | J;
        const bool up = (((g0 + e) & k) == 0);
        const bool gt = sort_pair_gt(key[e], idx[e], key[f], idx[f]);
        if (gt == up) {
            const double tk = key[e]; key[e] = key[f]; key[f] = tk;
            const int ti = idx[e]; idx[e] = idx[f]; idx[f] = ti;
        }
    }
}

// value held by lane (lane ^ LX).  LX = 1, 2: DPP quad_perm (VALU rate, no LDS pipe); LX = 16, 32: the gfx950 row /
// half swaps v_permlane16_swap / v_permlane32_swap; LX = 8: row_ror:8; LX = 4: two bank-masked row shifts.
template <int LX>
TSFA_DEV int lane_xor_i32(int v) {
    if (LX == 1) return __builtin_amdgcn_update_dpp(v, v, TSFA_DPP_QUAD_XOR1, 0xf, 0xf, false);
    if (LX == 2) return __builtin_amdgcn_update_dpp(v, v, TSFA_DPP_QUAD_XOR2, 0xf, 0xf, false);
    if (LX == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return ((threadIdx.x >> 4) & 1) ? (int)r[0] : (int)r[1];  // odd rows got their partner in vdst, even rows in vsrc
    }
    if (LX == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return ((threadIdx.x >> 5) & 1) ? (int)r[0] : (int)r[1];
    }
    if (LX == 8) return __builtin_amdgcn_update_dpp(v, v, 0x128 /* row_ror:8 */, 0xf, 0xf, false);
    if (LX == 4) {  // lanes 0-3 / 8-11 of a row take from lane + 4, lanes 4-7 / 12-15 from lane - 4
        const int t = __builtin_amdgcn_update_dpp(v, v, 0x104 /* row_shl:4 */, 0xf, 0x5, false);
        return __builtin_amdgcn_update_dpp(t, v, 0x114 /* row_shr:4 */, 0xf, 0xa, false);
    }
    return __shfl_xor(v, LX);
}
template <int LX>
TSFA_DEV double lane_xor_f64(double v) {
    union { double d; int i[2]; } a, r;
    a.d = v;
    r.i[0] = lane_xor_i32<LX>(a.i[0]);
    r.i[1] = lane_xor_i32<LX>(a.i[1]);
    return r.d;
}

template <int E, int LX>
TSFA_DEV void sort_stage_lanes(double (&key)[E], int (&idx)[E], int g0, int k, int j) {
    const bool lower = ((g0 & j) == 0);  // this thread holds the lower index of each pair
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const double pk = lane_xor_f64<LX>(key[e]);
        const int pi = lane_xor_i32<LX>(idx[e]);
        const bool up = (((g0 + e) & k) == 0);
        const bool gt = sort_pair_gt(key[e], idx[e], pk, pi);
        // keep the minimum when (lower == up), the maximum otherwise
        const bool take = (lower == up) ? gt : !gt;
        if (take) { key[e] = pk; idx[e] = pi; }
    }
}

template <int E, class KF>
TSFA_DEV void blk_sort_pairs_regs(const Blk &b, double (&key)[E], int (&idx)[E], unsigned short *xchg_idx, KF keyof) {
    const int np2 = E * b.nt;
    const int g0 = b.tid * E;
    for (int k = 2; k <= np2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j < E) {
                switch (j) {
                case 1: if (E > 1) sort_stage_regs<E, (E > 1 ? 1 : 0)>(key, idx, g0, k); break;
                case 2: if (E > 2) sort_stage_regs<E, (E > 2 ? 2 : 0)>(key, idx, g0, k); break;
                case 4: if (E > 4) sort_stage_regs<E, (E > 4 ? 4 : 0)>(key, idx, g0, k); break;
                default: break;
                }
            } else if (j < 64 * E) {
                switch (j / E) {
                case 1: sort_stage_lanes<E, 1>(key, idx, g0, k, j); break;
                case 2: sort_stage_lanes<E, 2>(key, idx, g0, k, j); break;
                case 4: sort_stage_lanes<E, 4>(key, idx, g0, k, j); break;
                case 8: sort_stage_lanes<E, 8>(key, idx, g0, k, j); break;
                case 16: sort_stage_lanes<E, 16>(key, idx, g0, k, j); break;
                default: sort_stage_lanes<E, 32>(key, idx, g0, k, j); break;
                }
            } else {
                blk_sync();
#pragma unroll
                for (int e = 0; e < E; ++e) xchg_idx[g0 + e] = (unsigned short)idx[e];
                blk_sync();
                const bool lower = ((g0 & j) == 0);
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int pi = xchg_idx[(g0 + e) ^ j];
                    const double pk = keyof(pi);
                    const bool up = (((g0 + e) & k) == 0);
                    const bool gt = sort_pair_gt(key[e], idx[e], pk, pi);
                    const bool take = (lower == up) ? gt : !gt;
                    if (take) { key[e] = pk; idx[e] = pi; }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same sort for keys that are exact float32 values: (order-preserving 32-bit image of the key) << 16 | index is
// ONE 64-bit integer, so a compare-exchange is one v_cmp_gt_u64, a scalar XOR with the direction mask and two
// v_cndmask per element -- about a third of the instructions of the (double key, index) version above.
// keyof(i) returns the float key of index i (cross-wavefront stages only exchange indices, as above).
// ---------------------------------------------------------------------------------------------
TSFA_DEV unsigned long long sort_pack_f32(float key, int idx) {
    const unsigned int u = __float_as_uint(key);
    const unsigned int m = u ^ ((unsigned int)((int)u >> 31) | 0x80000000u);  // monotone: negative floats reversed
    return ((unsigned long long)m << 16) | (unsigned long long)(unsigned int)idx;
}
template <int LX>
TSFA_DEV unsigned long long lane_xor_u64(unsigned long long v) {
    const int lo = lane_xor_i32<LX>((int)(unsigned int)v), hi = lane_xor_i32<LX>((int)(unsigned int)(v >> 32));
    return ((unsigned long long)(unsigned int)hi << 32) | (unsigned long long)(unsigned int)lo;
}
template <int E, int J>
TSFA_DEV void sortp_stage_regs(unsigned long long (&pk)[E], int g0, int k) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if ((e & J) != 0) continue;
        const int f = e | J;
        const bool up = (((g0 + e) & k) == 0);
        const bool gt = pk[e] > pk[f];
        const unsigned long long lo = gt ? pk[f] : pk[e], hi = gt ? pk[e] : pk[f];
        pk[e] = up ? lo : hi;
        pk[f] = up ? hi : lo;
    }
}
template <int E, int LX>
TSFA_DEV void sortp_stage_lanes(unsigned long long (&pk)[E], int g0, int k, int j) {
    const bool lower = ((g0 & j) == 0);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const unsigned long long o = lane_xor_u64<LX>(pk[e]);
        const bool up = (((g0 + e) & k) == 0);
        const bool take = ((pk[e] > o) == (lower == up));  // keep the minimum when lower == up, else the maximum
        pk[e] = take ? o : pk[e];
    }
}
template <int E, class KF>
TSFA_DEV void blk_sort_packed_regs(const Blk &b, unsigned long long (&pk)[E], unsigned short *xchg_idx, KF keyof) {
    const int np2 = E * b.nt;
    const int g0 = b.tid * E;
    for (int k = 2; k <= np2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j < E) {
                switch (j) {
                case 1: if (E > 1) sortp_stage_regs<E, (E > 1 ? 1 : 0)>(pk, g0, k); break;
                case 2: if (E > 2) sortp_stage_regs<E, (E > 2 ? 2 : 0)>(pk, g0, k); break;
                default: break;
                }
            } else if (j < 64 * E) {
                switch (j / E) {
                case 1: sortp_stage_lanes<E, 1>(pk, g0, k, j); break;
                case 2: sortp_stage_lanes<E, 2>(pk, g0, k, j); break;
                case 4: sortp_stage_lanes<E, 4>(pk, g0, k, j); break;
                case 8: sortp_stage_lanes<E, 8>(pk, g0, k, j); break;
                case 16: sortp_stage_lanes<E, 16>(pk, g0, k, j); break;
                default: sortp_stage_lanes<E, 32>(pk, g0, k, j); break;
                }
            } else {
                blk_sync();
#pragma unroll
                for (int e = 0; e < E; ++e) xchg_idx[g0 + e] = (unsigned short)(pk[e] & 0xFFFFull);
                blk_sync();
                const bool lower = ((g0 & j) == 0);
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int pi = xchg_idx[(g0 + e) ^ j];
                    const unsigned long long o = sort_pack_f32(keyof(pi), pi);
                    const bool up = (((g0 + e) & k) == 0);
                    const bool take = ((pk[e] > o) == (lower == up));
                    pk[e] = take ? o : pk[e];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Keys only (float or double), E = np2 / nt of them per thread: a compare-exchange is v_min + v_max (+ a select on
// the direction); only the strides >= 64 * E go through LDS (xchg, np2 keys -- the destination array itself).
// For 1024 float keys on 128 threads: 27 in-register + 27 cross-lane stages and ONE barrier pair instead of 55.
// ---------------------------------------------------------------------------------------------
TSFA_DEV float sort_min(float a, float b) { return __builtin_fminf(a, b); }
TSFA_DEV float sort_max(float a, float b) { return __builtin_fmaxf(a, b); }
TSFA_DEV double sort_min(double a, double b) { return __builtin_fmin(a, b); }
TSFA_DEV double sort_max(double a, double b) { return __builtin_fmax(a, b); }
template <int LX>
TSFA_DEV float lane_xor_key(float v) { return __int_as_float(lane_xor_i32<LX>(__float_as_int(v))); }
template <int LX>
TSFA_DEV double lane_xor_key(double v) { return lane_xor_f64<LX>(v); }

template <int E, int J, typename K>
TSFA_DEV void sortk_stage_regs(K (&key)[E], int g0, int k) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if ((e & J) != 0) continue;
        const int f = e | J;
        const bool up = (((g0 + e) & k) == 0);
        const K lo = sort_min(key[e], key[f]), hi = sort_max(key[e], key[f]);
        key[e] = up ? lo : hi;
        key[f] = up ? hi : lo;
    }
}
template <int E, int LX, typename K>
TSFA_DEV void sortk_stage_lanes(K (&key)[E], int g0, int k, int j) {
    const bool lower = ((g0 & j) == 0);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const K o = lane_xor_key<LX>(key[e]);
        const bool up = (((g0 + e) & k) == 0);
        const K lo = sort_min(key[e], o), hi = sort_max(key[e], o);
        key[e] = (lower == up) ? lo : hi;
    }
}
template <int E, typename K>
TSFA_DEV void blk_sort_keys_regs(const Blk &b, K (&key)[E], K *xchg) {
    const int np2 = E * b.nt;
    const int g0 = b.tid * E;
    for (int k = 2; k <= np2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j < E) {
                switch (j) {
                case 1: if (E > 1) sortk_stage_regs<E, (E > 1 ? 1 : 0)>(key, g0, k); break;
                case 2: if (E > 2) sortk_stage_regs<E, (E > 2 ? 2 : 0)>(key, g0, k); break;
                case 4: if (E > 4) sortk_stage_regs<E, (E > 4 ? 4 : 0)>(key, g0, k); break;
                case 8: if (E > 8) sortk_stage_regs<E, (E > 8 ? 8 : 0)>(key, g0, k); break;
                default: break;
                }
            } else if (j < 64 * E) {
                switch (j / E) {
                case 1: sortk_stage_lanes<E, 1>(key, g0, k, j); break;
                case 2: sortk_stage_lanes<E, 2>(key, g0, k, j); break;
                case 4: sortk_stage_lanes<E, 4>(key, g0, k, j); break;
                case 8: sortk_stage_lanes<E, 8>(key, g0, k, j); break;
                case 16: sortk_stage_lanes<E, 16>(key, g0, k, j); break;
                default: sortk_stage_lanes<E, 32>(key, g0, k, j); break;
                }
            } else {
                blk_sync();
#pragma unroll
                for (int e = 0; e < E; ++e) xchg[g0 + e] = key[e];
                blk_sync();
                const bool lower = ((g0 & j) == 0);
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const K o = xchg[(g0 + e) ^ j];
                    const bool up = (((g0 + e) & k) == 0);
                    const K lo = sort_min(key[e], o), hi = sort_max(key[e], o);
                    key[e] = (lower == up) ? lo : hi;
                }
            }
        }
    }
}
// dst[0 .. np2) = src[0 .. n) sorted ascending, padded with +inf; false if the shape has no register-blocked variant
template <int E, typename K>
TSFA_DEVN void blk_sorted_copy_regs_e(const Blk &b, const K *src, int n, K *dst) {
    K key[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int g = b.tid * E + e;
        key[e] = (g < n) ? src[g] : (K)TSFA_INF;
    }
    blk_sort_keys_regs<E>(b, key, dst);
    blk_sync();
#pragma unroll
    for (int e = 0; e < E; ++e) dst[b.tid * E + e] = key[e];
    blk_sync();
}
template <typename K>
TSFA_DEV bool blk_sorted_copy_regs(const Blk &b, const K *src, int n, K *dst, int np2) {
    if (np2 == 2 * b.nt) { blk_sorted_copy_regs_e<2>(b, src, n, dst); return true; }
    if (np2 == 4 * b.nt) { blk_sorted_copy_regs_e<4>(b, src, n, dst); return true; }
    if (np2 == 8 * b.nt) { blk_sorted_copy_regs_e<8>(b, src, n, dst); return true; }
    if (np2 == 16 * b.nt) { blk_sorted_copy_regs_e<16>(b, src, n, dst); return true; }
    return false;
}
#endif

// ---------------------------------------------------------------------------------------------
// special functions
// ---------------------------------------------------------------------------------------------
// sin(pi x), cos(pi x) for x in [0, 2): exact octant reduction, so table entries are symmetric
TSFA_DEV void tsfa_sincospi(double x, double *s, double *c) {
#if TSFA_GPU
    sincospi(x, s, c);
#else
    // reduce to r in [-0.25, 0.25] around the nearest multiple of 0.5
    const double q = floor(x * 2.0 + 0.5);
    const double r = x - q * 0.5;
    const int qi = ((int)q) & 3;
    const double sr = sin(M_PI * r), cr = cos(M_PI * r);
    switch (qi) {
    case 0: *s = sr; *c = cr; break;
    case 1: *s = cr; *c = -sr; break;
    case 2: *s = -sr; *c = -cr; break;
    default: *s = -cr; *c = sr; break;
    }
#endif
}

// log-gamma (Lanczos, g = 7, n = 9), |rel err| ~ 1e-15 for x > 0
TSFA_DEV double tsfa_lgamma(double x) {
    const double c[9] = {0.99999999999980993,  676.5203681218851,     -1259.1392167224028,
                         771.32342877765313,   -176.61502916214059,   12.507343278686905,
                         -0.13857109526572012, 9.9843695780195716e-6, 1.5056327351493116e-7};
    if (x < 0.5) return log(M_PI / fabs(sin(M_PI * x))) - tsfa_lgamma(1.0 - x);
    x -= 1.0;
    double a = c[0];
    const double t = x + 7.5;
    for (int i = 1; i < 9; ++i) a += c[i] / (x + (double)i);
    return 0.5 * log(2.0 * M_PI) + (x + 0.5) * log(t) - t + log(a);
}

// continued fraction for the regularized incomplete beta (modified Lentz)
TSFA_DEV double tsfa_betacf(double a, double b, double x) {
    const double FPMIN = 1e-300;
    const double qab = a + b, qap = a + 1.0, qam = a - 1.0;
    double c = 1.0, d = 1.0 - qab * x / qap;
    if (fabs(d) < FPMIN) d = FPMIN;
    d = 1.0 / d;
    double h = d;
    for (int m = 1; m <= 10000; ++m) {
        const double m2 = 2.0 * m;
        double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
        d = 1.0 + aa * d;
        if (fabs(d) < FPMIN) d = FPMIN;
        c = 1.0 + aa / c;
        if (fabs(c) < FPMIN) c = FPMIN;
        d = 1.0 / d;
        h *= d * c;
        aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
        d = 1.0 + aa * d;
        if (fabs(d) < FPMIN) d = FPMIN;
        c = 1.0 + aa / c;
        if (fabs(c) < FPMIN) c = FPMIN;
        d = 1.0 / d;
        const double del = d * c;
        h *= del;
        if (fabs(del - 1.0) < 1e-16) break;
    }
    return h;
}

// regularized incomplete beta I_x(a, b)
TSFA_DEV double tsfa_betainc(double a, double b, double x) {
    if (x != x) return TSFA_NAN;
    if (x <= 0.0) return 0.0;
    if (x >= 1.0) return 1.0;
    const double lbt = tsfa_lgamma(a + b) - tsfa_lgamma(a) - tsfa_lgamma(b) + a * log(x) + b * log1p(-x);
    if (x < (a + 1.0) / (a + b + 2.0)) return exp(lbt) * tsfa_betacf(a, b, x) / a;
    return 1.0 - exp(lbt) * tsfa_betacf(b, a, 1.0 - x) / b;
}

// two-sided p-value of a Student-t statistic (scipy.stats.linregress: 2 * stdtr(df, -|t|))
TSFA_DEV double tsfa_t_pvalue2(double t, double df) {
    if (t != t || !(df > 0.0)) return TSFA_NAN;
    if (isinf(t)) return 0.0;
    return tsfa_betainc(0.5 * df, 0.5, df / (df + t * t));
}

// standard normal CDF
TSFA_DEV double tsfa_norm_cdf(double x) { return 0.5 * erfc(-x * 0.70710678118654752440); }

// ---------------------------------------------------------------------------------------------
// scipy.stats.linregress(range(m), y) for y in LDS (or computed by G(i)); every thread gets the result
// out[0..4] = pvalue, rvalue, intercept, slope, stderr       (scipy/stats/_stats_py.py, linregress)
// ---------------------------------------------------------------------------------------------
// want_p: the p-value costs a continued fraction (incomplete beta); agg_linear_trend never asks for it
template <class BT, class G>
TSFA_DEV void blk_linregress_index(const BT &b, int m, G g, double *out5, bool want_p = true) {
    const double dm = (double)m;
    double sy = 0.0;
    for (int i = b.tid; i < m; i += b.nt) sy += g(i);
    sy = blk_sum(b, sy);
    const double ymean = sy / dm;
    const double xmean = (dm - 1.0) * 0.5;
    double sxy = 0.0, syy = 0.0;
    for (int i = b.tid; i < m; i += b.nt) {
        const double dy = g(i) - ymean;
        const double dx = (double)i - xmean;
        sxy += dx * dy;
        syy += dy * dy;
    }
    sxy = blk_sum(b, sxy);
    syy = blk_sum(b, syy);
    // sum_i (i - xmean)^2 = m (m^2 - 1) / 12
    const double ssxm = (dm * (dm * dm - 1.0) / 12.0) / dm;
    const double ssxym = sxy / dm;
    const double ssym = syy / dm;
    double r;
    if (ssxm == 0.0 || ssym == 0.0) {
        r = 0.0;
    } else {
        r = ssxym / sqrt(ssxm * ssym);
        if (r > 1.0) r = 1.0;
        else if (r < -1.0) r = -1.0;
    }
    const double slope = ssxym / ssxm;
    const double intercept = ymean - slope * xmean;
    double prob, stderr_;
    if (m == 2) {
        prob = (g(0) == g(1)) ? 1.0 : 0.0;
        stderr_ = 0.0;
    } else {
        const double df = dm - 2.0;
        const double TINY = 1.0e-20;
        const double t = r * sqrt(df / ((1.0 - r + TINY) * (1.0 + r + TINY)));
        prob = want_p ? tsfa_t_pvalue2(t, df) : TSFA_NAN;
        stderr_ = sqrt((1.0 - r * r) * ssym / ssxm / df);
    }
    out5[TSFA_ATTR_PVALUE] = prob;
    out5[TSFA_ATTR_RVALUE] = r;
    out5[TSFA_ATTR_INTERCEPT] = intercept;
    out5[TSFA_ATTR_SLOPE] = slope;
    out5[TSFA_ATTR_STDERR] = stderr_;
}

// scipy.stats.linregress(x, y) for an arbitrary abscissa (fc.py:2274 linear_trend_timewise: x = hours since the
// first timestamp).  Centred sums as np.cov(x, y, bias=1) forms them.  All x identical: scipy raises ValueError
// ("Cannot calculate a linear regression if all x values are identical"); here every attribute is NaN.
template <class BT, class GX, class GY>
TSFA_DEV void blk_linregress_xy(const BT &b, int m, GX gx, GY gy, double *out5) {
    const double dm = (double)m;
    double sx = 0.0, sy = 0.0;
    for (int i = b.tid; i < m; i += b.nt) { sx += gx(i); sy += gy(i); }
    sx = blk_sum(b, sx);
    sy = blk_sum(b, sy);
    const double xmean = sx / dm, ymean = sy / dm;
    double sxx = 0.0, sxy = 0.0, syy = 0.0;
    for (int i = b.tid; i < m; i += b.nt) {
        const double dx = gx(i) - xmean, dy = gy(i) - ymean;
        sxx += dx * dx;
        sxy += dx * dy;
        syy += dy * dy;
    }
    sxx = blk_sum(b, sxx);
    sxy = blk_sum(b, sxy);
    syy = blk_sum(b, syy);
    const double ssxm = sxx / dm, ssxym = sxy / dm, ssym = syy / dm;
    if (ssxm == 0.0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) out5[k] = TSFA_NAN;
        return;
    }
    double r;
    if (ssym == 0.0) {
        r = 0.0;
    } else {
        r = ssxym / sqrt(ssxm * ssym);
        if (r > 1.0) r = 1.0;
        else if (r < -1.0) r = -1.0;
    }
    const double slope = ssxym / ssxm;
    const double intercept = ymean - slope * xmean;
    double prob, stderr_;
    if (m == 2) {
        prob = (gy(0) == gy(1)) ? 1.0 : 0.0;
        stderr_ = 0.0;
    } else {
        const double df = dm - 2.0;
        const double TINY = 1.0e-20;
        const double t = r * sqrt(df / ((1.0 - r + TINY) * (1.0 + r + TINY)));
        prob = tsfa_t_pvalue2(t, df);
        stderr_ = sqrt((1.0 - r * r) * ssym / ssxm / df);
    }
    out5[TSFA_ATTR_PVALUE] = prob;
    out5[TSFA_ATTR_RVALUE] = r;
    out5[TSFA_ATTR_INTERCEPT] = intercept;
    out5[TSFA_ATTR_SLOPE] = slope;
    out5[TSFA_ATTR_STDERR] = stderr_;
}

// ---------------------------------------------------------------------------------------------
// np.histogram(v, bins) with uniform bins over [vmin, vmax] followed by the entropy of the bin
// probabilities (feature_calculators.py:1666 binned_entropy).  `cnt` = LDS int array of >= bins ints.
// G(i) returns element i of the histogrammed vector of length m.
// numpy/lib/_histograms_impl.py: uniform-bin fast path incl. the edge corrections.
// ---------------------------------------------------------------------------------------------
TSFA_DEV double np_linspace_at(double start, double stop, int num_edges, int i) {
    // np.linspace(start, stop, num_edges)[i]
    const int div = num_edges - 1;
    if (i == div && num_edges > 1) return stop;
    const double delta = stop - start;
    const double step = delta / (double)div;
    if (step == 0.0) return ((double)i / (double)div) * delta + start;
    return (double)i * step + start;
}

template <bool RECIP, class BT, class G>
TSFA_DEV void binned_scatter(const BT &b, int m, G g, int bins, double first, double last, double norm, double inv_norm, double step,
                             bool flat, int c0, int cb, int *cnt) {
    for (int i = b.tid; i < m; i += b.nt) {
        const double v = g(i);
        if (!(v >= first && v <= last)) continue;
        const double fidx = RECIP ? ((v - first) * inv_norm) * (double)bins : ((v - first) / norm) * (double)bins;
        int idx = (fidx < 2147483000.0) ? (int)fidx : (bins - 1);
        if (idx >= bins) idx = bins - 1;
        if (idx < 0) idx = 0;
        const double e0 = flat ? np_linspace_at(first, last, bins + 1, idx) : ((idx == bins) ? last : (double)idx * step + first);
        if (v < e0) idx -= 1;
        if (idx != bins - 1) {
            const double e1 = flat ? np_linspace_at(first, last, bins + 1, idx + 1)
                                   : ((idx + 1 == bins) ? last : (double)(idx + 1) * step + first);
            if (v >= e1) idx += 1;
        }
        if (idx < c0 || idx >= c0 + cb) continue;
#if TSFA_GPU
        atomicAdd(&cnt[idx - c0], 1);
#else
        cnt[idx - c0] += 1;
#endif
    }
}

template <class BT, class G>
TSFA_DEV double blk_binned_entropy(const BT &b, int m, G g, int bins, double vmin, double vmax, int *cnt, int cap = 256) {
    double first = vmin, last = vmax;
    // np.histogram raises "autodetected range of [..] is not finite" for a series holding +-inf (fc.py:1691); the host
    // turns the NaN of this cell into that ValueError (feature_extraction/reference_errors.py)
    if (!(fabs(first) < TSFA_INF) || !(fabs(last) < TSFA_INF)) return TSFA_NAN;
    if (first == last) {
        first = first - 0.5;
        last = last + 0.5;
    }
    const double norm = last - first;
    // numpy: idx = int((v - first) / norm * bins), then one step down if v < edge[idx], one step up if v >= edge[idx + 1]:
    // the two corrections make idx the bin whose edges enclose v for ANY estimate within one bin of it, so the estimate
    // may use the reciprocal (one division per call instead of one per element), and the edges np.linspace(first, last,
    // bins + 1)[i] = i * step + first with the step formed once (numpy's own expression; three divisions per element
    // before -- a float64 division is ~30 instructions)
    const double inv_norm = 1.0 / norm;
    // ... unless the reciprocal is not a float64 worth multiplying by: a range of subnormal width has 1 / norm = inf
    // (every sample landed in bin 0), a range beyond 2^1022 a subnormal reciprocal -- numpy's own quotient then
    const bool recip = (inv_norm < TSFA_INF) && (inv_norm >= 2.2250738585072014e-308);
    const double delta = last - first, step = delta / (double)bins;
    const bool flat = (step == 0.0);
    {
        // numpy >= 2.0 (_histograms_impl.py:452): "Too many bins for data range. Cannot create N finite-sized bins." when two
        // neighbouring edges of the linspace coincide (a range of a few ulps: 2^53 + {0, 2, 4}); the host turns the NaN
        // into that ValueError (reference_errors.py)
        // (two edges k step + first can only round to the same float64 where the step is within a few ulps of the larger
        //  end of the range: everywhere else the test -- a sweep over the bins and a workgroup OR -- is skipped)
        const double amax = fmax(fabs(first), fabs(last));
        if (!(step > 8.9e-16 * amax)) {
            unsigned stuck = 0;
            for (int k = b.tid; k < bins; k += b.nt)
                if (np_linspace_at(first, last, bins + 1, k) >= np_linspace_at(first, last, bins + 1, k + 1)) stuck = 1;
            if (blk_or16(b, stuck) != 0) return TSFA_NAN;
        }
    }
    // The counters hold `cap` bins (LDS); more bins than that (the reference takes any max_bins: a from_columns() settings
    // object may ask for 1000) are counted in rounds of `cap`, each a sweep over the samples that keeps the ones of its bins.
    // Up to `cap` bins: one round, the sums below in the order they always had.
    double e = 0.0;
    for (int c0 = 0; c0 < bins; c0 += cap) {
        const int cb = (bins - c0 < cap) ? (bins - c0) : cap;
        blk_sync();
        for (int k = b.tid; k < cb; k += b.nt) cnt[k] = 0;
        blk_sync();
        // (two loops: under one loop with a select the compiler evaluates the quotient for every sample -- a float64 division
        //  is ~30 instructions -- although `recip` is the same for the whole series)
        if (recip) binned_scatter<true>(b, m, g, bins, first, last, norm, inv_norm, step, flat, c0, cb, cnt);
        else binned_scatter<false>(b, m, g, bins, first, last, norm, inv_norm, step, flat, c0, cb, cnt);
        blk_sync();
        for (int k = b.tid; k < cb; k += b.nt) {
            const int c = cnt[k];
            if (c > 0) {
                const double p = (double)c / (double)m;
                e += p * log(p);
            }
        }
    }
    e = blk_sum(b, e);
    return -e;
}

#endif  // TSFA_COMMON_H
