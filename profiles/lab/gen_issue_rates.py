#!/usr/bin/env python
"""Writes profiles/lab/issue_rates.hip: one kernel per instruction, a loop of 64 independent instructions (eight chains x
eight rounds), W = 1, 2, 4, 8 wavefronts per SIMD.  The table below is the instruction mix of the family kernels
(llvm-objdump histogram of libtsfresh_amd's code object: profiles/lab/valu_mix.py).

    python profiles/lab/gen_issue_rates.py && hipcc --offload-arch=gfx950 -O2 profiles/lab/issue_rates.hip -o profiles/lab/build/issue_rates

Operands: {d} chain register (read-modify-write), {n} the NEXT chain's register, {c} a per-lane constant, {a} an LDS byte
address, {k} chain number.  width 32: chains are 32-bit VGPRs; 64: VGPR pairs (doubles).
"""
import os

T = []   # (name, class, width, asm, per-instruction extra count)


def op(name, klass, width, asm):
    T.append((name, klass, width, asm))


# ---- 32-bit integer / bit ops
for o in ("v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32",
          "v_min_i32", "v_max_u32", "v_mul_u32_u24", "v_add_f32", "v_mul_f32", "v_min_f32", "v_max_f32"):
    op(o[2:], "vop2", 32, o + " {d}, {d}, {c}")
op("mov_b32", "vop1", 32, "v_mov_b32 {d}, {n}")
op("mov_b64", "vop1", 64, "v_mov_b64 {d}, {n}")
op("not_b32", "vop1", 32, "v_not_b32 {d}, {d}")
for o in ("v_alignbit_b32", "v_bfe_u32", "v_lshl_add_u32", "v_add3_u32", "v_lshl_or_b32", "v_and_or_b32", "v_mad_u32_u24",
          "v_fma_f32", "v_perm_b32", "v_med3_f32", "v_min3_f32"):
    op(o[2:], "vop3 3-src", 32, o + " {d}, {d}, {c}, {n}")
op("bitop3_b32", "vop3 3-src", 32, "v_bitop3_b32 {d}, {d}, {c}, {n} bitop3:0x96")
op("bcnt_u32_b32", "vop3", 32, "v_bcnt_u32_b32 {d}, {d}, {c}")
op("mul_lo_u32", "vop3", 32, "v_mul_lo_u32 {d}, {d}, {c}")
op("mul_hi_u32", "vop3", 32, "v_mul_hi_u32 {d}, {d}, {c}")
op("mbcnt_lo", "vop3", 32, "v_mbcnt_lo_u32_b32 {d}, {d}, {c}")
op("fmac_f32", "vop2", 32, "v_fmac_f32 {d}, {c}, {c}")
op("cndmask_vcc", "vop2 reads vcc", 32, "v_cndmask_b32 {d}, {d}, {c}, vcc")
op("cndmask_sgpr", "vop3 reads s[20:21]", 32, "v_cndmask_b32 {d}, {d}, {c}, s[20:21]")
op("addc_co_u32", "vop2 vcc in/out", 32, "v_addc_co_u32 {d}, vcc, {d}, {c}, vcc")
op("add_co_u32", "vop2 vcc out", 32, "v_add_co_u32 {d}, vcc, {d}, {c}")
for o in ("v_cmp_lt_u32", "v_cmp_eq_u32", "v_cmp_gt_i32", "v_cmp_lt_f32"):
    op(o[2:] + "_vcc", "vopc", 32, o + " vcc, {d}, {c}")
op("cmp_lt_i32_sgpr", "vopc e64", 32, "v_cmp_lt_i32 s[20:21], {d}, {c}")
op("cmp_then_cndmask", "pair: v_cmp vcc + v_cndmask vcc (counted as 2)", 32, "v_cmp_lt_u32 vcc, {d}, {c}\n\tv_cndmask_b32 {d}, {d}, {c}, vcc")
# ---- cross-lane
DPPS = {"quad_perm": "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "row_shr1": "row_shr:1 row_mask:0xf bank_mask:0xf",
        "row_mirror": "row_mirror row_mask:0xf bank_mask:0xf", "wave_shl1": "wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0",
        "wave_shr1": "wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0", "row_bcast15": "row_bcast:15 row_mask:0xa bank_mask:0xf",
        "row_bcast31": "row_bcast:31 row_mask:0xc bank_mask:0xf"}
for k, v in DPPS.items():
    op("mov_b32_dpp_" + k, "dpp", 32, "v_mov_b32_dpp {d}, {n} " + v)
op("and_b32_dpp_wave_shl1", "dpp", 32, "v_and_b32_dpp {d}, {d}, {c} " + DPPS["wave_shl1"])
op("and_b32_dpp_row_shr1", "dpp", 32, "v_and_b32_dpp {d}, {d}, {c} " + DPPS["row_shr1"])
op("or_b32_dpp_row_shr1", "dpp", 32, "v_or_b32_dpp {d}, {d}, {c} " + DPPS["row_shr1"])
op("add_u32_dpp_quad", "dpp", 32, "v_add_u32_dpp {d}, {d}, {c} " + DPPS["quad_perm"])
op("add_f32_dpp_row_shr1", "dpp", 32, "v_add_f32_dpp {d}, {d}, {c} " + DPPS["row_shr1"])
op("permlane32_swap", "vop1 swap", 32, "v_permlane32_swap_b32 {d}, {n}")
op("permlane16_swap", "vop1 swap", 32, "v_permlane16_swap_b32 {d}, {n}")
op("readlane", "v_readlane -> sgpr", 32, "v_readlane_b32 s2{k}, {d}, 5")
op("readfirstlane", "v_readfirstlane -> sgpr", 32, "v_readfirstlane_b32 s2{k}, {d}")
op("writelane", "v_writelane <- sgpr", 32, "v_writelane_b32 {d}, s20, 5")
op("ds_bpermute_b32", "lds crossbar", 32, "ds_bpermute_b32 {d}, {a}, {n}")
op("ds_swizzle_b32", "lds crossbar", 32, "ds_swizzle_b32 {d}, {n} offset:swizzle(SWAP,1)")
# ---- conversions / transcendental
for o in ("v_cvt_f32_u32", "v_cvt_u32_f32", "v_cvt_f32_i32", "v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32",
          "v_rcp_iflag_f32", "v_floor_f32", "v_rndne_f32"):
    op(o[2:], "vop1", 32, o + " {d}, {d}")
# ---- float64
for o in ("v_add_f64", "v_mul_f64", "v_min_f64", "v_max_f64", "v_ldexp_f64"):
    op(o[2:], "f64", 64, o + " {d}, {d}, {c}" if o != "v_ldexp_f64" else "v_ldexp_f64 {d}, {d}, 1")
op("fma_f64", "f64 3-src", 64, "v_fma_f64 {d}, {d}, {c}, {n}")
op("fmac_f64", "f64 vop2", 64, "v_fmac_f64 {d}, {c}, {c}")
op("div_scale_f64", "f64", 64, "v_div_scale_f64 {d}, vcc, {d}, {c}, {d}")
op("div_fmas_f64", "f64 reads vcc", 64, "v_div_fmas_f64 {d}, {d}, {c}, {n}")
op("div_fixup_f64", "f64", 64, "v_div_fixup_f64 {d}, {d}, {c}, {n}")
for o in ("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_rndne_f64", "v_floor_f64", "v_fract_f64"):
    op(o[2:], "f64 vop1", 64, o + " {d}, {d}")
op("frexp_exp_i32_f64", "f64 vop1", 64, "v_frexp_exp_i32_f64 {lo}, {d}")
op("frexp_mant_f64", "f64 vop1", 64, "v_frexp_mant_f64 {d}, {d}")
for o in ("v_cmp_le_f64", "v_cmp_lt_f64", "v_cmp_class_f64"):
    op(o[2:] + "_vcc", "f64 vopc", 64, o + " vcc, {d}, {c}" if "class" not in o else o + " vcc, {d}, {lo}")
op("cvt_f64_f32", "cvt", 64, "v_cvt_f64_f32 {d}, {lo}")
op("cvt_f32_f64", "cvt", 64, "v_cvt_f32_f64 {lo}, {d}")
op("cvt_f64_u32", "cvt", 64, "v_cvt_f64_u32 {d}, {lo}")
op("cvt_f64_i32", "cvt", 64, "v_cvt_f64_i32 {d}, {lo}")
op("cvt_i32_f64", "cvt", 64, "v_cvt_i32_f64 {lo}, {d}")
op("lshlrev_b64", "64-bit int", 64, "v_lshlrev_b64 {d}, 3, {d}")
op("lshl_add_u64", "64-bit int", 64, "v_lshl_add_u64 {d}, {d}, 3, {c}")
op("mad_u64_u32", "64-bit int", 64, "v_mad_u64_u32 {d}, vcc, {lo}, {lo}, {d}")
op("pk_add_f32", "packed", 64, "v_pk_add_f32 {d}, {d}, {c}")
op("pk_fma_f32", "packed", 64, "v_pk_fma_f32 {d}, {d}, {c}, {n}")
# ---- LDS (conflict-free addresses: lane x access width)
op("ds_read_b32", "lds", 32, "ds_read_b32 {d}, {a4} offset:{off}")
op("ds_read_b64", "lds", 64, "ds_read_b64 {d}, {a8} offset:{off}")
op("ds_read_u16", "lds", 32, "ds_read_u16 {d}, {a4} offset:{off}")
op("ds_write_b32", "lds", 32, "ds_write_b32 {a4}, {d} offset:{off}")
op("ds_write_b64", "lds", 64, "ds_write_b64 {a8}, {d} offset:{off}")
op("ds_add_u32", "lds atomic (conflict-free)", 32, "ds_add_u32 {a4}, {d} offset:{off}")
# ---- mixes
op("mix_valu_salu", "mix: 1 v_and_b32 + 1 s_add_u32 (counted as 1)", 32, "v_and_b32 {d}, {d}, {c}\n\ts_add_u32 s2{k}, s2{k}, 1")
op("mix_valu_2salu", "mix: 1 v_and_b32 + 2 s_add_u32 (counted as 1)", 32, "v_and_b32 {d}, {d}, {c}\n\ts_add_u32 s2{k}, s2{k}, 1\n\ts_and_b32 s3{k}, s3{k}, s20")
op("mix_f64_salu", "mix: 1 v_add_f64 + 1 s_add_u32 (counted as 1)", 64, "v_add_f64 {d}, {d}, {c}\n\ts_add_u32 s2{k}, s2{k}, 1")
op("mix_and_ds_read", "mix: 1 v_and_b32 + 1 ds_read_b32 of another chain (counted as 1)", 32, "v_and_b32 {d}, {d}, {c}\n\tds_read_b32 {n}, {a4} offset:{off}")
op("mix_and_dpp_bcnt", "mix: v_and_b32_dpp + v_bcnt (counted as 2)", 32, "v_and_b32_dpp {d}, {d}, {c} " + DPPS["wave_shl1"] + "\n\tv_bcnt_u32_b32 {n}, {d}, {n}")
op("mix_full_half", "mix: v_and_b32 + v_bcnt (counted as 2)", 32, "v_and_b32 {d}, {d}, {c}\n\tv_bcnt_u32_b32 {n}, {n}, {c}")
op("salu_only", "64 s_add_u32", 32, "s_add_u32 s2{k}, s2{k}, 1")

HEAD = r'''// GENERATED by profiles/lab/gen_issue_rates.py -- do not edit.
// Issue rates of the instructions the family kernels are made of, measured on the box (VERDICT r3 "Next" #3).  Every
// kernel is a loop of 64 INDEPENDENT instructions of one kind (eight chains x eight rounds), run by W = 1, 2, 4, 8
// wavefronts per SIMD on every SIMD (workgroups of 256 threads = one wavefront per SIMD; the LDS request caps the
// workgroups per CU at W; the grid is 8 resident sets, so the chip is in steady state).  Per line: cycles per
// wave-instruction per SIMD = wall time x shader clock / (instructions per wavefront x wavefronts per SIMD); the shader
// clock is MEASURED in the same launch (s_memtime ticks per s_memrealtime tick x 100 MHz), because it moves with the
// instruction mix.  A diagnostic program: nothing in the package builds or loads it.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Tick { unsigned long long shader, real; };
'''

K32 = r'''
__global__ void __launch_bounds__(256) k_%(name)s(unsigned *out, int iters, Tick *ticks) {
    extern __shared__ unsigned lds[];
    unsigned r0 = threadIdx.x + 1, r1 = r0 * 3 + 1, r2 = r0 * 5 + 2, r3 = r0 * 7 + 3, r4 = r0 + 11, r5 = r0 ^ 0x55, r6 = r0 | 0x100, r7 = ~r0;
    const unsigned c = 0x3f800000u | (threadIdx.x & 15), a = ((threadIdx.x & 63) ^ 1) * 4, a4 = (threadIdx.x & 63) * 4, a8 = (threadIdx.x & 63) * 8;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
    __syncthreads();
    asm volatile("s_mov_b64 s[20:21], exec\n\ts_mov_b32 s22, 0\n\ts_mov_b32 s23, 0\n\ts_mov_b32 s24, 0\n\ts_mov_b32 s25, 0\n\ts_mov_b32 s26, 0\n\ts_mov_b32 s27, 0"
                 ::: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), q0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile(%(body)s
                     : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                     : "v"(c), "v"(a), "v"(a4), "v"(a8)
                     : "vcc", "memory", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s30", "s31", "s32", "s33", "s34", "s35", "s36", "s37");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), q1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = Tick{t1 - t0, q1 - q0};
    out[blockIdx.x * 256 + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
}
'''

K64 = r'''
__global__ void __launch_bounds__(256) k_%(name)s(unsigned *out, int iters, Tick *ticks) {
    extern __shared__ unsigned lds[];
    double r0 = threadIdx.x * 1e-3 + 1.0, r1 = r0 * 1.01, r2 = r0 * 1.02, r3 = r0 * 1.03, r4 = r0 * 1.04, r5 = r0 * 1.05, r6 = r0 * 1.06, r7 = r0 * 1.07;
    const double c = 1.0 + 1e-9 * threadIdx.x;
    unsigned w0 = threadIdx.x + 1, w1 = w0 + 1, w2 = w0 + 2, w3 = w0 + 3, w4 = w0 + 4, w5 = w0 + 5, w6 = w0 + 6, w7 = w0 + 7;   // 32-bit sides of the conversions
    const unsigned a = ((threadIdx.x & 63) ^ 1) * 4, a4 = (threadIdx.x & 63) * 4, a8 = (threadIdx.x & 63) * 8;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
    __syncthreads();
    asm volatile("s_mov_b64 s[20:21], exec\n\ts_mov_b32 s22, 0\n\ts_mov_b32 s23, 0\n\ts_mov_b32 s24, 0\n\ts_mov_b32 s25, 0\n\ts_mov_b32 s26, 0\n\ts_mov_b32 s27, 0"
                 ::: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), q0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile(%(body)s
                     : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7),
                       "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7)
                     : "v"(c), "v"(a), "v"(a4), "v"(a8)
                     : "vcc", "memory", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s30", "s31", "s32", "s33", "s34", "s35", "s36", "s37");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), q1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = Tick{t1 - t0, q1 - q0};
    out[blockIdx.x * 256 + threadIdx.x] = (unsigned)(r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7) ^ w0 ^ w1 ^ w2 ^ w3 ^ w4 ^ w5 ^ w6 ^ w7;
}
'''

MAIN = r'''
typedef void (*kfn)(unsigned *, int, Tick *);
struct Entry { const char *name; const char *klass; kfn fn; int per_iter; };

int main(int argc, char **argv) {
    const char *only = argc > 1 ? argv[1] : nullptr;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int rounds = 8;
    std::vector<Entry> es = {
%(entries)s
    };
    unsigned *out;
    Tick *ticks;
    const int max_blocks = cus * 8 * rounds;
    CHECK(hipMalloc(&out, (size_t)max_blocks * 256 * 4));
    CHECK(hipMalloc(&ticks, (size_t)max_blocks * 4 * sizeof(Tick)));
    std::vector<Tick> h(max_blocks * 4);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int iters = 2000;
    for (const auto &e : es) {
        if (only && !strstr(e.name, only)) continue;
        for (int W : {1, 2, 4, 8}) {
            const size_t lds = (size_t)(160 * 1024 / W) - 1024;    // at most W workgroups (one wavefront per SIMD each) per CU
            CHECK(hipFuncSetAttribute((const void *)e.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const int blocks = cus * W * rounds;
            e.fn<<<cus * W, 256, lds>>>(out, 20, ticks);   // warm
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            e.fn<<<blocks, 256, lds>>>(out, iters, ticks);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(h.data(), ticks, (size_t)blocks * 4 * sizeof(Tick), hipMemcpyDeviceToHost));
            double ts = 0, tr = 0;
            for (int i = 0; i < blocks * 4; ++i) { ts += (double)h[i].shader; tr += (double)h[i].real; }
            const double ghz = ts / tr * 0.1;                                       // s_memrealtime: 100 MHz
            const double per_wave = (double)iters * e.per_iter;
            const double wave_ns = tr / (blocks * 4) * 10.0;                        // a wavefront's own elapsed time
            const double cyc_wall = ms * 1e-3 * ghz * 1e9 / (per_wave * W * rounds);
            const double cyc_wave = wave_ns * ghz / per_wave / W;                   // = cyc_wall when W wavefronts share a SIMD throughout
            printf("{\"inst\": \"%%s\", \"class\": \"%%s\", \"waves_per_simd\": %%d, \"cycles_per_inst\": %%.3f, \"cycles_per_inst_from_wave_time\": %%.3f, "
                   "\"shader_ghz\": %%.3f, \"ms\": %%.4f, \"ns_per_inst_per_simd\": %%.4f}\n",
                   e.name, e.klass, W, cyc_wall, cyc_wave, ghz, ms, ms * 1e6 / (per_wave * W * rounds));
            fflush(stdout);
        }
    }
    return 0;
}
'''


def body(asm, width):
    lines = []
    for rnd in range(8):
        for k in range(8):
            n = (k + 1) % 8
            d = "%%%d" % k
            nn = "%%%d" % n
            if width == 32:
                lo, c, a, a4, a8 = d, "%8", "%9", "%10", "%11"
            else:   # operands 8 .. 15 are the 32-bit side registers of the chains
                lo, c, a, a4, a8 = "%%%d" % (8 + k), "%16", "%17", "%18", "%19"
            s = asm.format(d=d, n=nn, c=c, a=a, a4=a4, a8=a8, k=k, lo=lo, off=(rnd * 8 + k) * 512 % 8192)
            lines.append(s)
    return "\n".join('                     "%s\\n\\t"' % ln.replace("\n\t", '\\n\\t"\n                     "') for ln in lines).lstrip()


def main():
    src = [HEAD]
    entries = []
    for name, klass, width, asm in T:
        n_inst = 2 if "counted as 2" in klass else 1
        src.append((K32 if width == 32 else K64) % {"name": name, "body": body(asm, width)})
        entries.append('        {"%s", "%s", k_%s, %d},' % (name, klass, name, 64 * n_inst))
    src.append(MAIN % {"entries": "\n".join(entries)})
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "issue_rates.hip")
    open(out, "w").write("".join(src))
    print("wrote", out, len(T), "kernels")


if __name__ == "__main__":
    main()
