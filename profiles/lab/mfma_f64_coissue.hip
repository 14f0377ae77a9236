// Does v_mfma_f64_16x16x4_f64 run BESIDE another wavefront's vector instructions on gfx950, or in their place?
// (round 5: number_cwt_peaks' Ricker convolutions on the matrix cores measured slower than the float64 FMA tiles although
// the wavefronts issue 1/8 of the instructions.)  Workgroups of 512 threads = two wavefronts per SIMD, one workgroup per CU
// (the LDS request sees to that), 8 waves of the grid per CU.  The first four wavefronts of a workgroup (one per SIMD) run
// MFMA chains, the other four a loop of one kind of vector instruction; each role alone, then both.  If the pipes are
// separate the combined time is the larger of the two; if the f64 matrix instruction occupies the vector unit it is the sum.
// A diagnostic program: nothing in the package builds or loads it.  hipcc --offload-arch=gfx950 -O3 -o mfma_f64_coissue ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

// kind: 0 = v_fma_f64, 1 = v_and_b32, 2 = v_fma_f32, 3 = v_cvt_f64_f32 + nothing else, 4 = ds_read_b64
template <int KIND>
__global__ void __launch_bounds__(512) k_co(double *out, int mfma_iters, int valu_iters, int mfma_f32) {
    extern __shared__ double lds[];
    const int wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 2048; i += 512) lds[i] = 1.0 + i * 1e-9;
    __syncthreads();
    double res = 0.0;
    if (wave < 4) {
        if (mfma_f32) {
            f4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
            const float x = 1.0f + threadIdx.x * 1e-6f, y = 0.5f;
            for (int i = 0; i < mfma_iters; ++i) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
                }
            }
            res = a0[0] + a1[1];
        } else {
            d4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
            const double x = 1.0 + threadIdx.x * 1e-9, y = 0.5;
            for (int i = 0; i < mfma_iters; ++i) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
                }
            }
            res = a0[0] + a1[1];
        }
    } else {
        double r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
        const double c = 1.0000001, d = 1e-9;
        unsigned u0 = threadIdx.x, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7, u4 = u0 + 9, u5 = u0 ^ 0x55, u6 = u0 | 256, u7 = ~u0;
        float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4_ = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
        const int a8 = (threadIdx.x & 63) * 8;
        for (int i = 0; i < valu_iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (KIND == 0) {
                    asm volatile("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t"
                                 "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9"
                                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c), "v"(d));
                } else if (KIND == 1) {
                    asm volatile("v_and_b32 %0, %0, %8\n\tv_and_b32 %1, %1, %8\n\tv_and_b32 %2, %2, %8\n\tv_and_b32 %3, %3, %8\n\t"
                                 "v_and_b32 %4, %4, %8\n\tv_and_b32 %5, %5, %8\n\tv_and_b32 %6, %6, %8\n\tv_and_b32 %7, %7, %8"
                                 : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(0xfffffff7u));
                } else if (KIND == 2) {
                    asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                                 "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                                 : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4_), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(1.0000001f), "v"(1e-9f));
                } else {
                    asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:512\n\tds_read_b64 %2, %8 offset:1024\n\tds_read_b64 %3, %8 offset:1536\n\t"
                                 "ds_read_b64 %4, %8 offset:2048\n\tds_read_b64 %5, %8 offset:2560\n\tds_read_b64 %6, %8 offset:3072\n\tds_read_b64 %7, %8 offset:3584\n\t"
                                 "s_waitcnt lgkmcnt(0)"
                                 : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(a8));
                }
            }
        }
        res = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + (double)(u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7) + (double)(f0 + f1 + f2 + f3 + f4_ + f5 + f6 + f7);
    }
    if (res == 123.456) out[threadIdx.x] = res;
}

template <int KIND>
static float run(int mi, int vi, int f32) {
    double *out;
    CHECK(hipMalloc(&out, 4096));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const size_t lds = 96 * 1024;   // one workgroup per CU
    CHECK(hipFuncSetAttribute((const void *)k_co<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    k_co<KIND><<<256 * 4, 512, lds>>>(out, mi, vi, f32);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k_co<KIND><<<256 * 4, 512, lds>>>(out, mi, vi, f32);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipFree(out));
    return ms;
}

template <int KIND>
static void table(const char *name, int vi_per_mi, int f32) {
    const int MI = 2000;                 // 2000 x 16 MFMAs per wavefront
    const int VI = MI * vi_per_mi;       // x 64 vector instructions per iteration
    const float tm = run<KIND>(MI, 0, f32), tv = run<KIND>(0, VI, f32), tb = run<KIND>(MI, VI, f32);
    // 4 waves of workgroups per CU in sequence; cycles per instruction per SIMD at 2.4 GHz
    const double cyc_m = tm * 1e-3 * 2.4e9 / (4.0 * MI * 16), cyc_v = tv * 1e-3 * 2.4e9 / (4.0 * VI * 64);
    printf("{\"mfma\": \"%s\", \"beside\": \"%s\", \"ms_mfma_alone\": %.3f, \"ms_vector_alone\": %.3f, \"ms_both\": %.3f, \"sum\": %.3f, \"max\": %.3f, "
           "\"cycles_per_mfma\": %.1f, \"cycles_per_vector_inst\": %.2f, \"both_over_max\": %.3f, \"both_over_sum\": %.3f}\n",
           f32 ? "v_mfma_f32_16x16x4_f32" : "v_mfma_f64_16x16x4_f64", name, tm, tv, tb, tm + tv, tm > tv ? tm : tv, cyc_m, cyc_v,
           tb / (tm > tv ? tm : tv), tb / (tm + tv));
}

int main() {
    // vector work sized to take about as long as the MFMA chain
    table<0>("v_fma_f64", 4, 0);
    table<1>("v_and_b32", 7, 0);
    table<2>("v_fma_f32", 6, 0);
    table<4>("ds_read_b64", 2, 0);
    table<0>("v_fma_f64", 2, 1);
    table<1>("v_and_b32", 3, 1);
    table<2>("v_fma_f32", 3, 1);
    return 0;
}
