// The experiment north_star names: fft_coefficient (np.fft.rfft(x)[k], k = 0 .. 99; fc.py:1067-1120) of 1024-sample
// series as a dense contraction  X[n_series x 1024] . T[1024 x 200]  (T = cos / -sin twiddle columns) on the float64
// matrix cores (v_mfma_f64_16x16x4_f64), against what the product does (radix-2 LDS FFT inside k_spectral, which also
// feeds fft_aggregated's 513 magnitudes and therefore stays whatever this kernel does).
//
//   hipcc --offload-arch=gfx950 -O3 profiles/lab/mfma_dft.hip -o profiles/lab/build/mfma_dft && profiles/lab/build/mfma_dft
//
// Workgroup = 4 wavefronts = 64 series (one 16-series M tile per wavefront), N = 208 columns (13 tiles of 16: 200 used),
// K = 1024 in chunks of 16: the twiddle chunk [16 x 208] and the sample chunk [64 x 16] are staged in LDS, double
// buffered; 52 MFMAs per wavefront and chunk against 56 ds_read_b64.  Prints time, TFLOP/s, and the largest error
// against a float64 DFT evaluated on the host for a few series.  A diagnostic program: nothing in the package loads it.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int L = 1024, NB = 100, NC = 2 * NB, NT = 13, NCP = NT * 16, KC = 16, MT = 64;

__global__ void __launch_bounds__(256) k_dft_mfma(const float *__restrict__ x, const double *__restrict__ T, double *__restrict__ out,
                                                   int n_series) {
    // T: [L][NCP] row-major (row k: 208 columns)
    __shared__ double sB[2][KC][NCP + 2];   // +2: rows start on different banks
    __shared__ double sA[2][MT][KC + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const int64_t s0 = (int64_t)blockIdx.x * MT;
    d4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
    auto stage = [&](int buf, int k0) {
        for (int i = tid; i < KC * NCP; i += 256) {
            const int k = i / NCP, c = i - k * NCP;
            sB[buf][k][c] = T[(size_t)(k0 + k) * NCP + c];
        }
        for (int i = tid; i < MT * KC; i += 256) {
            const int m = i / KC, k = i - m * KC;
            const int64_t s = s0 + m;
            sA[buf][m][k] = (s < n_series) ? (double)x[s * L + k0 + k] : 0.0;
        }
    };
    stage(0, 0);
    __syncthreads();
    for (int c = 0; c < L / KC; ++c) {
        const int buf = c & 1;
        if (c + 1 < L / KC) stage(buf ^ 1, (c + 1) * KC);
#pragma unroll
        for (int ks = 0; ks < KC; ks += 4) {
            const double a = sA[buf][wave * 16 + r][ks + kq];        // A[i = r][k = kq]
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const double bv = sB[buf][ks + kq][t * 16 + r];      // B[k = kq][j = r]
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // D[i = 4 v + kq][j = r]
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = t * 16 + r;
        if (col >= NC) continue;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int64_t s = s0 + wave * 16 + 4 * v + kq;
            if (s < n_series) out[s * NC + col] = acc[t][v];
        }
    }
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 100000;
    std::vector<float> hx((size_t)n * L);
    unsigned long long st = 88172645463325252ull;
    for (auto &v : hx) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; v = (float)((double)(st >> 11) / 9007199254740992.0 - 0.5); }
    std::vector<double> hT((size_t)L * NCP, 0.0);
    for (int k = 0; k < L; ++k)
        for (int b = 0; b < NB; ++b) {
            const int idx = (int)(((long long)k * b) % L);     // exact phase reduction
            const double ang = 2.0 * M_PI * (double)idx / (double)L;
            hT[(size_t)k * NCP + 2 * b] = cos(ang);
            hT[(size_t)k * NCP + 2 * b + 1] = -sin(ang);
        }
    float *dx; double *dT, *dout;
    CHECK(hipMalloc(&dx, hx.size() * 4));
    CHECK(hipMalloc(&dT, hT.size() * 8));
    CHECK(hipMalloc(&dout, (size_t)n * NC * 8));
    CHECK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dT, hT.data(), hT.size() * 8, hipMemcpyHostToDevice));
    const unsigned grid = (unsigned)((n + MT - 1) / MT);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k_dft_mfma<<<grid, 256>>>(dx, dT, dout, n);
    CHECK(hipDeviceSynchronize());
    const int reps = 10;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) k_dft_mfma<<<grid, 256>>>(dx, dT, dout, n);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    std::vector<double> ho((size_t)n * NC);
    CHECK(hipMemcpy(ho.data(), dout, ho.size() * 8, hipMemcpyDeviceToHost));
    double maxrel = 0.0;
    for (int s : {0, 1, n / 2, n - 1}) {
        for (int b = 0; b < NB; ++b) {
            long double re = 0, im = 0;
            for (int k = 0; k < L; ++k) {
                const int idx = (int)(((long long)k * b) % L);
                const long double ang = 2.0L * 3.141592653589793238462643383279502884L * idx / L;
                re += (long double)hx[(size_t)s * L + k] * cosl(ang);
                im -= (long double)hx[(size_t)s * L + k] * sinl(ang);
            }
            const double mag = sqrt((double)(re * re + im * im)) + 1e-300;
            maxrel = fmax(maxrel, fabs(ho[(size_t)s * NC + 2 * b] - (double)re) / mag);
            maxrel = fmax(maxrel, fabs(ho[(size_t)s * NC + 2 * b + 1] - (double)im) / mag);
        }
    }
    const double flop = 2.0 * (double)n * L * NCP;
    printf("{\"kernel\": \"k_dft_mfma\", \"n_series\": %d, \"length\": %d, \"bins\": %d, \"ms\": %.4f, \"tflops_f64_incl_padding\": %.2f, "
           "\"mfma_per_launch\": %.0f, \"max_err_rel_to_bin_magnitude\": %.3e}\n",
           n, L, NB, ms, flop / (ms * 1e-3) / 1e12, (double)grid * 4 * (L / 4) * NT, maxrel);
    return 0;
}
