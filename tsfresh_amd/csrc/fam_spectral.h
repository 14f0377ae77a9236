// Family SPECTRAL: np.fft.rfft based features and scipy.signal.welch based features.
#ifndef TSFA_FAM_SPECTRAL_H
#define TSFA_FAM_SPECTRAL_H

#include "tsfa_common.h"

// Global twiddle table shared by every series: twc[j] = cos(2 pi j / TSFA_TW_N), tws[j] = -sin(2 pi j / TSFA_TW_N),
// j < TSFA_TW_N / 2, computed once per plan on the host in float64.
#define TSFA_TW_N 65536

TSFA_DEV bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

TSFA_DEV void blk_fft_stages_dit(const Blk &b, double *re, double *im, int M, const double *twc, const double *tws);

// in-place radix-2 complex FFT (forward, e^{-i...}) of size M = 2^logM on LDS arrays
TSFA_DEV void blk_fft_pow2(const Blk &b, double *re, double *im, int M, const double *twc, const double *tws) {
    int logM = 0;
    while ((1 << logM) < M) ++logM;
    blk_sync();
    for (int i = b.tid; i < M; i += b.nt) {
#if TSFA_GPU
        const int j = (logM > 0) ? (int)(__builtin_bitreverse32((unsigned)i) >> (32 - logM)) : 0;  // v_bfrev_b32
#else
        unsigned r = 0, x = (unsigned)i;
        for (int k = 0; k < logM; ++k) {
            r = (r << 1) | (x & 1u);
            x >>= 1;
        }
        const int j = (int)r;
#endif
        if (j > i) {
            const double tr = re[i], ti = im[i];
            re[i] = re[j];
            im[i] = im[j];
            re[j] = tr;
            im[j] = ti;
        }
    }
    blk_fft_stages_dit(b, re, im, M, twc, tws);
}

// The butterfly stages of a decimation-in-time FFT of M = 2^logM points whose input already sits in bit-reversed order
// (natural-order output).
TSFA_DEV void blk_fft_stages_dit(const Blk &b, double *re, double *im, int M, const double *twc, const double *tws) {
    int logM = 0;
    while ((1 << logM) < M) ++logM;
    // Radix-2 stages taken TWO AT A TIME: the four elements i0 + {0, h, 2h, 3h} of two consecutive stages (half-lengths h
    // and 2h) meet only each other, so a thread carries them through both stages in registers -- the same butterflies
    // with the same twiddles in the same order as one stage per pass (bit-identical results), with half the LDS reads,
    // writes and barriers: the stages were what the kernel waited on (54 % issue, SQ_WAIT_INST_LDS).  An odd number of
    // stages starts with one single stage.
    int st = 0;   // stages done; stage st has half-length 1 << st
    if (logM & 1) {
        blk_sync();
        for (int t = b.tid; t < (M >> 1); t += b.nt) {   // len = 2: twiddle 1
            const int i0 = 2 * t, i1 = i0 + 1;
            const double wr = twc[0], wi = tws[0];
            const double xr = re[i1], xi = im[i1];
            const double tr = xr * wr - xi * wi;
            const double ti = xr * wi + xi * wr;
            const double ur = re[i0], ui = im[i0];
            re[i1] = ur - tr;
            im[i1] = ui - ti;
            re[i0] = ur + tr;
            im[i0] = ui + ti;
        }
        st = 1;
    }
    for (; st + 1 < logM; st += 2) {
        const int h = 1 << st;
        const int strideA = TSFA_TW_N / (2 * h), strideB = TSFA_TW_N / (4 * h);
        blk_sync();
        for (int t = b.tid; t < (M >> 2); t += b.nt) {
            const int grp = t >> st, k = t & (h - 1);
            const int i0 = grp * 4 * h + k, i1 = i0 + h, i2 = i0 + 2 * h, i3 = i0 + 3 * h;
            const double war = twc[k * strideA], wai = tws[k * strideA];
            const double wbr0 = twc[k * strideB], wbi0 = tws[k * strideB];
            const double wbr1 = twc[(k + h) * strideB], wbi1 = tws[(k + h) * strideB];
            const double x0r = re[i0], x0i = im[i0], x1r = re[i1], x1i = im[i1];
            const double x2r = re[i2], x2i = im[i2], x3r = re[i3], x3i = im[i3];
            // stage A (half-length h): (i0, i1) and (i2, i3), twiddle w_A(k)
            double tr = x1r * war - x1i * wai, ti = x1r * wai + x1i * war;
            const double a1r = x0r - tr, a1i = x0i - ti, a0r = x0r + tr, a0i = x0i + ti;
            tr = x3r * war - x3i * wai;
            ti = x3r * wai + x3i * war;
            const double a3r = x2r - tr, a3i = x2i - ti, a2r = x2r + tr, a2i = x2i + ti;
            // stage B (half-length 2h): (i0, i2) with w_B(k), (i1, i3) with w_B(k + h)
            tr = a2r * wbr0 - a2i * wbi0;
            ti = a2r * wbi0 + a2i * wbr0;
            re[i2] = a0r - tr;
            im[i2] = a0i - ti;
            re[i0] = a0r + tr;
            im[i0] = a0i + ti;
            tr = a3r * wbr1 - a3i * wbi1;
            ti = a3r * wbi1 + a3i * wbr1;
            re[i3] = a1r - tr;
            im[i3] = a1i - ti;
            re[i1] = a1r + tr;
            im[i1] = a1i + ti;
        }
    }
    blk_sync();
}

// The same transform on nb blocks of M contiguous elements at once (re / im hold nb * M doubles): one butterfly index space
// over all blocks, so a stage keeps every thread busy when M / 4 is smaller than the workgroup (the 256-sample Welch
// segments: M = 128, 32 four-element groups per stage).  Per block the operations are those of blk_fft_pow2.
TSFA_DEV void blk_fft_pow2_batch(const Blk &b, double *re, double *im, int M, int nb, const double *twc, const double *tws) {
    int logM = 0;
    while ((1 << logM) < M) ++logM;
    blk_sync();
    for (int ii = b.tid; ii < nb * M; ii += b.nt) {
        const int base = ii & ~(M - 1), i = ii & (M - 1);
#if TSFA_GPU
        const int j = (logM > 0) ? (int)(__builtin_bitreverse32((unsigned)i) >> (32 - logM)) : 0;
#else
        unsigned r = 0, x = (unsigned)i;
        for (int k = 0; k < logM; ++k) {
            r = (r << 1) | (x & 1u);
            x >>= 1;
        }
        const int j = (int)r;
#endif
        if (j > i) {
            const double tr = re[base + i], ti = im[base + i];
            re[base + i] = re[base + j];
            im[base + i] = im[base + j];
            re[base + j] = tr;
            im[base + j] = ti;
        }
    }
    int st = 0;
    if (logM & 1) {
        blk_sync();
        for (int t = b.tid; t < nb * (M >> 1); t += b.nt) {
            const int i0 = 2 * t, i1 = i0 + 1;   // (pairs never straddle a block: M is even)
            const double wr = twc[0], wi = tws[0];
            const double xr = re[i1], xi = im[i1];
            const double tr = xr * wr - xi * wi;
            const double ti = xr * wi + xi * wr;
            const double ur = re[i0], ui = im[i0];
            re[i1] = ur - tr;
            im[i1] = ui - ti;
            re[i0] = ur + tr;
            im[i0] = ui + ti;
        }
        st = 1;
    }
    const int lq = logM - 2;   // log2(M / 4)
    for (; st + 1 < logM; st += 2) {
        const int h = 1 << st;
        const int strideA = TSFA_TW_N / (2 * h), strideB = TSFA_TW_N / (4 * h);
        blk_sync();
        for (int tt = b.tid; tt < nb * (M >> 2); tt += b.nt) {
            const int blk = tt >> lq, t = tt & ((M >> 2) - 1);
            const int grp = t >> st, k = t & (h - 1);
            const int i0 = blk * M + grp * 4 * h + k, i1 = i0 + h, i2 = i0 + 2 * h, i3 = i0 + 3 * h;
            const double war = twc[k * strideA], wai = tws[k * strideA];
            const double wbr0 = twc[k * strideB], wbi0 = tws[k * strideB];
            const double wbr1 = twc[(k + h) * strideB], wbi1 = tws[(k + h) * strideB];
            const double x0r = re[i0], x0i = im[i0], x1r = re[i1], x1i = im[i1];
            const double x2r = re[i2], x2i = im[i2], x3r = re[i3], x3i = im[i3];
            double tr = x1r * war - x1i * wai, ti = x1r * wai + x1i * war;
            const double a1r = x0r - tr, a1i = x0i - ti, a0r = x0r + tr, a0i = x0i + ti;
            tr = x3r * war - x3i * wai;
            ti = x3r * wai + x3i * war;
            const double a3r = x2r - tr, a3i = x2i - ti, a2r = x2r + tr, a2i = x2i + ti;
            tr = a2r * wbr0 - a2i * wbi0;
            ti = a2r * wbi0 + a2i * wbr0;
            re[i2] = a0r - tr;
            im[i2] = a0i - ti;
            re[i0] = a0r + tr;
            im[i0] = a0i + ti;
            tr = a3r * wbr1 - a3i * wbi1;
            ti = a3r * wbi1 + a3i * wbr1;
            re[i3] = a1r - tr;
            im[i3] = a1i - ti;
            re[i1] = a1r + tr;
            im[i1] = a1i + ti;
        }
    }
    blk_sync();
}

#define TSFA_GOERTZEL_MIN 257  // non-power-of-two lengths from here on use the Goertzel sweep

// rfft of G(i), i < n, into Xr/Xi[0 .. n/2].
//   pow2 n >= 4      : half-size complex FFT + split (Xr/Xi need n/2 + 1 doubles)
//   other n <= 256   : direct DFT with a per-series twiddle table tc/ts (n doubles each, LDS)
//   other n  > 256   : Goertzel, lane = frequency bin, four bins per lane side by side: s_j = x_j + 2cos(th) s_{j-1}
//                      - s_{j-2} with x_j an LDS broadcast read, X_k = s_{n-1} e^{i th} - s_{n-2}.  O(n^2) like the
//                      direct DFT but register-resident (3 VALU per term, no twiddle table, no gather); its round-off
//                      grows like n * eps / th: <= 1.3e-9 relative on 8191 samples, far inside the 1e-6 bar.
template <class G>
TSFA_DEV void blk_rfft(const Blk &b, int n, G g, double *Xr, double *Xi, double *tc, double *ts,
                       const double *twc, const double *tws) {
    const int nh = n / 2;
    if (is_pow2(n) && n >= 4 && n <= TSFA_TW_N) {
        const int M = nh;
        blk_sync();
        for (int k = b.tid; k < M; k += b.nt) {
            Xr[k] = g(2 * k);
            Xi[k] = g(2 * k + 1);
        }
        blk_fft_pow2(b, Xr, Xi, M, twc, tws);
        const int stride = TSFA_TW_N / n;
        // in-place split: pair (k, M-k)
        for (int k = b.tid; k <= M / 2; k += b.nt) {
            if (k == 0) {
                const double zr = Xr[0], zi = Xi[0];
                Xr[0] = zr + zi;
                Xi[0] = 0.0;
                Xr[M] = zr - zi;
                Xi[M] = 0.0;
            } else {
                const int k2 = M - k;
                const double ar = Xr[k], ai = Xi[k], br = Xr[k2], bi = Xi[k2];
                // X[k] = E + w^k O,  E = (Zk + conj(Zk2))/2,  O = -i (Zk - conj(Zk2))/2
                {
                    const double er = 0.5 * (ar + br), ei = 0.5 * (ai - bi);
                    const double orr = 0.5 * (ai + bi), oi = -0.5 * (ar - br);
                    const double wr = twc[k * stride], wi = tws[k * stride];
                    Xr[k] = er + (orr * wr - oi * wi);
                    Xi[k] = ei + (orr * wi + oi * wr);
                }
                if (k2 != k) {
                    const double er = 0.5 * (br + ar), ei = 0.5 * (bi - ai);
                    const double orr = 0.5 * (bi + ai), oi = -0.5 * (br - ar);
                    const double wr = twc[k2 * stride], wi = tws[k2 * stride];
                    Xr[k2] = er + (orr * wr - oi * wi);
                    Xi[k2] = ei + (orr * wi + oi * wr);
                }
            }
        }
        blk_sync();
        return;
    }
    // Both O(n^2) routes below transform x - mean: a constant only feeds bin 0 (added back there), and without it the
    // round-off of the other bins scales with the spread of the series instead of with its offset -- Goertzel's error
    // grows like n eps |x| / th, 3e-4 absolute on bin 1 of 300 samples at 1e7 +- 1 where numpy's mixed-radix FFT has 1e-6
    // (found by the offset fixtures of the real reference, tests/golden/ref_*_offset.npz).  Any constant near the mean
    // serves; a plain strided sum finds one.
    double x_first;
    {
        double acc = 0.0;
        for (int j = b.tid; j < n; j += b.nt) acc += g(j);
        x_first = blk_sum(b, acc) / (double)n;
        if (!(x_first == x_first) || isinf(x_first)) x_first = 0.0;
    }
    if (n > 32768) {
        // Beyond the chirp-z transform's reach (its convolution would need an FFT of more than 65 536 points) the sweep runs
        // in REINSCH's form: the Goertzel recurrence s_j = x_j + 2 cos(th) s_{j-1} - s_{j-2} loses ~eps n / th^2 near th = 0
        // and th = pi (a ramp of 70 000 samples: bin 1 off by 2.3 where numpy is good to 4e-8 -- the monotone series of
        // test_series_beyond_65535_samples); with d_j = s_j -+ s_{j-1} carried instead of s_{j-2},
        //     cos(th) >= 0:  d_j = x_j - 4 sin^2(th/2) s_{j-1} + d_{j-1},  s_j = s_{j-1} + d_j
        //     cos(th) <  0:  d_j = x_j + 4 cos^2(th/2) s_{j-1} - d_{j-1},  s_j = d_j - s_{j-1}
        // nothing cancels, and X_k = s_{n-1} e^{i th} - s_{n-2} becomes s (cos th -+ 1) +- d.  One more addition per term than
        // the plain form, which the lengths below keep (their error, eps n^2 / (2 pi k), is far inside the bar).
        blk_sync();
        for (int k0 = 4 * b.tid; k0 <= nh; k0 += 4 * b.nt) {
            double kap[4], sg[4], sn[4], cs[4], s1[4], d1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                double sh, ch;
                tsfa_sincospi(2.0 * (double)(k0 + u) / (double)n, &sn[u], &cs[u]);
                tsfa_sincospi((double)(k0 + u) / (double)n, &sh, &ch);      // th / 2
                const bool pos = cs[u] >= 0.0;
                kap[u] = pos ? -4.0 * sh * sh : 4.0 * ch * ch;
                sg[u] = pos ? 1.0 : -1.0;
                s1[u] = 0.0;
                d1[u] = 0.0;
            }
            for (int j = 0; j < n; ++j) {
                const double x = g(j) - x_first;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double d = (x + kap[u] * s1[u]) + sg[u] * d1[u];
                    s1[u] = sg[u] * s1[u] + d;
                    d1[u] = d;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + u;
                if (k > nh) continue;
                // s_{n-2} = sg (s_{n-1} - d): X_r = s cos th - s_{n-2} = s (cos th - sg) + sg d = s kap / 2 + sg d
                Xr[k] = 0.5 * kap[u] * s1[u] + sg[u] * d1[u] + ((k == 0) ? (double)n * x_first : 0.0);
                Xi[k] = (k == 0 || 2 * k == n) ? 0.0 : s1[u] * sn[u];
            }
        }
        blk_sync();
        return;
    }
    if (n >= TSFA_GOERTZEL_MIN) {
        blk_sync();
        for (int k0 = 4 * b.tid; k0 <= nh; k0 += 4 * b.nt) {
            double c[4], sn[4], cs[4], s1[4], s2[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                // exp(i th), th = 2 pi k / n, argument reduced to an octant for accuracy
                tsfa_sincospi(2.0 * (double)(k0 + u) / (double)n, &sn[u], &cs[u]);
                c[u] = 2.0 * cs[u];
                s1[u] = 0.0;
                s2[u] = 0.0;
            }
            for (int j = 0; j < n; ++j) {
                const double x = g(j) - x_first;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double s0 = (x + c[u] * s1[u]) - s2[u];
                    s2[u] = s1[u];
                    s1[u] = s0;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + u;
                if (k > nh) continue;
                Xr[k] = s1[u] * cs[u] - s2[u] + ((k == 0) ? (double)n * x_first : 0.0);
                Xi[k] = (k == 0 || 2 * k == n) ? 0.0 : s1[u] * sn[u];
            }
        }
        blk_sync();
        return;
    }
    // direct DFT
    blk_sync();
    for (int j = b.tid; j < n; j += b.nt) {
        // exp(-2 pi i j / n) with the argument reduced to an octant for accuracy
        double s, c;
        tsfa_sincospi(2.0 * (double)j / (double)n, &s, &c);
        tc[j] = c;
        ts[j] = -s;
    }
    blk_sync_all();  // tc / ts may live in the HBM scratch slab (long non-power-of-two series)
    for (int k = b.tid; k <= nh; k += b.nt) {
        double ar = 0.0, ai = 0.0;
        int idx = 0;
        for (int j = 0; j < n; ++j) {
            const double x = g(j) - x_first;
            ar += x * tc[idx];
            ai += x * ts[idx];
            idx += k;
            if (idx >= n) idx -= n;
        }
        if (k == 0 || (2 * k == n)) ai = 0.0;
        if (k == 0) ar += (double)n * x_first;
        Xr[k] = ar;
        Xi[k] = ai;
    }
    blk_sync();
}

// ---------------------------------------------------------------------------------------------------------------
// Long series of arbitrary length: Bluestein's chirp-z transform.  X_k = c_k sum_j (x_j c_j) conj(c_{k-j}) with
// c_j = exp(-i pi j^2 / n) turns the length-n DFT into a circular convolution of length M = 2^ceil(log2(2n - 1)),
// i.e. three power-of-two FFTs -- O(n log n) against the O(n^2) of the Goertzel sweep above (33 M multiply-adds at
// n = 8192, every bin of which fft_aggregated reads).  M complex points do not fit LDS beside the series (256 KB at
// n = 8192), so they live in a slot of HBM scratch per workgroup (L2 / Infinity Cache resident) and pass through LDS in
// tiles: gfft_pow2 runs the first log2(T) butterfly stages on T-point tiles staged in Xr/Xi (which the transform's
// result only occupies at the very end) and the last log2(M / T) stages in place in the scratch.  The chirp phases come
// from j^2 mod 2n in integer arithmetic, so they stay accurate to the last bit for every n.
// ---------------------------------------------------------------------------------------------------------------
// TSFA_BLUESTEIN_MIN (tsfa_specs.h): the crossover against the Goertzel sweep
#define TSFA_BLUESTEIN_MAXM 65536   // the shared twiddle table (TSFA_TW_N) serves FFTs up to this size: n <= 32768

TSFA_DEV int tsfa_bitrev(int i, int bits) {
#if TSFA_GPU
    return (bits > 0) ? (int)(__builtin_bitreverse32((unsigned)i) >> (32 - bits)) : 0;
#else
    unsigned r = 0, x = (unsigned)i;
    for (int k = 0; k < bits; ++k) { r = (r << 1) | (x & 1u); x >>= 1; }
    return (int)r;
#endif
}

// The butterfly stages of a decimation-in-FREQUENCY FFT of M = 2^logM points: natural-order input, bit-reversed output
// (element i holds frequency bitrev(i)).  Stages two at a time like blk_fft_stages_dit: the four elements
// i0 + {0, h, 2h, 3h} pass the stage of half-length 2h and then the one of half-length h in registers.
TSFA_DEV void blk_fft_stages_dif(const Blk &b, double *re, double *im, int M, const double *twc, const double *tws) {
    int logM = 0;
    while ((1 << logM) < M) ++logM;
    int st = logM;   // stages left; the next one has half-length 1 << (st - 1)
    for (; st >= 2; st -= 2) {
        const int h = 1 << (st - 2);
        const int strideA = TSFA_TW_N / (4 * h), strideB = TSFA_TW_N / (2 * h);
        blk_sync();
        for (int t = b.tid; t < (M >> 2); t += b.nt) {
            const int grp = t >> (st - 2), k = t & (h - 1);
            const int i0 = grp * 4 * h + k, i1 = i0 + h, i2 = i0 + 2 * h, i3 = i0 + 3 * h;
            const double wa0r = twc[k * strideA], wa0i = tws[k * strideA];
            const double wa1r = twc[(k + h) * strideA], wa1i = tws[(k + h) * strideA];
            const double wbr = twc[k * strideB], wbi = tws[k * strideB];
            const double x0r = re[i0], x0i = im[i0], x1r = re[i1], x1i = im[i1];
            const double x2r = re[i2], x2i = im[i2], x3r = re[i3], x3i = im[i3];
            // stage A (half-length 2h): (i0, i2) with w_A(k), (i1, i3) with w_A(k + h):  a' = a + b, b' = (a - b) w
            const double a0r = x0r + x2r, a0i = x0i + x2i, d0r = x0r - x2r, d0i = x0i - x2i;
            const double a2r = d0r * wa0r - d0i * wa0i, a2i = d0r * wa0i + d0i * wa0r;
            const double a1r = x1r + x3r, a1i = x1i + x3i, d1r = x1r - x3r, d1i = x1i - x3i;
            const double a3r = d1r * wa1r - d1i * wa1i, a3i = d1r * wa1i + d1i * wa1r;
            // stage B (half-length h): (i0, i1) and (i2, i3), both with w_B(k)
            re[i0] = a0r + a1r;
            im[i0] = a0i + a1i;
            const double e0r = a0r - a1r, e0i = a0i - a1i;
            re[i1] = e0r * wbr - e0i * wbi;
            im[i1] = e0r * wbi + e0i * wbr;
            re[i2] = a2r + a3r;
            im[i2] = a2i + a3i;
            const double e1r = a2r - a3r, e1i = a2i - a3i;
            re[i3] = e1r * wbr - e1i * wbi;
            im[i3] = e1r * wbi + e1i * wbr;
        }
    }
    if (st == 1) {
        blk_sync();
        for (int t = b.tid; t < (M >> 1); t += b.nt) {   // half-length 1: twiddle 1
            const int i0 = 2 * t, i1 = i0 + 1;
            const double ur = re[i0], ui = im[i0], xr = re[i1], xi = im[i1];
            re[i0] = ur + xr;
            im[i0] = ui + xi;
            re[i1] = ur - xr;
            im[i1] = ui - xi;
        }
    }
    blk_sync();
}

// One stage of the P = log2(R) stages of an FFT of M = R T points that cross its R tiles of T contiguous points, on the R
// elements t + T q (q < R) of position t held in registers.  sq: half-length of the stage in tiles = 1 << sq (its length:
// 2 T << sq).  DIF: a' = a + b, b' = (a - b) w; else (decimation in time): a' = a + w b, b' = a - w b.
template <int P, bool DIF>
TSFA_DEV void gfft_cross_stage(double (&vr)[1 << P], double (&vi)[1 << P], int sq, int t, int T, const double *twc, const double *tws) {
    constexpr int R = 1 << P;
    const int hq = 1 << sq;
    const int stride = TSFA_TW_N / (2 * T * hq);
#pragma unroll
    for (int u = 0; u < R / 2; ++u) {
        const int kq = u & (hq - 1), q0 = ((u >> sq) << (sq + 1)) + kq, q1 = q0 + hq;
        const int ti = (t + T * kq) * stride;
        const double wr = twc[ti], wi = tws[ti];
        if (DIF) {
            const double ar = vr[q0], ai = vi[q0], dr = ar - vr[q1], di = ai - vi[q1];
            vr[q0] = ar + vr[q1];
            vi[q0] = ai + vi[q1];
            vr[q1] = dr * wr - di * wi;
            vi[q1] = dr * wi + di * wr;
        } else {
            const double xr = vr[q1], xi = vi[q1];
            const double tr = xr * wr - xi * wi, tj = xr * wi + xi * wr;
            vr[q1] = vr[q0] - tr;
            vi[q1] = vi[q0] - tj;
            vr[q0] = vr[q0] + tr;
            vi[q0] = vi[q0] + tj;
        }
    }
}

// The cross-tile stages of BOTH forward transforms, the product and the cross-tile stages of the inverse in ONE pass over
// the HBM-resident arrays: position t of every tile of A (the modulated series) and of B (the chirp filter), each already
// through its own tiles' decimation-in-time stages, pass the last P stages in registers (lengths 2 T .. M); the product
// conj(A B) of the two finished spectra -- natural order, same positions -- passes the FIRST P stages of the
// decimation-in-frequency inverse (lengths M .. 2 T) and is stored over A.  Elements t + T q are coalesced over t.
// (Round 5: separate passes moved 10 M complex values per series through HBM scratch, this form 6 M -- the transform is
// bound by that traffic: 512 KB of scratch per resident workgroup is far beyond the L2.)
template <int P>
TSFA_DEV void gfft_fused_cross(const Blk &b, double *Are, double *Aim, const double *Bre, const double *Bim, int T,
                               const double *twc, const double *tws) {
    constexpr int R = 1 << P;
    for (int t = b.tid; t < T; t += b.nt) {
        double ar[R], ai[R], br[R], bi[R];
#pragma unroll
        for (int q = 0; q < R; ++q) { ar[q] = Are[t + T * q]; ai[q] = Aim[t + T * q]; br[q] = Bre[t + T * q]; bi[q] = Bim[t + T * q]; }
#pragma unroll
        for (int sq = 0; sq < P; ++sq) {
            gfft_cross_stage<P, false>(ar, ai, sq, t, T, twc, tws);
            gfft_cross_stage<P, false>(br, bi, sq, t, T, twc, tws);
        }
#pragma unroll
        for (int q = 0; q < R; ++q) {   // conj(A B): the inverse transform is conj(FFT(conj(.))) / M
            const double pr = ar[q] * br[q] - ai[q] * bi[q], pi = -(ar[q] * bi[q] + ai[q] * br[q]);
            ar[q] = pr;
            ai[q] = pi;
        }
#pragma unroll
        for (int sq = P - 1; sq >= 0; --sq) gfft_cross_stage<P, true>(ar, ai, sq, t, T, twc, tws);
#pragma unroll
        for (int q = 0; q < R; ++q) { Are[t + T * q] = ar[q]; Aim[t + T * q] = ai[q]; }
    }
}

// The tiles' own stages of a decimation-in-time FFT of the M points src(j, &re, &im) (j in natural order): tile by tile
// (T points, the input fetched in bit-reversed order -- src is a computation, not a gather) log2(T) stages in LDS
// (lre / lim), the result to dre / dim; the cross-tile stages are gfft_fused_cross's.
template <class SRC>
TSFA_DEV void gfft_tiles_dit(const Blk &b, SRC src, double *dre, double *dim, int M, double *lre, double *lim, int T,
                             const double *twc, const double *tws) {
    int logM = 0;
    while ((1 << logM) < M) ++logM;
    for (int t0 = 0; t0 < M; t0 += T) {
        blk_sync();
        for (int i = b.tid; i < T; i += b.nt) {
            double re, im;
            src(tsfa_bitrev(t0 + i, logM), &re, &im);
            lre[i] = re;
            lim[i] = im;
        }
        blk_fft_stages_dit(b, lre, lim, T, twc, tws);
        for (int i = b.tid; i < T; i += b.nt) {
            dre[t0 + i] = lre[i];
            dim[t0 + i] = lim[i];
        }
    }
}

TSFA_DEV int bluestein_pow2(int len) {   // convolution length of a chirp-z transform of `len` points
    int M = 1;
    while (M < 2 * len - 1) M <<= 1;
    return M;
}
// The transform of an EVEN n runs on the n / 2 complex points x[2j] + i x[2j + 1] (half the convolution length, half the
// traffic) and is split afterwards like the power-of-two path's; an odd n on its n real points.
TSFA_DEV int bluestein_points(int n) { return (n & 1) ? n : n / 2; }
TSFA_DEV int bluestein_m(int n) { return bluestein_pow2(bluestein_points(n)); }
// the tile of blk_rfft_bluestein: the largest power of two the n/2 + 2 doubles of Xr / Xi hold
TSFA_DEV int bluestein_tile(int n) {
    int T = 1;
    while (2 * T <= n / 2 + 1) T <<= 1;
    return T;
}
// M / T is 4 or 8 for odd n, 2 or 4 for even n (T = 2^k <= n/2 + 1 < 2^(k+1)): the cross passes exist for 2, 4 and 8
TSFA_DEV bool bluestein_tiles_ok(int n) {
    const int M = bluestein_m(n), T = bluestein_tile(n);
    return n >= 16 && (M == 2 * T || M == 4 * T || M == 8 * T) && M <= TSFA_BLUESTEIN_MAXM && n <= 32768;
}

// rfft of G(i), i < n, into Xr/Xi[0 .. n/2] through gs (4 * bluestein_m(n) doubles of HBM scratch).
// Bluestein: Z_k = c_k sum_j (z_j c_j) conj(c_{k-j}), c_j = exp(-i pi j^2 / N) over the N = bluestein_points(n) points:
// a circular convolution of length M = 2^ceil(log2(2N - 1)) = three power-of-two FFTs.  Round 5 (0.38 ns
// per sample against 0.023 on the radix-2 path -- 321 barrier intervals, three HBM passes per transform, a 64-bit modulo
// and a sincospi per chirp value, a bit-reversed GATHER of the whole product; 11.98 ms per 5 000 series of 4096..8192):
//   * the two forward transforms (chirp filter, modulated series) run decimation-in-time from COMPUTED inputs, the tiles'
//     stages two at a time in LDS; their cross-tile stages, the product and the cross-tile stages of the inverse are ONE
//     pass over the scratch (gfft_fused_cross);
//   * the inverse runs decimation-in-FREQUENCY on that natural-order product, tile by tile with contiguous loads, and only
//     the wanted bins are picked out of the bit-reversed result;
//   * an even n is transformed as n / 2 complex points and split;
//   * chirp values exp(i pi r / n), r < 2n <= 65536, are products of two table entries, r = 256 a + b: 2 x 256 sincospi
//     per series (tab: 1024 doubles of LDS) instead of one per use; j^2 mod 2N in float64 (exact: j < 2^15).
// Accuracy: 4e-16 sum|x| on 9 .. 32 767 samples (the Goertzel sweep: 6e-10).
// tab: cos / sin of pi 256 a / n (a < 256) and of pi b / n (b < 256): 4 x 256 doubles.
template <class G>
TSFA_DEV void blk_rfft_bluestein(const Blk &b, int n, G g, double *Xr, double *Xi, double *gs, double *tab,
                                 const double *twc, const double *tws) {
    const bool half = (n & 1) == 0;
    const int N = bluestein_points(n), M = bluestein_m(n), T = bluestein_tile(n);
    int logM = 0, P = 0;
    while ((1 << logM) < M) ++logM;
    while ((T << P) < M) ++P;
    double *Are = gs, *Aim = gs + M, *Bre = gs + 2 * (size_t)M, *Bim = gs + 3 * (size_t)M;
    const double dn = (double)n, two_N = 2.0 * (double)N, inv_two_N = 1.0 / two_N;
    double *cA = tab, *sA = tab + 256, *cB = tab + 512, *sB = tab + 768;
    blk_sync();
    for (int a = b.tid; a < 512; a += b.nt) {
        double sv, cv;
        const int e = a & 255;
        // exp(i pi 256 e / n): the angle reduced to [0, 2) in integers first (256 e may exceed 2n)
        const int num = (a < 256) ? (256 * e) % (2 * n) : e;
        tsfa_sincospi((double)num / dn, &sv, &cv);
        if (a < 256) { cA[e] = cv; sA[e] = sv; } else { cB[e] = cv; sB[e] = sv; }
    }
    blk_sync();
    // exp(+i pi r / n), 0 <= r < 2n
    auto unit = [=](int ri, double *c, double *s) {
        const int a = ri >> 8, e = ri & 255;
        const double ca = cA[a], sa = sA[a], cb = cB[e], sb = sB[e];
        *c = ca * cb - sa * sb;
        *s = sa * cb + ca * sb;
    };
    // conj(c_j) = exp(+i pi j^2 / N) = exp(+i pi r / n) with r = (j^2 mod 2N) * (n / N)
    const int rmul = half ? 2 : 1;
    auto chirp = [=](int j, double *c, double *s) {
        const double jj = (double)j * (double)j;                 // exact: j < 2^15
        double r = jj - floor(jj * inv_two_N) * two_N;           // j^2 mod 2N, up to one wrap either way
        r = (r < 0.0) ? r + two_N : r;
        r = (r >= two_N) ? r - two_N : r;
        unit((int)r * rmul, c, s);
    };
    // B: the chirp filter conj(c_j), j in (-N, N) wrapped into [0, M)
    gfft_tiles_dit(b, [=](int j, double *re, double *im) {
        const int jj = (j < N) ? j : ((M - j < N) ? M - j : -1);
        if (jj < 0) { *re = 0.0; *im = 0.0; return; }
        chirp(jj, re, im);
    }, Bre, Bim, M, Xr, Xi, T, twc, tws);
    // A: z_j c_j
    gfft_tiles_dit(b, [=](int j, double *re, double *im) {
        if (j >= N) { *re = 0.0; *im = 0.0; return; }
        double c, s;
        chirp(j, &c, &s);
        const double zr = half ? g(2 * j) : g(j), zi = half ? g(2 * j + 1) : 0.0;
        *re = zr * c + zi * s;   // (zr + i zi)(c - i s)
        *im = zi * c - zr * s;
    }, Are, Aim, M, Xr, Xi, T, twc, tws);
    blk_sync_all();
    if (P == 1) gfft_fused_cross<1>(b, Are, Aim, Bre, Bim, T, twc, tws);
    else if (P == 2) gfft_fused_cross<2>(b, Are, Aim, Bre, Bim, T, twc, tws);
    else gfft_fused_cross<3>(b, Are, Aim, Bre, Bim, T, twc, tws);
    blk_sync_all();
    // the tiles of the decimation-in-frequency inverse; element i of the whole result holds bin bitrev(i), and only the
    // wanted bins are kept (in Bre / Bim: the filter's transform is spent): all N of a half-length transform, 0 .. n/2 else
    const int keep = half ? N : n / 2 + 1;
    for (int t0 = 0; t0 < M; t0 += T) {
        blk_sync();
        for (int i = b.tid; i < T; i += b.nt) {
            Xr[i] = Are[t0 + i];
            Xi[i] = Aim[t0 + i];
        }
        blk_fft_stages_dif(b, Xr, Xi, T, twc, tws);
        for (int i = b.tid; i < T; i += b.nt) {
            const int k = tsfa_bitrev(t0 + i, logM);
            if (k < keep) { Bre[k] = Xr[i]; Bim[k] = Xi[i]; }
        }
    }
    blk_sync_all();
    const double inv_m = 1.0 / (double)M;
    // Z_k = conj(conv_k) / M * c_k
    auto zbin = [=](int k, double *zr, double *zi) {
        double c, s;
        chirp(k, &c, &s);
        const double cr = Bre[k] * inv_m, ci = -(Bim[k] * inv_m);
        *zr = cr * c + ci * s;  // (cr + i ci)(c - i s)
        *zi = ci * c - cr * s;
    };
    const int nh = n / 2;
    if (!half) {
        for (int k = b.tid; k <= nh; k += b.nt) {
            double zr, zi;
            zbin(k, &zr, &zi);
            Xr[k] = zr;
            Xi[k] = (k == 0) ? 0.0 : zi;
        }
    } else {
        // split of the half-length transform (pairs k, N - k): X_k = E + w^k O, E = (Z_k + conj(Z_{N-k})) / 2,
        // O = -i (Z_k - conj(Z_{N-k})) / 2, w^k = exp(-2 pi i k / n) = conj(unit(2 k))
        for (int k = b.tid; k <= N / 2; k += b.nt) {
            if (k == 0) {
                double zr, zi;
                zbin(0, &zr, &zi);
                Xr[0] = zr + zi;
                Xi[0] = 0.0;
                Xr[N] = zr - zi;
                Xi[N] = 0.0;
            } else {
                const int k2 = N - k;
                double ar, ai, br, bi;
                zbin(k, &ar, &ai);
                zbin(k2, &br, &bi);
                {
                    const double er = 0.5 * (ar + br), ei = 0.5 * (ai - bi);
                    const double orr = 0.5 * (ai + bi), oi = -0.5 * (ar - br);
                    double c, s;
                    unit(2 * k, &c, &s);
                    const double wr = c, wi = -s;
                    Xr[k] = er + (orr * wr - oi * wi);
                    Xi[k] = ei + (orr * wi + oi * wr);
                }
                if (k2 != k) {
                    const double er = 0.5 * (br + ar), ei = 0.5 * (bi - ai);
                    const double orr = 0.5 * (bi + ai), oi = -0.5 * (br - ar);
                    double c, s;
                    unit(2 * k2, &c, &s);
                    const double wr = c, wi = -s;
                    Xr[k2] = er + (orr * wr - oi * wi);
                    Xi[k2] = ei + (orr * wi + oi * wr);
                }
            }
        }
    }
    blk_sync();
}

// scipy.signal.welch(x, nperseg=min(n, 256)) -> pxx[0 .. nperseg/2] (fs=1, hann, 50% overlap,
// constant detrend, density scaling, mean over segments).  fc.py:1418, fc.py:1809.
//   win : LDS >= 256 doubles;  pxx : LDS >= 129 doubles;  Xr/Xi/tc/ts : FFT scratch (>= 256 each is enough);
//   xcap : doubles available in Xr and in Xi (the parallel form needs a 256-double slice per wavefront)
template <class ST>
TSFA_DEV int blk_welch(const Blk &b, const ST *xs, int n, double *win, double *pxx, double *Xr, double *Xi,
                       double *tc, double *ts, const double *twc, const double *tws, int xcap = 0,
                       const double *hann256 = nullptr) {
    const int nper = (n < 256) ? n : 256;
    const int nover = nper / 2;
    const int step = nper - nover;
    const int nseg = (n - nover) / step;
    const int nf = nper / 2 + 1;
    blk_sync();
    double w2 = 0.0;
    for (int j = b.tid; j < nper; j += b.nt) {
        // scipy.signal.windows.general_cosine(M, [0.5, 0.5], sym=False): fac = linspace(-pi, pi, M + 1)[:M]
        // (the 256-sample window -- every series of 256 samples or more -- comes from the plan's table: tsfa_build_consts)
        double wv;
        if (nper == 256 && hann256 != nullptr) wv = hann256[j];
        else {
            const double fac = np_linspace_at(-M_PI, M_PI, nper + 1, j);
            wv = (nper <= 1) ? 1.0 : (0.5 + 0.5 * cos(fac));  // _len_guards: M <= 1 -> ones
        }
        win[j] = wv;
        w2 += wv * wv;
    }
    for (int k = b.tid; k < nf; k += b.nt) pxx[k] = 0.0;
    w2 = blk_sum(b, w2);
    const double scale = 1.0 / w2;
    // The 256-sample segments in BATCHES of up to four (as many half-size transforms of 128 complex points as Xr / Xi
    // hold): one butterfly index space over the batch (blk_fft_pow2_batch) instead of one segment per wavefront, whose
    // 32 four-element groups per stage left half the lanes idle and whose rounds (four for seven segments on two
    // wavefronts) each paid every stage's barrier.  The real-input split and the periodogram are taken straight from the
    // half-size spectrum, pair (k, 128 - k) by pair, in place; the periodograms are added to pxx in SEGMENT order -- the
    // additions of the sequential loop below, in the same order, on the same values.
    // (the segment means travel through b.red: at most TSFA_RED_DOUBLES / 2 segments per batch)
    const int nbmax = (xcap / 128 < TSFA_RED_DOUBLES / 2) ? xcap / 128 : TSFA_RED_DOUBLES / 2;
    if (nper == 256 && nseg > 1 && nbmax >= 2) {
        const int M = 128;
        const int stride_tw = TSFA_TW_N / 256;
        for (int s0 = 0; s0 < nseg; s0 += nbmax) {
            const int nb = (nseg - s0 < nbmax) ? (nseg - s0) : nbmax;
            blk_sync();
            // segment means: one WAVEFRONT per segment (lane-strided partial sums + the wave butterfly: the summation
            // order of the one-segment-per-wavefront form this replaces)
#if TSFA_GPU
            {
                const int lane = b.tid & 63, wave = b.tid >> 6, nwav = (b.nt + 63) >> 6;
                for (int q = wave; q < nb; q += nwav) {
                    const XsView<ST> seg{xs + (s0 + q) * step};
                    double sm = 0.0;
                    for (int j = lane; j < nper; j += 64) sm += seg[j];
                    const double mu = wave_sum(sm) / (double)nper;
                    if (lane == 0) b.red[q] = mu;
                }
            }
#else
            for (int q = b.tid; q < nb; q += b.nt) {
                const XsView<ST> seg{xs + (s0 + q) * step};
                double sm = 0.0;
                for (int j = 0; j < nper; ++j) sm += seg[j];
                b.red[q] = sm / (double)nper;
            }
#endif
            blk_sync();
            for (int i = b.tid; i < nb * M; i += b.nt) {
                const int q = i >> 7, k = i & (M - 1);
                const XsView<ST> seg{xs + (s0 + q) * step};
                const double mu = b.red[q];
                Xr[i] = (seg[2 * k] - mu) * win[2 * k];
                Xi[i] = (seg[2 * k + 1] - mu) * win[2 * k + 1];
            }
            blk_fft_pow2_batch(b, Xr, Xi, M, nb, twc, tws);
            // split + periodogram: thread = (segment, pair k <= 64); pw of bin k -> Xr[q M + k], of bin 128 -> Xi[q M]
            for (int i = b.tid; i < nb * (M / 2 + 1); i += b.nt) {
                const int q = i / (M / 2 + 1), k = i - q * (M / 2 + 1);
                double *zr = Xr + q * M, *zi = Xi + q * M;
                if (k == 0) {
                    const double a0 = zr[0], b0 = zi[0];
                    const double x0 = a0 + b0, xm = a0 - b0;       // X[0], X[128] (both real)
                    zr[0] = (x0 * x0 + 0.0 * 0.0) * scale;
                    zi[0] = (xm * xm + 0.0 * 0.0) * scale;
                } else {
                    const int k2 = M - k;
                    const double ar = zr[k], ai = zi[k], br = zr[k2], bi = zi[k2];
                    double pk, pk2 = 0.0;
                    {
                        const double er = 0.5 * (ar + br), ei = 0.5 * (ai - bi);
                        const double orr = 0.5 * (ai + bi), oi = -0.5 * (ar - br);
                        const double wr = twc[k * stride_tw], wi = tws[k * stride_tw];
                        const double xr = er + (orr * wr - oi * wi), xi = ei + (orr * wi + oi * wr);
                        pk = (xr * xr + xi * xi) * scale * 2.0;
                    }
                    if (k2 != k) {
                        const double er = 0.5 * (br + ar), ei = 0.5 * (bi - ai);
                        const double orr = 0.5 * (bi + ai), oi = -0.5 * (br - ar);
                        const double wr = twc[k2 * stride_tw], wi = tws[k2 * stride_tw];
                        const double xr = er + (orr * wr - oi * wi), xi = ei + (orr * wi + oi * wr);
                        pk2 = (xr * xr + xi * xi) * scale * 2.0;
                    }
                    zr[k] = pk;
                    if (k2 != k) zr[k2] = pk2;
                }
            }
            blk_sync();
            for (int k = b.tid; k < nf; k += b.nt) {
                double acc = pxx[k];
                for (int q = 0; q < nb; ++q) acc += (k < M) ? Xr[q * M + k] : Xi[q * M];
                pxx[k] = acc;
            }
            blk_sync();
        }
    } else
    for (int sgi = 0; sgi < nseg; ++sgi) {
        const XsView<ST> seg{xs + sgi * step};
        double sm = 0.0;
        for (int j = b.tid; j < nper; j += b.nt) sm += seg[j];
        const double mu = blk_sum(b, sm) / (double)nper;
        const double *wn = win;
        blk_rfft(b, nper, [=](int j) { return (seg[j] - mu) * wn[j]; }, Xr, Xi, tc, ts, twc, tws);
        for (int k = b.tid; k < nf; k += b.nt) {
            double pw = (Xr[k] * Xr[k] + Xi[k] * Xi[k]) * scale;
            const bool edge = (k == 0) || ((nper % 2 == 0) && (k == nf - 1));
            if (!edge) pw *= 2.0;
            pxx[k] += pw;
        }
        blk_sync();
    }
    if (nseg > 0)
        for (int k = b.tid; k < nf; k += b.nt) pxx[k] = pxx[k] / (double)nseg;
    blk_sync();
    return nf;
}

// Evaluate the SPECTRAL specs of one series.
//   Xr, Xi : LDS, >= n/2 + 2 doubles each (and >= 130)
//   tc, ts : per-series DFT twiddles, >= n doubles each when n is not a power of two (LDS or global scratch)
//   win    : LDS >= 256;  pxx : LDS >= 132;  iw : LDS ints >= 128
// ST: element type of the LDS-resident series (the input precision; read as float64)
// BL: the instantiation that holds the chirp-z transform (launches with HBM scratch; the others stay lean)
template <class ST, bool BL = true>
TSFA_DEV void fam_spectral_series(const Blk &b, const ST *xs_raw, int n, const TsfaSpec *specs, int nspecs,
                                  double *out_row, double *Xr, double *Xi, double *tc, double *ts, double *win,
                                  double *pxx, int *iw, const double *twc, const double *tws, int flags, int nlead,
                                  double *gs = nullptr, const double *hann256 = nullptr, double *chirp_tab = nullptr,
                                  int bl_min = TSFA_BLUESTEIN_PACK(TSFA_BLUESTEIN_MIN, TSFA_BLUESTEIN_MIN_EVEN)) {
    const XsView<ST> xs{xs_raw};
    // flags / nlead come from tsfa_prepare_family (host): the Welch-based specs are the first nlead of the list
    const bool need_fft = (flags & 1) != 0, need_welch = (flags & 2) != 0;
    const int nf = n / 2 + 1;
    TSFA_TICKER(tk, 0);

    // ---- Welch first (it reuses the FFT scratch), results stay in pxx ----
    int npx = 0;
    double pmax = 0.0, pmin = 0.0;
    bool pnan = false;
    if (need_welch) {
        npx = blk_welch(b, xs_raw, n, win, pxx, Xr, Xi, tc, ts, twc, tws, (n / 2 + 2 > 260) ? n / 2 + 2 : 260, hann256);
        double mx = -TSFA_INF, mn = TSFA_INF, nn = 0.0;
        for (int k = b.tid; k < npx; k += b.nt) {
            mx = fmax(mx, pxx[k]);
            mn = fmin(mn, pxx[k]);
            nn += (pxx[k] != pxx[k]) ? 1.0 : 0.0;
        }
        pmax = blk_max(b, mx);
        pmin = blk_min(b, mn);
        pnan = blk_sum(b, nn) > 0.0;
    }
    TSFA_TICK(tk, b, 170);
    for (int s = 0; s < nlead; ++s) {
        const TsfaSpec sp = specs[s];
        double v = TSFA_NAN;
        if (sp.calc == TSFA_C_SPKT_WELCH_DENSITY) {                      // fc.py:1418
            const int c = (int)sp.p[0];
            v = (c >= 0 && c < npx) ? pxx[c] : TSFA_NAN;
        } else if (sp.calc == TSFA_C_FOURIER_ENTROPY) {                  // fc.py:1809
            const int bins = (int)sp.p[0];
            // binned_entropy(pxx / max(pxx), bins)
            const double *px = pxx;
            const double pm = pmax;
            const double lo = pmin / pmax, hi = pmax / pmax;
            if (pnan || lo != lo || hi != hi || isinf(lo) || isinf(hi)) v = TSFA_NAN;
            else v = blk_binned_entropy(b, npx, [=](int i) { return px[i] / pm; }, bins, lo, hi, iw, 128);   // (iw: 128 counters in the Hann window's storage; more bins go in rounds)
        } else {
            continue;
        }
        if (b.tid == 0) out_row[sp.col] = v;
    }
    TSFA_TICK(tk, b, 171);
    if (!need_fft) return;

    // ---- full-length rfft ----
    // gs: 4 * bluestein_m(n) doubles of HBM scratch (or null: long lengths of arbitrary factorisation fall back on the
    // O(n^2) Goertzel sweep of blk_rfft)
    if (BL && gs != nullptr && chirp_tab != nullptr && n >= ((n & 1) ? (bl_min & 0xFFFF) : (int)((unsigned)bl_min >> 16)) && !is_pow2(n) && bluestein_tiles_ok(n))
        blk_rfft_bluestein(b, n, [=](int j) { return xs[j]; }, Xr, Xi, gs, chirp_tab, twc, tws);
    else
        blk_rfft(b, n, [=](int j) { return xs[j]; }, Xr, Xi, tc, ts, twc, tws);
    TSFA_TICK(tk, b, 172);
    // moments of |X| over the bin index (fc.py:1123)
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0;
    for (int k = b.tid; k < nf; k += b.nt) {
        const double a = hypot(Xr[k], Xi[k]);
        const double dk = (double)k;
        s0 += a;
        s1 += a * dk;
        s2 += a * (dk * dk);
        s3 += a * (dk * dk * dk);
        s4 += a * (dk * dk * dk * dk);
    }
    {
        double s5[5] = {s0, s1, s2, s3, s4};  // reduced together: one barrier pair, a reduce-scatter for four of them
        blk_sum_multi<5>(b, s5);
        s0 = s5[0]; s1 = s5[1]; s2 = s5[2]; s3 = s5[3]; s4 = s5[4];
    }
    const double m1 = s1 / s0, m2 = s2 / s0, m3 = s3 / s0, m4 = s4 / s0;
    const double var = m2 - m1 * m1;
    for (int s = nlead + b.tid; s < nspecs; s += b.nt) {  // one lane per column
        const TsfaSpec sp = specs[s];
        double v = TSFA_NAN;
        if (sp.calc == TSFA_C_FFT_COEFFICIENT) {                         // fc.py:1067
            const int k = (int)sp.p[0], attr = (int)sp.p[1];
            if (k >= 0 && k < nf) {
                const double r = Xr[k], i = Xi[k];
                if (attr == TSFA_FFT_REAL) v = r;
                else if (attr == TSFA_FFT_IMAG) v = i;
                else if (attr == TSFA_FFT_ABS) v = hypot(r, i);
                else v = atan2(i, r) * (180.0 / M_PI);
            }
        } else if (sp.calc == TSFA_C_FFT_AGGREGATED) {
            const int t = (int)sp.p[0];
            if (t == TSFA_FFTAGG_CENTROID) v = m1;
            else if (t == TSFA_FFTAGG_VARIANCE) v = var;
            else if (t == TSFA_FFTAGG_SKEW) v = (var < 0.5) ? TSFA_NAN : (m3 - 3.0 * m1 * var - m1 * m1 * m1) / pow(var, 1.5);
            else v = (var < 0.5) ? TSFA_NAN : (m4 - 4.0 * m1 * m3 + 6.0 * m2 * m1 * m1 - 3.0 * m1) / (var * var);
        } else {
            continue;
        }
        out_row[sp.col] = v;
    }
    TSFA_TICK(tk, b, 173);
}

#endif
