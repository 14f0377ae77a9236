// Family CWT (per-series part): number_cwt_peaks (fc.py:1320) =
//   len(scipy.signal.find_peaks_cwt(x, widths=1..n, wavelet=_ricker))
// restated from scipy/signal/_peak_finding.py (find_peaks_cwt, _identify_ridge_lines, _filter_ridge_lines,
// _boolrelextrema), scipy/signal/_wavelets.py (_cwt) and fc.py:1307 (_ricker).
//   phase A (all lanes)   : CWT rows by direct convolution with the Ricker taps, relative-maximum bit mask per column
//   phase B (one lane)    : ridge-line linking, O(rows * n) with a column -> line map (the reference is O(lines^2))
//   phase C (lane = line) : length / signal-to-noise filter (10th percentile of the width-1 row in a window)
// The cwt_coefficients contraction is a separate MFMA kernel (tsfa_kernels.hip: k_cwt_gemm).
#ifndef TSFA_FAM_CWT_H
#define TSFA_FAM_CWT_H

#include "tsfa_common.h"

#define TSFA_CWTP_MAXW 16
#define TSFA_CWTP_MAXTAPS (10 * TSFA_CWTP_MAXW)

// fc.py:1307 _ricker(points, a)[k]
TSFA_DEV double ricker_tap(int points, double a, int k) {
    const double A = 2.0 / (sqrt(3.0 * a) * pow(M_PI, 0.25));
    const double wsq = a * a;
    const double vec = (double)k - ((double)points - 1.0) / 2.0;
    const double xsq = vec * vec;
    const double mod = 1.0 - xsq / wsq;
    const double gauss = exp(-xsq / (2.0 * wsq));
    return A * mod * gauss;
}

// scipy.signal.convolve(x, h, mode="same")[c] for len(h) = nw <= n: full[c + (nw-1)/2], full[m] = sum_k h[k] x[m-k]
template <class X>
TSFA_DEV double conv_same_at(X xv, int n, const double *h, int nw, int c) {
    const int m = c + (nw - 1) / 2;
    int k0 = m - (n - 1);
    if (k0 < 0) k0 = 0;
    int k1 = (m < nw - 1) ? m : (nw - 1);
    double acc = 0.0;
    // numpy's correlate kernel accumulates in increasing data index: x[j] * h[m - j]
    for (int k = k1; k >= k0; --k) acc += xv(m - k) * h[k];
    return acc;
}

struct CwtPeaksLds {
    double *red; double *row0; double *taps; unsigned short *mask; unsigned short *lcol; unsigned short *linf;
    unsigned short *colmap; int *misc;
};

// linf packing
#define TSFA_LI_LEN(v) ((v) & 63)
#define TSFA_LI_GAP(v) (((v) >> 6) & 3)
#define TSFA_LI_DEAD(v) (((v) >> 8) & 1)
#define TSFA_LI_ROW(v) (((v) >> 9) & 15)
#define TSFA_LI_PACK(len, gap, dead, row) ((unsigned short)(((len) & 63) | (((gap) & 3) << 6) | (((dead) & 1) << 8) | (((row) & 15) << 9)))

template <class X>
TSFA_DEV double number_cwt_peaks_one(const Blk &b, X xv, int n, int W, const CwtPeaksLds &L) {
    const int cap = n;  // line capacity
    blk_sync();
    for (int c = b.tid; c < n; c += b.nt) { L.mask[c] = 0; L.colmap[c] = 0; }
    blk_sync();
    // ---- phase A ----
    for (int w = 1; w <= W; ++w) {
        const int nw = (10 * w < n) ? 10 * w : n;
        blk_sync();
        for (int k = b.tid; k < nw; k += b.nt) L.taps[k] = ricker_tap(nw, (double)w, nw - 1 - k);  // reversed
        blk_sync();
        const double *h = L.taps;
        for (int c = b.tid; c < n; c += b.nt) {
            const double v = conv_same_at(xv, n, h, nw, c);
            if (w == 1) L.row0[c] = v;
            if (c > 0 && c < n - 1) {
                const double vl = conv_same_at(xv, n, h, nw, c - 1);
                const double vr = conv_same_at(xv, n, h, nw, c + 1);
                if (v > vl && v > vr) L.mask[c] |= (unsigned short)(1u << (w - 1));
            }
        }
    }
    blk_sync();
    // ---- phase B ----
    if (b.tid == 0) {
        int nlines = 0, overflow = 0;
        unsigned any = 0;
        for (int c = 0; c < n; ++c) any |= L.mask[c];
        if (any != 0) {
            int start_row = 0;
            for (int r = 0; r < W; ++r)
                if (any & (1u << r)) start_row = r;
            for (int c = 0; c < n; ++c) {
                if (L.mask[c] & (1u << start_row)) {
                    if (nlines < cap) {
                        L.lcol[nlines] = (unsigned short)c;
                        L.linf[nlines] = TSFA_LI_PACK(1, 0, 0, start_row);
                        ++nlines;
                    } else overflow = 1;
                }
            }
            const int gap_thresh = 1;  // ceil(widths[0])
            for (int row = start_row - 1; row >= 0; --row) {
                const int nprev = nlines;
                // gap += 1 for every live line; snapshot of their last columns (earliest line wins a column)
                for (int l = 0; l < nprev; ++l) {
                    const unsigned short v = L.linf[l];
                    if (TSFA_LI_DEAD(v)) continue;
                    int g = TSFA_LI_GAP(v) + 1;
                    if (g > 3) g = 3;
                    L.linf[l] = TSFA_LI_PACK(TSFA_LI_LEN(v), g, 0, TSFA_LI_ROW(v));
                    const int c = L.lcol[l];
                    if (L.colmap[c] == 0) L.colmap[c] = (unsigned short)(l + 1);
                }
                // max_distances[row] = widths[row] / 4  ->  integer distance <= floor((row + 1) / 4)
                const int D = (row + 1) / 4;
                const unsigned bit = 1u << row;
                // the snapshot must survive the whole row: remember matches in place, apply column updates after
                // (a line's lcol may only change AFTER every column of this row was matched against the snapshot;
                //  colmap holds the snapshot, so updating lcol immediately is safe)
                for (int c = 0; c < n; ++c) {
                    if (!(L.mask[c] & bit)) continue;
                    int line = -1;
                    for (int d = 0; d <= D && line < 0; ++d) {
                        int best = 0;
                        if (c - d >= 0 && L.colmap[c - d]) best = L.colmap[c - d];
                        if (d > 0 && c + d < n && L.colmap[c + d]) {
                            const int o = L.colmap[c + d];
                            if (best == 0 || o < best) best = o;
                        }
                        if (best) line = best - 1;
                    }
                    if (line >= 0) {
                        const unsigned short v = L.linf[line];
                        int len = TSFA_LI_LEN(v) + 1;
                        if (len > 63) len = 63;
                        L.linf[line] = TSFA_LI_PACK(len, 0, 0, row);
                        L.lcol[line] = (unsigned short)c;
                    } else if (nlines < cap) {
                        L.lcol[nlines] = (unsigned short)c;
                        L.linf[nlines] = TSFA_LI_PACK(1, 0, 0, row);
                        ++nlines;
                    } else {
                        overflow = 1;
                    }
                }
                // clear the snapshot; it was built from the columns the lines had BEFORE this row, which are gone
                // for matched lines, so wipe by scanning (cheap: one pass over the columns)
                for (int c = 0; c < n; ++c) L.colmap[c] = 0;
                // retire lines whose gap exceeds the threshold (they stay in the output list)
                for (int l = 0; l < nprev; ++l) {
                    const unsigned short v = L.linf[l];
                    if (!TSFA_LI_DEAD(v) && TSFA_LI_GAP(v) > gap_thresh)
                        L.linf[l] = TSFA_LI_PACK(TSFA_LI_LEN(v), TSFA_LI_GAP(v), 1, TSFA_LI_ROW(v));
                }
            }
        }
        L.misc[0] = nlines;
        L.misc[1] = overflow;
    }
    blk_sync();
    const int nlines = L.misc[0];
    const int overflow = L.misc[1];
    // ---- phase C ----
    const int min_length = (W + 3) / 4;             // ceil(rows / 4)
    const int window = (n + 19) / 20;               // ceil(num_points / 20)
    const int hf = window / 2, odd = window % 2;
    double kept = 0.0;
    for (int l = b.tid; l < nlines; l += b.nt) {
        const unsigned short v = L.linf[l];
        if (TSFA_LI_LEN(v) < min_length) continue;
        const int col = L.lcol[l], row = TSFA_LI_ROW(v);
        // signal: cwt[row, col]
        double sig;
        if (row == 0) {
            sig = L.row0[col];
        } else {
            const int w = row + 1;
            const int nw = (10 * w < n) ? 10 * w : n;
            const int m = col + (nw - 1) / 2;
            int k0 = m - (n - 1);
            if (k0 < 0) k0 = 0;
            const int k1 = (m < nw - 1) ? m : (nw - 1);
            double acc = 0.0;
            for (int k = k1; k >= k0; --k) acc += xv(m - k) * ricker_tap(nw, (double)w, nw - 1 - k);
            sig = acc;
        }
        // noise: scipy.stats.scoreatpercentile(row0[ws:we], 10)
        const int ws = (col - hf > 0) ? col - hf : 0;
        const int we = (col + hf + odd < n) ? col + hf + odd : n;
        const int m = we - ws;
        const double idx = 10.0 / 100.0 * (double)(m - 1);
        const int i0 = (int)idx;
        double s0 = 0.0, s1 = 0.0;
        for (int a = 0; a < m; ++a) {
            const double ea = L.row0[ws + a];
            int rank = 0;
            for (int c = 0; c < m; ++c) {
                const double ec = L.row0[ws + c];
                rank += (ec < ea || (ec == ea && c < a)) ? 1 : 0;
            }
            if (rank == i0) s0 = ea;
            if (rank == i0 + 1) s1 = ea;
        }
        double noise;
        if ((double)i0 == idx) {
            noise = s0;
        } else {
            const double j = (double)(i0 + 1);
            const double w0 = j - idx, w1 = idx - (double)i0;
            noise = (s0 * w0 + s1 * w1) / (w0 + w1);
        }
        const double snr = fabs(sig / noise);
        if (!(snr < 1.0)) kept += 1.0;
    }
    kept = blk_sum(b, kept);
    return overflow ? TSFA_NAN : kept;
}

template <class X>
TSFA_DEV void fam_cwtpeaks_series(const Blk &b, X xv, int n, const TsfaSpec *specs, int nspecs, double *out_row,
                                  const CwtPeaksLds &L) {
    for (int s = 0; s < nspecs; ++s) {
        const TsfaSpec sp = specs[s];
        if (sp.calc != TSFA_C_NUMBER_CWT_PEAKS) continue;
        const double v = number_cwt_peaks_one(b, xv, n, (int)sp.p[0], L);
        if (b.tid == 0) out_row[sp.col] = v;
    }
}

#endif
