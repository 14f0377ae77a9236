// Family SEQ: lempel_ziv_complexity (fc.py:1825).
//
// The reference parses the binned sequence into phrases with a Python set of tuples.  Every phrase that enters the
// set is a previously entered phrase plus one symbol (all its proper prefixes starting at the same position were
// found in the set), so the set is prefix-closed: a trie.  The trie lives in an LDS open-addressing hash keyed by
// (parent slot, symbol); the slot index doubles as the node id.
//
// The parse is inherently sequential, so each requested `bins` value gets its own lane and its own table and the
// lanes run side by side.  Every loop iteration consumes exactly ONE symbol whether the phrase is extended or
// closed, so all lanes walk the series in lockstep (no divergence except hash collisions).
#ifndef TSFA_FAM_SEQ_H
#define TSFA_FAM_SEQ_H

#include "tsfa_common.h"

#if defined(__HIPCC__)
#define TSFA_SEQ_HD __host__ __device__ inline
#else
#define TSFA_SEQ_HD inline
#endif

#define TSFA_LZ_MAX_GROUP 8

// Most phrases a parse of n symbols over an alphabet of b symbols can produce: all phrases are distinct strings, so
// the count is maximised by taking every string of length 1, then every string of length 2, ...
TSFA_SEQ_HD int lz_max_phrases(int b, int n) {
    long long remaining = n, count = 0, of_len = b;
    int len = 1;
    while (remaining > 0) {
        const long long can = remaining / len;
        if (can <= of_len) {
            count += can;
            break;
        }
        count += of_len;
        remaining -= of_len * len;
        ++len;
        of_len = (of_len > (1LL << 40)) ? (1LL << 40) : of_len * b;
    }
    return (int)count;
}
// hash-table slots for that parse: power of two, load factor <= 0.5
TSFA_SEQ_HD int lz_table_cap(int bins, int n) {
    const int need = 2 * lz_max_phrases(bins, n) + 2;
    int p = 16;
    while (p < need) p <<= 1;
    return p;
}

// LDS budget of the family for a batch: specs are processed `group` at a time (<= TSFA_LZ_MAX_GROUP); returns the
// table slots / edge doubles the largest group needs.  Shared by the host (LDS sizing) and the emulation.
TSFA_SEQ_HD void lz_group_budget(const TsfaSpec *specs, int nspecs, int group, int maxn, int *tab_entries,
                                 int *edge_doubles) {
    int tmax = 0, emax = 0;
    for (int s0 = 0; s0 < nspecs; s0 += group) {
        int t = 0, e = 0;
        for (int k = s0; k < nspecs && k < s0 + group; ++k) {
            t += lz_table_cap((int)specs[k].p[0], maxn);
            e += (int)specs[k].p[0];
        }
        tmax = t > tmax ? t : tmax;
        emax = e > emax ? e : emax;
    }
    *tab_entries = tmax;
    *edge_doubles = emax;
}

// Evaluate the SEQ specs of one series, `group` specs at a time.
//   x(i)   : sample accessor (the kernel reads HBM directly: the series is only touched twice)
//   seq    : LDS bytes,   >= group * n
//   tab    : LDS uint32,  >= max over groups of sum(lz_table_cap(bins, maxn))
//   edges  : LDS doubles, >= max over groups of sum(bins)
//   maxn   : longest series of the batch (tables are laid out for it, identically on host and device)
template <class X>
TSFA_DEV void fam_seq_series(const Blk &b, X xv, int n, const TsfaSpec *specs, int nspecs, double *out_row,
                             unsigned char *seq, uint32_t *tab, double *edges, int group, int maxn) {
    double mn = TSFA_INF, mx = -TSFA_INF;
    for (int i = b.tid; i < n; i += b.nt) {
        const double x = xv(i);
        mn = fmin(mn, x);
        mx = fmax(mx, x);
    }
    const double vmin = blk_min(b, mn), vmax = blk_max(b, mx);

    for (int s0 = 0; s0 < nspecs; s0 += group) {
        const int nb = (nspecs - s0 < group) ? (nspecs - s0) : group;
        int toff[TSFA_LZ_MAX_GROUP + 1], eoff[TSFA_LZ_MAX_GROUP + 1];
        toff[0] = 0;
        eoff[0] = 0;
        int ttotal = 0;
#pragma unroll
        for (int t = 0; t < TSFA_LZ_MAX_GROUP; ++t) {
            const int bins = (t < nb) ? (int)specs[s0 + t].p[0] : 0;
            toff[t + 1] = toff[t] + ((t < nb) ? lz_table_cap(bins, maxn) : 0);
            eoff[t + 1] = eoff[t] + bins;
            ttotal = toff[t + 1];
        }
        blk_sync();
        // bin edges: np.linspace(min, max, bins + 1)[1:]
#pragma unroll
        for (int t = 0; t < TSFA_LZ_MAX_GROUP; ++t) {
            if (t >= nb) continue;
            const int bins = eoff[t + 1] - eoff[t];
            for (int k = b.tid; k < bins; k += b.nt) edges[eoff[t] + k] = np_linspace_at(vmin, vmax, bins + 1, k + 1);
        }
        for (int k = b.tid; k < ttotal; k += b.nt) tab[k] = 0u;
        blk_sync();
        // symbols: np.searchsorted(edges, x, side="left") = #{edges < x}
#pragma unroll
        for (int t = 0; t < TSFA_LZ_MAX_GROUP; ++t) {
            if (t >= nb) continue;
            const int bins = eoff[t + 1] - eoff[t];
            const double *ed = edges + eoff[t];
            unsigned char *sq = seq + (size_t)t * n;
            for (int i = b.tid; i < n; i += b.nt) {
                const double x = xv(i);
                int lo = 0, hi = bins;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (ed[mid] < x) lo = mid + 1;
                    else hi = mid;
                }
                sq[i] = (unsigned char)lo;
            }
        }
        blk_sync();
        for (int t = b.tid; t < nb; t += b.nt) {
            int off = 0, capt = 0;
#pragma unroll
            for (int q = 0; q < TSFA_LZ_MAX_GROUP; ++q)
                if (q == t) { off = toff[q]; capt = toff[q + 1] - toff[q]; }
            const unsigned char *sq = seq + (size_t)t * n;
            uint32_t *tb = tab + off;
            const uint32_t mask = (uint32_t)capt - 1u;
            int lg = 0;
            while ((1 << lg) < capt) ++lg;
            uint32_t node = 0u;  // 0 = root, slot + 1 otherwise
            int count = 0;
            for (int pos = 0; pos < n; ++pos) {
                const uint32_t key = ((node << 8) | (uint32_t)sq[pos]) + 1u;  // non-zero
                uint32_t h = (key * 2654435761u) >> (32 - lg);
                bool found = false;
                for (;;) {
                    const uint32_t cur = tb[h];
                    if (cur == key) { found = true; break; }
                    if (cur == 0u) break;
                    h = (h + 1u) & mask;
                }
                if (found) {
                    node = h + 1u;  // known phrase: extend it with the next symbol
                } else {
                    tb[h] = key;    // new phrase: record it and restart at the root
                    ++count;
                    node = 0u;
                }
            }
            // a trailing, already-known phrase is not added (the reference's while loop ends first)
            out_row[specs[s0 + t].col] = (double)count / (double)n;
        }
    }
    blk_sync();
}

#endif
