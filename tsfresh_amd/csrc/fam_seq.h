// Family SEQ: lempel_ziv_complexity (fc.py:1825).
//
// The reference parses the binned sequence into phrases with a Python set of tuples.  Every phrase that enters the
// set is a previously entered phrase plus one symbol (all its proper prefixes starting at the same position were
// found in the set), so the set is prefix-closed: a trie.  The trie lives in an LDS open-addressing hash keyed by
// (parent slot, symbol); the slot index doubles as the node id.  The parse itself is inherently sequential, so
// each requested `bins` value gets its own lane and its own table; lanes run the parses side by side.
#ifndef TSFA_FAM_SEQ_H
#define TSFA_FAM_SEQ_H

#include "tsfa_common.h"

#define TSFA_LZ_MAX_BINS 255

// A phrase set over n symbols holds at most ~n/2 + bins/2 phrases (all singles, then doubles, ...), so a table
// of next_pow2(n + 256) slots keeps the load factor <= 0.5.

// Evaluate the SEQ specs of one series.
//   seq   : LDS bytes, >= ntab * n
//   tab   : LDS uint32, >= ntab * cap (cap = power of two >= n + 256)
//   ntab  : how many parses can run side by side (tables that fit in LDS)
//   x(i)  : sample accessor (the kernel reads HBM directly: the series is only touched twice)
template <class X>
TSFA_DEV void fam_seq_series(const Blk &b, X xv, int n, const TsfaSpec *specs, int nspecs,
                             double *out_row, unsigned char *seq, uint32_t *tab, int ntab, int cap) {
    double mn = TSFA_INF, mx = -TSFA_INF;
    for (int i = b.tid; i < n; i += b.nt) {
        const double x = xv(i);
        mn = fmin(mn, x);
        mx = fmax(mx, x);
    }
    const double vmin = blk_min(b, mn), vmax = blk_max(b, mx);
    const uint32_t mask = (uint32_t)cap - 1u;

    for (int s0 = 0; s0 < nspecs; s0 += ntab) {
        const int nb = (nspecs - s0 < ntab) ? (nspecs - s0) : ntab;
        blk_sync();
        // symbols: np.searchsorted(np.linspace(min, max, bins + 1)[1:], x, side="left") = #{edges < x}
        for (int t = 0; t < nb; ++t) {
            const int bins = (int)specs[s0 + t].p[0];
            unsigned char *sq = seq + (size_t)t * n;
            for (int i = b.tid; i < n; i += b.nt) {
                const double x = xv(i);
                int lo = 0, hi = bins;  // edges e_k = linspace[k + 1], k in [0, bins)
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (np_linspace_at(vmin, vmax, bins + 1, mid + 1) < x) lo = mid + 1;
                    else hi = mid;
                }
                sq[i] = (unsigned char)lo;
            }
        }
        for (int k = b.tid; k < nb * cap; k += b.nt) tab[k] = 0u;
        blk_sync();
        for (int t = b.tid; t < nb; t += b.nt) {
            const unsigned char *sq = seq + (size_t)t * n;
            uint32_t *tb = tab + (size_t)t * cap;
            int count = 0, ind = 0;
            while (ind < n) {
                uint32_t node = 0u;  // 0 = root, slot + 1 otherwise
                int inc = 0;
                bool added = false;
                while (ind + inc < n) {
                    const uint32_t key = ((node << 8) | (uint32_t)sq[ind + inc]) + 1u;  // non-zero
                    uint32_t h = (key * 2654435761u) & mask;
                    bool found = false;
                    for (;;) {
                        const uint32_t cur = tb[h];
                        if (cur == key) { found = true; break; }
                        if (cur == 0u) break;
                        h = (h + 1u) & mask;
                    }
                    ++inc;
                    if (found) {
                        node = h + 1u;
                    } else {
                        tb[h] = key;
                        ++count;
                        added = true;
                        break;
                    }
                }
                if (!added) break;  // ran off the end while extending a known phrase
                ind += inc;
            }
            out_row[specs[s0 + t].col] = (double)count / (double)n;
        }
    }
    blk_sync();
}

#endif
