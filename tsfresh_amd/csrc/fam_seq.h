// Family SEQ: lempel_ziv_complexity (fc.py:1825).
//
// The reference parses the binned sequence into phrases with a Python set of tuples.  Every phrase that enters the
// set is a previously entered phrase plus one symbol (all its proper prefixes starting at the same position were
// found in the set), so the set is prefix-closed: a trie, held in LDS in the cheapest form that fits:
//   DIRECT16 : child[node * bins + symbol] (uint16 node ids) -- one dependent LDS read per symbol; small alphabets
//   HASH16   : open addressing keyed by (parent slot, symbol) packed in 16 bits, the slot index is the node id
//   HASH32   : the same with 32-bit keys (large alphabets)
//
// The parse is inherently sequential (one dependent table lookup per symbol), so the kernel's throughput is
// (series resident per CU) / (latency of one step): the tables are sized tightly from the largest possible phrase
// count so that more workgroups fit in the 160 KB of LDS, each requested `bins` value gets its own lane and table,
// and the chains of one series are dealt over the wavefronts of the workgroup so that they step concurrently.
// Every loop iteration consumes exactly ONE symbol whether the phrase is extended or closed; symbols are fetched
// four at a time (one ds_read_b32) ahead of the dependent chain.
#ifndef TSFA_FAM_SEQ_H
#define TSFA_FAM_SEQ_H

#include "tsfa_common.h"

#if defined(__HIPCC__)
#define TSFA_SEQ_HD __host__ __device__ inline
#else
#define TSFA_SEQ_HD inline
#endif

#define TSFA_LZ_MAX_GROUP 8
#define TSFA_LZ_DIRECT_MAX_BYTES 4096

enum { TSFA_LZ_DIRECT16 = 0, TSFA_LZ_HASH16 = 1, TSFA_LZ_HASH32 = 2 };

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

struct LzTable {
    int mode;   // TSFA_LZ_*
    int cap;    // hash modes: slots (power of two, load factor <= 0.6); direct: (max phrases + 1) * bins entries
    int lg;     // hash modes: log2(cap)
    int words;  // uint32 words of LDS
};
// Table of the parse of <= n symbols over `bins` symbols.  Shared by the host (LDS sizing), the device and the
// emulation, so they can never disagree.
TSFA_SEQ_HD LzTable lz_table_plan(int bins, int n) {
    LzTable t;
    const int P = lz_max_phrases(bins, n);
    const long long direct_bytes = 2LL * (P + 1) * bins;
    if (direct_bytes <= TSFA_LZ_DIRECT_MAX_BYTES) {
        t.mode = TSFA_LZ_DIRECT16;
        t.cap = (P + 1) * bins;
        t.lg = 0;
        t.words = (t.cap + 1) / 2;
    } else {
        int cap = 16, lg = 4;
        while (3LL * cap < 5LL * P + 5) { cap <<= 1; ++lg; }  // cap >= (P + 1) / 0.6
        t.cap = cap;
        t.lg = lg;
        if ((long long)(cap + 1) * bins <= 65535) {
            t.mode = TSFA_LZ_HASH16;
            t.words = cap / 2;
        } else {
            t.mode = TSFA_LZ_HASH32;
            t.words = cap;
        }
    }
    t.words = (t.words + 3) & ~3;  // 16-byte granules
    return t;
}

// bytes between the symbol rows of two chains (rows are read one 32-bit word = four symbols at a time)
TSFA_SEQ_HD int lz_seq_stride(int maxn) { return ((maxn + 3) & ~3) + 4; }

// One launch of the kernel parses up to TSFA_LZ_MAX_GROUP `bins` values side by side.  Everything that depends on
// (bins, longest series of the batch) is worked out ONCE on the host -- lz_max_phrases divides 64-bit integers, far
// too slow to repeat in every thread -- and handed to the kernel by value.
struct TsfaSeqGroup {
    int nb;                           // chains of this group
    int ttotal;                       // uint32 words of table storage
    int etotal;                       // bin edges (doubles)
    int stride;                       // bytes between the symbol rows of two chains
    int bins[TSFA_LZ_MAX_GROUP];
    int mode[TSFA_LZ_MAX_GROUP];      // TSFA_LZ_*
    int cap[TSFA_LZ_MAX_GROUP];
    int lg[TSFA_LZ_MAX_GROUP];
    int toff[TSFA_LZ_MAX_GROUP];      // table offset in uint32 words
    int eoff[TSFA_LZ_MAX_GROUP];      // edge offset in doubles
    int col[TSFA_LZ_MAX_GROUP];       // output column
};
inline void lz_build_group(const TsfaSpec *specs, int nb, int maxn, TsfaSeqGroup *g) {
    g->nb = nb;
    g->stride = lz_seq_stride(maxn);
    int t = 0, e = 0;
    for (int k = 0; k < TSFA_LZ_MAX_GROUP; ++k) {
        const int bins = (k < nb) ? (int)specs[k].p[0] : 1;
        const LzTable lt = lz_table_plan(bins, maxn);
        g->bins[k] = bins;
        g->mode[k] = lt.mode;
        g->cap[k] = lt.cap;
        g->lg[k] = lt.lg;
        g->toff[k] = t;
        g->eoff[k] = e;
        g->col[k] = (k < nb) ? specs[k].col : 0;
        if (k < nb) { t += lt.words; e += bins; }
    }
    g->ttotal = t;
    g->etotal = e;
}

// One parse.  sq: the chain's symbols (4-byte aligned, readable up to the next multiple of 4); tb: its table (zeroed).
TSFA_DEV int lz_parse(const unsigned char *sq, int n, int bins, const LzTable &lt, uint32_t *tb) {
    int count = 0;
    uint32_t node = 0u;  // 0 = root
    const uint32_t *sw = (const uint32_t *)(const void *)sq;
    if (lt.mode == TSFA_LZ_DIRECT16) {
        unsigned short *t16 = (unsigned short *)(void *)tb;
        for (int pos = 0; pos < n; pos += 4) {
            uint32_t w = sw[pos >> 2];
            const int lim = (n - pos < 4) ? (n - pos) : 4;
            for (int k = 0; k < lim; ++k, w >>= 8) {
                const uint32_t idx = node * (uint32_t)bins + (w & 255u);
                const uint32_t c = t16[idx];
                if (c != 0u) {
                    node = c;  // known phrase: extend it with the next symbol
                } else {
                    t16[idx] = (unsigned short)(++count);  // new phrase: record it and restart at the root
                    node = 0u;
                }
            }
        }
    } else if (lt.mode == TSFA_LZ_HASH16) {
        unsigned short *t16 = (unsigned short *)(void *)tb;
        const uint32_t mask = (uint32_t)lt.cap - 1u;
        const int sh = 32 - lt.lg;
        for (int pos = 0; pos < n; pos += 4) {
            uint32_t w = sw[pos >> 2];
            const int lim = (n - pos < 4) ? (n - pos) : 4;
            for (int k = 0; k < lim; ++k, w >>= 8) {
                const uint32_t key = node * (uint32_t)bins + (w & 255u) + 1u;  // non-zero, <= (cap + 1) * bins
                uint32_t h = (key * 2654435761u) >> sh;
                bool found = false;
                for (;;) {
                    const uint32_t cur = t16[h];
                    if (cur == key) { found = true; break; }
                    if (cur == 0u) break;
                    h = (h + 1u) & mask;
                }
                if (found) {
                    node = h + 1u;
                } else {
                    t16[h] = (unsigned short)key;
                    ++count;
                    node = 0u;
                }
            }
        }
    } else {
        const uint32_t mask = (uint32_t)lt.cap - 1u;
        const int sh = 32 - lt.lg;
        for (int pos = 0; pos < n; pos += 4) {
            uint32_t w = sw[pos >> 2];
            const int lim = (n - pos < 4) ? (n - pos) : 4;
            for (int k = 0; k < lim; ++k, w >>= 8) {
                const uint32_t key = ((node << 8) | (w & 255u)) + 1u;  // non-zero
                uint32_t h = (key * 2654435761u) >> sh;
                bool found = false;
                for (;;) {
                    const uint32_t cur = tb[h];
                    if (cur == key) { found = true; break; }
                    if (cur == 0u) break;
                    h = (h + 1u) & mask;
                }
                if (found) {
                    node = h + 1u;
                } else {
                    tb[h] = key;
                    ++count;
                    node = 0u;
                }
            }
        }
    }
    // a trailing, already-known phrase is not added (the reference's while loop ends first)
    return count;
}

// Evaluate one group of SEQ specs for one series.
//   x(i)   : sample accessor (the kernel reads HBM directly: the series is only touched twice)
//   seq    : LDS bytes,   >= nb * g.stride, 4-byte aligned
//   tab    : LDS uint32,  >= g.ttotal
//   edges  : LDS doubles, >= g.etotal
template <class X>
TSFA_DEV void fam_seq_series(const Blk &b, X xv, int n, const TsfaSeqGroup &g, double *out_row, unsigned char *seq,
                             uint32_t *tab, double *edges) {
    double mn = TSFA_INF, mx = -TSFA_INF;
    for (int i = b.tid; i < n; i += b.nt) {
        const double x = xv(i);
        mn = fmin(mn, x);
        mx = fmax(mx, x);
    }
    const double vmin = blk_min(b, mn), vmax = blk_max(b, mx);
    const int nb = g.nb, stride = g.stride;
    blk_sync();
    // bin edges: np.linspace(min, max, bins + 1)[1:]
#pragma unroll
    for (int t = 0; t < TSFA_LZ_MAX_GROUP; ++t) {
        if (t >= nb) continue;
        const int bins = g.bins[t];
        for (int k = b.tid; k < bins; k += b.nt) edges[g.eoff[t] + k] = np_linspace_at(vmin, vmax, bins + 1, k + 1);
    }
    for (int k = b.tid; k < g.ttotal; k += b.nt) tab[k] = 0u;
    blk_sync();
    // symbols: np.searchsorted(edges, x, side="left") = #{edges < x}
    for (int i = b.tid; i < n; i += b.nt) {
        const double x = xv(i);
#pragma unroll
        for (int t = 0; t < TSFA_LZ_MAX_GROUP; ++t) {
            if (t >= nb) continue;
            const double *ed = edges + g.eoff[t];
            int lo = 0, hi = g.bins[t];
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (ed[mid] < x) lo = mid + 1;
                else hi = mid;
            }
            seq[(size_t)t * stride + i] = (unsigned char)lo;
        }
    }
    blk_sync();
#if TSFA_GPU
    // chains are dealt over the wavefronts (chain t -> wave t % nw, lane t / nw) so that they step concurrently
    const int nw = b.nt >> 6;
    for (int t = (b.tid & 63) * nw + (b.tid >> 6); t < nb; t += b.nt) {
#else
    for (int t = b.tid; t < nb; t += b.nt) {
#endif
        LzTable lt;
        int off = 0, bins = 1, col = 0;
        lt.mode = 0; lt.cap = 0; lt.lg = 0; lt.words = 0;
#pragma unroll
        for (int q = 0; q < TSFA_LZ_MAX_GROUP; ++q)
            if (q == t) { off = g.toff[q]; bins = g.bins[q]; col = g.col[q]; lt.mode = g.mode[q]; lt.cap = g.cap[q]; lt.lg = g.lg[q]; }
        const int count = lz_parse(seq + (size_t)t * stride, n, bins, lt, tab + off);
        out_row[col] = (double)count / (double)n;
    }
    blk_sync();
}

#endif
