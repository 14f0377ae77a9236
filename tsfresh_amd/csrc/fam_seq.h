// Family SEQ: lempel_ziv_complexity (fc.py:1825).
//
// The reference parses the binned sequence into phrases with a Python set of tuples.  Every phrase that enters the
// set is a previously entered phrase plus one symbol (all its proper prefixes starting at the same position were
// found in the set), so the set is prefix-closed: a trie, held in LDS in the cheapest form that fits:
//   DIRECT16 : child[node * bins + symbol] (uint16 node ids) -- one dependent LDS read per symbol; small alphabets
//   HASH32   : open addressing keyed by (parent slot, symbol), the slot index is the node id (large alphabets)
//
// The parse is inherently sequential (one dependent table lookup per symbol), so the kernel's throughput is
// (series resident per CU) / (latency of one step): the tables are sized tightly from the largest possible phrase
// count so that more workgroups fit in the 160 KB of LDS, each requested `bins` value gets its own lane and table,
// and the two table kinds run on different wavefronts of the workgroup (lane = chain), each in lockstep and without
// per-symbol branches: a step is load -> compare -> select, the table entry is stored unconditionally (a known phrase
// re-stores its own value).  Every step consumes exactly ONE symbol whether the phrase is extended or closed;
// symbols are fetched four at a time (one ds_read_b32) ahead of the dependent chain.
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

enum { TSFA_LZ_DIRECT16 = 0, TSFA_LZ_HASH32 = 2, TSFA_LZ_BITS = 3 };
#define TSFA_LZ_BITS_MAX 16384  // bits of the implicit-trie table of a TSFA_LZ_BITS chain

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
    int cap;    // hash: slots (power of two, load factor <= 0.6); direct: (max phrases + 1) * bins entries
    int lg;     // hash: log2(cap)
    int words;  // uint32 words of LDS
    int nbt;    // TSFA_LZ_BITS: nodes of the implicit trie (ids 0 .. nbt-1 own a row of `bins` child bits)
};
// Table of the parse of <= n symbols over `bins` symbols.  Shared by the host (LDS sizing) and the emulation.
TSFA_SEQ_HD LzTable lz_table_plan(int bins, int n) {
    LzTable t;
    t.nbt = 0;
    const int P = lz_max_phrases(bins, n);
    const long long direct_bytes = 2LL * (P + 1) * bins;
    // chains of one launch run in the lanes of ONE wavefront when they share a code path, so the implicit-trie
    // form is preferred for every alphabet it covers (direct child tables remain for larger alphabets that fit)
    if (bins > 255 && direct_bytes <= TSFA_LZ_DIRECT_MAX_BYTES && P < 65535) {
        t.mode = TSFA_LZ_DIRECT16;
        t.cap = (P + 1) * bins;
        t.lg = 0;
        t.words = (t.cap + 1) / 2;
    } else if (bins <= 255) {
        // TSFA_LZ_BITS: the shallow levels of the trie are IMPLICIT -- a node of depth < D is numbered like a heap
        // (child of node id by symbol s: id * bins + 1 + s), so "does this child exist" is one bit of a table indexed
        // by id * bins + s: no keys, no probing, a third of the LDS.  D is the largest depth whose bit table fits
        // TSFA_LZ_BITS_MAX; only phrases longer than D (at most n / (D + 1) of them) go to a small hash table.
        // ... and no deeper than the parse can fill: it creates at most P phrases, so a level of more than P nodes is
        // mostly empty -- the LDS it would take buys more resident series instead (the kernel is bound by the latency of
        // its serial parse times the series in flight; the few deeper phrases take the hash)
        long long nodes = 1, pw = 1;  // nodes of depth < D, bins^(D-1)
        int D = 1;
        while ((nodes + pw * bins) * bins <= TSFA_LZ_BITS_MAX && pw < (long long)P) { pw *= bins; nodes += pw; ++D; }
        const int deep = n / (D + 1);
        // load factor <= 0.85 (probing only costs the rare deep steps), and not a slot more: the slot of a key is
        // mulhi(hash, cap), any cap -- a power of two rounded 3212 slots up to 4096 at 8192 samples of 100 bins, and the
        // family is bound by (series resident per CU) x (latency of a step)
        int cap = (int)((20LL * deep + 20 + 16) / 17);
        if (cap < 16) cap = 16;
        t.mode = TSFA_LZ_BITS;
        t.nbt = (int)nodes;
        t.cap = cap;
        t.lg = 0;
        t.words = (int)((nodes * bins + 31) / 32) + 1 + cap;  // + one dummy word (lanes that are in the hashed part)
    } else {
        int cap = 16, lg = 4;
        while (3LL * cap < 5LL * P + 5) { cap <<= 1; ++lg; }  // cap >= (P + 1) / 0.6
        t.mode = TSFA_LZ_HASH32;
        t.cap = cap;
        t.lg = lg;
        t.words = cap;
    }
    t.words = (t.words + 3) & ~3;  // 16-byte granules
    return t;
}

// bytes between the symbol rows of two chains (rows are read one 32-bit word = four symbols at a time)
TSFA_SEQ_HD int lz_seq_stride(int maxn) { return ((maxn + 3) & ~3) + 8; }

// One launch of the kernel parses up to TSFA_LZ_MAX_GROUP `bins` values side by side.  Everything that depends on
// (bins, longest series of the batch) is worked out ONCE on the host -- lz_max_phrases divides 64-bit integers, far
// too slow to repeat in every thread -- and handed to the kernel by value.
struct TsfaSeqGroup {
    int nb;                           // chains of this group
    int ndirect;                      // the first ndirect chains use direct tables, the rest hashed ones
    int ttotal;                       // uint32 words of table storage
    int etotal;                       // bin edges (doubles)
    int stride;                       // total bytes of the symbol rows (SeqLds: one 'row' of this size)
    int soff[TSFA_LZ_MAX_GROUP];      // byte offset of a chain's symbol row
    int sbits[TSFA_LZ_MAX_GROUP];     // bits per stored symbol: 4 (two per byte; TSFA_LZ_BITS chains of <= 16 bins) or 8
    int bins[TSFA_LZ_MAX_GROUP];
    int mode[TSFA_LZ_MAX_GROUP];      // TSFA_LZ_*
    int nbt[TSFA_LZ_MAX_GROUP];       // TSFA_LZ_BITS: implicit-trie nodes
    int cap[TSFA_LZ_MAX_GROUP];
    int lg[TSFA_LZ_MAX_GROUP];
    int toff[TSFA_LZ_MAX_GROUP];      // table offset in uint32 words
    int eoff[TSFA_LZ_MAX_GROUP];      // edge offset in doubles
    int col[TSFA_LZ_MAX_GROUP];       // output column
    int grows;                        // the symbol rows live in HBM (one byte per symbol, 16-byte aligned rows): lz_parse_bits<true>
};
// bytes of one symbol row in HBM: whole 16-byte blocks + the three blocks the parse reads ahead
TSFA_SEQ_HD int lz_grow_bytes(int maxn) { return ((maxn + 15) & ~15) + 64; }
inline void lz_build_group(const TsfaSpec *specs, int nb, int maxn, TsfaSeqGroup *g, int grows = 0) {
    g->nb = nb;
    g->grows = grows;
    int sbytes = 0;
    // chains with direct tables first: they share a wavefront (lane = chain), the hashed ones share another
    int order[TSFA_LZ_MAX_GROUP], no = 0;
    for (int pass = 0; pass < 2; ++pass)
        for (int k = 0; k < nb; ++k)
            if ((lz_table_plan((int)specs[k].p[0], maxn).mode == TSFA_LZ_DIRECT16) == (pass == 0)) order[no++] = k;
    g->ndirect = 0;
    int t = 0, e = 0;
    for (int k = 0; k < TSFA_LZ_MAX_GROUP; ++k) {
        const TsfaSpec *sp = (k < nb) ? &specs[order[k]] : nullptr;
        const int bins = sp ? (int)sp->p[0] : 1;
        const LzTable lt = lz_table_plan(bins, maxn);
        if (sp && lt.mode == TSFA_LZ_DIRECT16) g->ndirect = k + 1;
        g->bins[k] = bins;
        g->mode[k] = lt.mode;
        g->nbt[k] = lt.nbt;
        g->cap[k] = lt.cap;
        g->lg[k] = lt.lg;
        g->toff[k] = t;
        g->eoff[k] = e;
        g->col[k] = sp ? sp->col : 0;
        g->sbits[k] = (!grows && lt.mode == TSFA_LZ_BITS && bins <= 16) ? 4 : 8;
        g->soff[k] = sbytes;
        if (k < nb) {
            t += lt.words; e += bins;
            sbytes += grows ? lz_grow_bytes(maxn) : (g->sbits[k] == 4) ? lz_seq_stride((maxn + 1) / 2) : lz_seq_stride(maxn);
        }
    }
    g->stride = sbytes;
    g->ttotal = t;
    g->etotal = e;
}

// Lockstep parse of the calling lane's chain with a DIRECT16 table.  All lanes of the wavefront that call this walk the
// same positions (one series, one length); `active` masks the lanes that own a chain.
TSFA_DEV int lz_parse_direct(const unsigned char *sq, int n, int bins, unsigned short *t16) {
    int count = 0;
    uint32_t node = 0u;  // 0 = root
    const uint32_t *sw = (const uint32_t *)(const void *)sq;
    const uint32_t ub = (uint32_t)bins;
    uint32_t wnext = sw[0];
    for (int pos = 0; pos < n; pos += 4) {
        uint32_t w = wnext;
        wnext = sw[(pos >> 2) + 1];  // the row is padded by one word
        const int lim = (n - pos < 4) ? (n - pos) : 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < lim) {  // wave-uniform
                const uint32_t idx = node * ub + (w & 255u);
                w >>= 8;
                const uint32_t c = t16[idx];
                const bool fresh = (c == 0u);
                count += fresh ? 1 : 0;
                t16[idx] = (unsigned short)(fresh ? (uint32_t)count : c);  // new phrase: record it; known: unchanged
                node = fresh ? 0u : c;                                     // ... and restart at the root / extend
            }
        }
    }
    // a trailing, already-known phrase is not added (the reference's while loop ends first)
    return count;
}

// Lockstep parse with an open-addressing table keyed by (parent slot + 1, symbol); the slot index is the node id.
TSFA_DEV int lz_parse_hash(const unsigned char *sq, int n, int cap, int lg, uint32_t *tb) {
    int count = 0;
    uint32_t node = 0u;
    const uint32_t *sw = (const uint32_t *)(const void *)sq;
    const uint32_t mask = (uint32_t)cap - 1u;
    const int sh = 32 - lg;
    uint32_t wnext = sw[0];
    for (int pos = 0; pos < n; pos += 4) {
        uint32_t w = wnext;
        wnext = sw[(pos >> 2) + 1];
        const int lim = (n - pos < 4) ? (n - pos) : 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < lim) {
                const uint32_t key = ((node << 8) | (w & 255u)) + 1u;  // non-zero
                w >>= 8;
                uint32_t h = (key * 2654435761u) >> sh;
                uint32_t cur = tb[h];
                while (cur != key && cur != 0u) {  // collisions are rare (load factor <= 0.6)
                    h = (h + 1u) & mask;
                    cur = tb[h];
                }
                const bool fresh = (cur == 0u);
                tb[h] = key;
                count += fresh ? 1 : 0;
                node = fresh ? 0u : (h + 1u);
            }
        }
    }
    return count;
}

// Parse with the implicit shallow trie (bit table) + a hash table for the deep phrases (TSFA_LZ_BITS).
// tb: [ceil(nbt * bins / 32) words of child bits][cap hash slots]
#if TSFA_GPU
TSFA_DEV uint32_t lz_mad24(uint32_t a, uint32_t b, uint32_t c) { return __umul24(a, b) + c; }  // v_mad_u32_u24 (full rate)
TSFA_DEV uint32_t lz_bit(uint32_t word, uint32_t pos) { return __builtin_amdgcn_ubfe(word, pos, 1u); }
TSFA_DEV uint32_t lz_slot(uint32_t key, uint32_t cap) { return __umulhi(key * 2654435761u, cap); }
#else
TSFA_DEV uint32_t lz_mad24(uint32_t a, uint32_t b, uint32_t c) { return a * b + c; }
TSFA_DEV uint32_t lz_bit(uint32_t word, uint32_t pos) { return (word >> pos) & 1u; }
TSFA_DEV uint32_t lz_slot(uint32_t key, uint32_t cap) { return (uint32_t)(((unsigned long long)(key * 2654435761u) * cap) >> 32); }
#endif

// One symbol of a TSFA_LZ_BITS chain.  The dependent chain per symbol is what bounds the kernel (one lane per chain,
// ~1000 strictly sequential steps), so the shallow step is kept to mad24 -> min -> shift -> address -> LDS read ->
// bit extract -> mul24: a lane that is in the hashed part is clamped onto the dummy word (no select, no branch), and
// the next node is child * seen instead of compare + select.
struct LzBits {
    uint32_t *tb, *hb;
    uint32_t ub, unb, dummy, hbase, cap;
};
TSFA_DEV void lz_bits_step(const LzBits &L, uint32_t sym, uint32_t &node, int &count) {
    const bool deep = (node >= L.unb);
    uint32_t p = lz_mad24(node, L.ub, sym);
    p = (p < L.dummy) ? p : L.dummy;  // v_min_u32: hashed nodes have ids above every table index
    uint32_t *wp = L.tb + (p >> 5);
    const uint32_t word = *wp;
    uint32_t seen = lz_bit(word, p & 31u);
    *wp = word | (1u << (p & 31u));
    uint32_t child = p + 1u;
#if TSFA_GPU
    if (__any(deep))
#endif
    {
        if (deep) {
            const uint32_t key = ((node << 8) | sym) + 1u;  // non-zero
            uint32_t h = lz_slot(key, L.cap);
            uint32_t cur = L.hb[h];
            while (cur != key && cur != 0u) {
                h = (h + 1u == L.cap) ? 0u : h + 1u;
                cur = L.hb[h];
            }
            seen = (cur == 0u) ? 0u : 1u;
            L.hb[h] = key;
            child = L.hbase + h;
        }
    }
    count += (int)(1u - seen);
    node = lz_mad24(child, seen, 0u);  // fresh phrase: back to the root (0)
}

// Parse with the implicit shallow trie (bit table) + a hash table for the deep phrases (TSFA_LZ_BITS).
// tb: [ceil(nbt * bins / 32) words of child bits][dummy word][cap hash slots]
// GROWS: the row lies in HBM, a byte per symbol (TsfaSeqGroup::grows) -- read 16 symbols at a time, three reads ahead of the
// chain (a step is ~250 cycles, a miss to HBM ~5 000)
struct LzU4 { uint32_t x, y, z, w; };
template <bool GROWS = false>
TSFA_DEV int lz_parse_bits(const unsigned char *sq, int n, int bins, int nbt, int cap, int lg, uint32_t *tb, int sbits) {
    (void)lg;
    int count = 0;
    uint32_t node = 0u;
    const uint32_t *sw = (const uint32_t *)(const void *)sq;
    const bool packed = (sbits == 4);            // eight symbols per word instead of four
    const uint32_t smask = packed ? 15u : 255u;
    const int wstep = packed ? 1 : 2;            // words per eight symbols
    LzBits L;
    L.ub = (uint32_t)bins;
    L.unb = (uint32_t)nbt;
    const uint32_t nbw = (L.unb * L.ub + 31u) >> 5;  // tb[nbw] is the dummy word
    L.tb = tb;
    L.hb = tb + nbw + 1u;
    L.dummy = nbw << 5;
    L.hbase = L.dummy + 32u;  // ids of hashed nodes start above every table index (and above the dummy's)
    L.cap = (uint32_t)cap;
    if (GROWS) {
        const LzU4 *sv = (const LzU4 *)(const void *)sq;
        LzU4 q0 = sv[0], q1 = sv[1], q2 = sv[2];
        int pos = 0, vi = 0;
        for (; pos + 16 <= n; pos += 16, ++vi) {
            const LzU4 cur = q0;
            q0 = q1; q1 = q2; q2 = sv[vi + 3];   // (the row is padded by three blocks)
            const uint32_t wd[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
            for (int k = 0; k < 16; ++k) lz_bits_step(L, (wd[k >> 2] >> (8 * (k & 3))) & 255u, node, count);
        }
        if (pos < n) {
            const uint32_t wd[4] = {q0.x, q0.y, q0.z, q0.w};
            for (int k = 0; pos + k < n; ++k) lz_bits_step(L, (wd[k >> 2] >> (8 * (k & 3))) & 255u, node, count);
        }
        return count;
    }
    uint32_t wa = sw[0], wb = sw[1];
    int pos = 0;
    for (; pos + 8 <= n; pos += 8) {  // full groups: no per-symbol bounds
        uint32_t w = wa;
        const uint32_t w2 = wb;
        const int nx = ((pos >> 3) + 1) * wstep;  // rows are padded: the look-ahead stays inside
        wa = sw[nx];
        wb = sw[nx + 1];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k == 4) w = packed ? w : w2;
            const uint32_t sym = w & smask;
            w >>= sbits;
            lz_bits_step(L, sym, node, count);
        }
    }
    if (pos < n) {
        uint32_t w = wa;
        for (int k = 0; pos + k < n; ++k) {
            if (k == 4) w = packed ? w : wb;
            const uint32_t sym = w & smask;
            w >>= sbits;
            lz_bits_step(L, sym, node, count);
        }
    }
    return count;
}

// Evaluate one group of SEQ specs for one series.
//   x(i)   : sample accessor (the kernel reads HBM directly: the series is only touched twice)
//   seq    : LDS bytes,   >= g.stride, 4-byte aligned
//   tab    : LDS uint32,  >= g.ttotal
//   edges  : LDS doubles, >= g.etotal
//   GROWS  : seq points into HBM (g.grows: a byte per symbol, rows of lz_grow_bytes), a slot of g.stride bytes of this workgroup
template <bool GROWS = false, class X>
TSFA_DEV void fam_seq_series(const Blk &b, X xv, int n, const TsfaSeqGroup &g, double *out_row, unsigned char *seq,
                             uint32_t *tab, double *edges, const double *stats = nullptr) {
    TSFA_TICKER(tk, 0);
    double vmin, vmax;
    if (stats != nullptr) {   // the record k_basic left for this series (TSFA_STATS_*)
        vmin = stats[TSFA_STATS_MIN];
        vmax = stats[TSFA_STATS_MAX];
    } else {
        double mn = TSFA_INF, mx = -TSFA_INF;
        for (int i = b.tid; i < n; i += b.nt) {
            const double x = xv(i);
            mn = fmin(mn, x);
            mx = fmax(mx, x);
        }
        vmin = blk_min(b, mn);
        vmax = blk_max(b, mx);
    }
    const int nb = g.nb;
    blk_sync();
    TSFA_TICK(tk, b, 160);
    // bin edges: np.linspace(min, max, bins + 1)[1:]
#pragma unroll
    for (int t = 0; t < TSFA_LZ_MAX_GROUP; ++t) {
        if (t >= nb) continue;
        const int bins = g.bins[t];
        for (int k = b.tid; k < bins; k += b.nt) edges[g.eoff[t] + k] = np_linspace_at(vmin, vmax, bins + 1, k + 1);
    }
    for (int k = b.tid; k < g.ttotal; k += b.nt) tab[k] = 0u;
    blk_sync();
    TSFA_TICK(tk, b, 161);
    // symbols: np.searchsorted(edges, x, side="left") = #{edges < x}
    // a thread bins two neighbouring samples (a 4-bit row packs them into one byte)
    for (int i = 2 * b.tid; i < n; i += 2 * b.nt) {
        const bool two = (i + 1 < n);
        const double x0 = xv(i), x1 = two ? xv(i + 1) : x0;
#pragma unroll
        for (int t = 0; t < TSFA_LZ_MAX_GROUP; ++t) {
            if (t >= nb) continue;
            const double *ed = edges + g.eoff[t];
            int lo0 = 0, hi0 = g.bins[t], lo1 = 0, hi1 = g.bins[t];
            while (lo0 < hi0 || lo1 < hi1) {
                if (lo0 < hi0) { const int mid = (lo0 + hi0) >> 1; if (ed[mid] < x0) lo0 = mid + 1; else hi0 = mid; }
                if (lo1 < hi1) { const int mid = (lo1 + hi1) >> 1; if (ed[mid] < x1) lo1 = mid + 1; else hi1 = mid; }
            }
            unsigned char *row = seq + g.soff[t];
            if (GROWS) {   // i is even, the row 16-byte aligned: one 16-bit store (the byte behind an odd n is padding)
                *(unsigned short *)(void *)(row + i) = (unsigned short)(lo0 | (two ? (lo1 << 8) : 0));
            } else if (g.sbits[t] == 4) {
                row[i >> 1] = (unsigned char)(lo0 | (two ? (lo1 << 4) : 0));
            } else {
                row[i] = (unsigned char)lo0;
                if (two) row[i + 1] = (unsigned char)lo1;
            }
        }
    }
    if (GROWS) blk_sync_all();   // the rows are global memory: the parsing wavefronts read what the others stored
    else blk_sync();
    TSFA_TICK(tk, b, 162);
    // lane = chain: the direct chains on wavefront 0, the hashed ones on wavefront 1 (if the workgroup has one)
    {
#if TSFA_GPU
        // direct chains: wavefront 0, lane = chain.  Hashed chains: dealt over the remaining wavefronts (a probe loop
        // stalls only the chains that share its wavefront).
        const int lane = b.tid & 63, wave = b.tid >> 6, nw = b.nt >> 6;
        const int nhw = (nw > 1) ? nw - 1 : 1;                 // wavefronts that take hashed chains
        const int hw = (nw > 1) ? wave - 1 : 0;                // this wavefront's index among them (-1: none)
        const int hidx = lane * nhw + hw;                      // hashed chain of this lane
        const bool do_direct = (wave == 0) && (lane < g.ndirect);
        const bool do_hash = (hw >= 0) && (hidx < nb - g.ndirect);
        const int td = lane, th = g.ndirect + hidx;
#else
        for (int t = 0; t < nb; ++t) {
        const bool do_direct = (t < g.ndirect), do_hash = !do_direct;
        const int td = t, th = t;
#endif
        if (do_direct) {
            int off = 0, bins = 1, col = 0, so = 0;
#pragma unroll
            for (int q = 0; q < TSFA_LZ_MAX_GROUP; ++q)
                if (q == td) { off = g.toff[q]; bins = g.bins[q]; col = g.col[q]; so = g.soff[q]; }
            const int count = lz_parse_direct(seq + so, n, bins, (unsigned short *)(void *)(tab + off));
            out_row[col] = (double)count / (double)n;
        }
        if (do_hash) {
            int off = 0, cap = 16, lg = 4, col = 0, mode = TSFA_LZ_HASH32, nbt = 0, hbins = 1, so = 0, sb = 8;
#pragma unroll
            for (int q = 0; q < TSFA_LZ_MAX_GROUP; ++q)
                if (q == th) { off = g.toff[q]; cap = g.cap[q]; lg = g.lg[q]; col = g.col[q]; mode = g.mode[q]; nbt = g.nbt[q]; hbins = g.bins[q]; so = g.soff[q]; sb = g.sbits[q]; }
            const int count = (mode == TSFA_LZ_BITS) ? lz_parse_bits<GROWS>(seq + so, n, hbins, nbt, cap, lg, tab + off, sb)
                                                     : lz_parse_hash(seq + so, n, cap, lg, tab + off);
            out_row[col] = (double)count / (double)n;
        }
#if !TSFA_GPU
        }
#endif
    }
    TSFA_TICK(tk, b, 163);
    blk_sync();
    TSFA_TICK(tk, b, 164);
}

#endif
