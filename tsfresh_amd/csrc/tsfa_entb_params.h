// Shape parameters of the bit-matrix entropy sweep (fam_entropy_bits.h), shared by the kernels, the LDS layout and
// the host's launch decisions.
#ifndef TSFA_ENTB_PARAMS_H
#define TSFA_ENTB_PARAMS_H
#include <stddef.h>
#if defined(__HIPCC__)
#define TSFA_ENTB_HD __host__ __device__
#else
#define TSFA_ENTB_HD
#endif

#define TSFA_ENTB_QW 11                   // diagonal words per column part, series up to TSFA_ENTB_MAXN samples
#define TSFA_ENTB_S (TSFA_ENTB_QW + 1)    // words per table entry (one halo word: the rotation by up to 31 bits); 48 B: three ds_read_b128
#define TSFA_ENTB_QW_LONG 3               // ... and for longer series (TSFA_ENTB_MAXN_LONG): 16-byte entries, so that the table of one
                                          // column part -- (n + 1) entries -- still fits LDS: 64 KB at 4096 samples
#define TSFA_ENTB_MAXT 14                 // tasks (strip x tolerance) per wavefront: register-resident ranges + counters
#define TSFA_ENTB_MAXK 6                  // tolerances per batch
#define TSFA_ENTB_STRIP 30                // templates per half-strip (32 lanes, two halo lanes); a wavefront sweeps two
#define TSFA_ENTB_MAXN 1024
#define TSFA_ENTB_MAXN_WIDE 2048          // ... and to here with ONE workgroup per CU (round 6): 6 column parts instead of the 22 of the
                                          // 16-byte-entry form, every tolerance's tasks still in the registers of 16 wavefronts
#define TSFA_ENTB_MAXN_LONG 4096          // beyond: the series, its sorted copy and the ranges of one tolerance exceed a CU's LDS
#define TSFA_ENTB_MAXWAVES 16
#define TSFA_ENTH_S 2                     // fam_entropy_hbits.h: words per table entry: one diagonal word + its halo
#define TSFA_ENTH_MAXN 17408              // fam_entropy_hbits.h: the table of one column part, (n + 1) x 17 / 16 x 8 bytes, fits a CU's LDS (17 x 1024: 17 entries per thread)

// tolerances per round: the (strip, tolerance) tasks of a round live in the wavefronts' registers
static inline TSFA_ENTB_HD int entb_kround(int maxn, int nk, int nw) {
    const int nstrips = ((maxn - 1 + TSFA_ENTB_STRIP - 1) / TSFA_ENTB_STRIP + 1) / 2;
    int kr = (TSFA_ENTB_MAXT * nw) / (nstrips > 0 ? nstrips : 1);
    if (kr > nk) kr = nk;
    if (kr > TSFA_ENTB_MAXK) kr = TSFA_ENTB_MAXK;
    return kr < 1 ? 1 : kr;
}

// words of the LDS work region (ranges / table / counters take turns in it)
//   S: words per table entry (QW + 1); kcap: tolerances per round (entb_kround)
static inline TSFA_ENTB_HD size_t entb_work_words(int maxn, int S = TSFA_ENTB_S, int kcap = TSFA_ENTB_MAXK) {
    size_t p2 = 1;
    while (p2 < (size_t)maxn) p2 <<= 1;
    const size_t ranges = 2 * p2 + (size_t)kcap * maxn;  // sorted copy (float64, padded) + packed ranges of a round
    const size_t table = (size_t)(maxn + 1) * S + (size_t)TSFA_ENTB_MAXWAVES * S;
    const size_t counts = (size_t)kcap * maxn + 2 * (2 * TSFA_ENTB_MAXK * TSFA_ENTB_MAXWAVES * 4) + 2 * TSFA_ENTB_MAXK * TSFA_ENTB_MAXWAVES + 12;  // + partial products, integer slots, log N of the two template counts
    size_t w = ranges > table ? ranges : table;
    if (counts > w) w = counts;
    return w + 8;
}


// wavefronts per workgroup for series of up to maxn samples, tolerances nk: every (strip, tolerance) task in registers
static inline TSFA_ENTB_HD int entb_waves_for(int maxn, int nk) {
    const int nstrips = ((maxn - 1 + TSFA_ENTB_STRIP - 1) / TSFA_ENTB_STRIP + 1) / 2;  // pairs of half-strips
    if (nk > TSFA_ENTB_MAXK) nk = TSFA_ENTB_MAXK;
    const int need = (nstrips * nk + TSFA_ENTB_MAXT - 1) / TSFA_ENTB_MAXT;
    int w = 1;
    while (w < need) w <<= 1;
    // at most two samples per thread: the sample sort, the range bisections and the table build are per-sample work
    // (and the register sort handles 1, 2 or 4 keys per thread), whatever the number of tolerances
    int p2 = 1;
    while (p2 < maxn) p2 <<= 1;
    while (w * 128 < p2 && w < TSFA_ENTB_MAXWAVES) w <<= 1;
    return w < 1 ? 1 : w;
}
#endif
