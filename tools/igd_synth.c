/* igd_synth.c -- deterministic synthetic IGD databases and query sets.
 *
 * Integer-only generation (a counter-based splitmix64; no libm, no numpy bit-stream), so
 * the build container and the GPU box produce byte-identical inputs.
 *
 *   genome 0 ("hg38"):  24 contigs chr1..chr22,chrX,chrY with the hg38 lengths
 *   genome 1 ("small"): chr1,chr2,chr3,chrX, 5e7 bp each            (BASELINE config 1)
 *
 * Database intervals, per file f and interval k (seed_f = seed + f):
 *   contig ~ length, value ~ U{0..1000},
 *   len_mode 0: L = 200 + min(E, 9800), E ~ piecewise-linear exponential, mean ~ 800
 *   len_mode 1: L ~ U[lenA, lenB]
 *   start ~ U[0, contig_len - L);  clustered=1: half of the intervals are placed around
 *   2000 hot spots (sd ~ 20 kb) to create very dense tiles.
 * Queries: contig ~ length, L ~ U[minLen,maxLen], start ~ U[0, len-L); optional contig id
 *   `unknown_every`-th query gets contig id -1 (an unknown name in BED text: "chr9");
 *   sorted=1 orders them by (contig, start) like a position-sorted BED.
 *
 * Built as libigd_synth.so (ctypes from bench.py / tests) and as bin/igd_synth.
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "igd_core.h"
#include "igd_synth_writer.h"

static const char *HG38_NAME[24] = {"chr1", "chr2", "chr3", "chr4", "chr5", "chr6", "chr7", "chr8",
    "chr9", "chr10", "chr11", "chr12", "chr13", "chr14", "chr15", "chr16", "chr17", "chr18",
    "chr19", "chr20", "chr21", "chr22", "chrX", "chrY"};
static const int64_t HG38_LEN[24] = {248956422, 242193529, 198295559, 190214555, 181538259,
    170805979, 159345973, 145138636, 138394717, 133797422, 135086622, 133275309, 114364328,
    107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468,
    156040895, 57227415};
static const char *SMALL_NAME[4] = {"chr1", "chr2", "chr3", "chrX"};
static const int64_t SMALL_LEN[4] = {50000000, 50000000, 50000000, 50000000};

typedef struct { int n; const char *const *name; const int64_t *len; int64_t cum[25]; } genome_t;

static void genome_init(genome_t *g, int which)
{
    if (which == 1) { g->n = 4; g->name = SMALL_NAME; g->len = SMALL_LEN; }
    else { g->n = 24; g->name = HG38_NAME; g->len = HG38_LEN; }
    g->cum[0] = 0;
    for (int i = 0; i < g->n; i++) g->cum[i + 1] = g->cum[i] + g->len[i];
}

static inline uint64_t mix64(uint64_t seed, uint64_t ctr)
{
    uint64_t z = seed * 0xD6E8FEB86659FD93ULL + (ctr + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
/* uniform integer in [0, range) */
static inline int64_t below(uint64_t r, int64_t range)
{
    return range <= 0 ? 0 : (int64_t)(((unsigned __int128)r * (unsigned __int128)(uint64_t)range) >> 64);
}
static int pick_contig(const genome_t *g, uint64_t r)
{
    int64_t x = below(r, g->cum[g->n]);
    int c = 0;
    while (c + 1 < g->n && x >= g->cum[c + 1]) c++;
    return c;
}
/* ~Exp(mean 800) in integer arithmetic: geometric number of halvings + linear remainder */
static int32_t exp800(uint64_t r)
{
    int e = r ? __builtin_clzll(r) : 64;                 /* P(e=k) = 2^-(k+1) */
    uint64_t mant = e < 63 ? ((r << (e + 1)) >> 54) : 0; /* 10 fresh bits */
    int64_t units = (int64_t)e * 1024 + (1023 - (int64_t)mant);
    return (int32_t)(units * 555 / 1024);                /* 555 ~ 800 ln 2 */
}

const char *igd_synth_contig_name(int genome, int i)
{
    genome_t g; genome_init(&g, genome);
    return (i >= 0 && i < g.n) ? g.name[i] : "chr9";
}
int igd_synth_ncontigs(int genome) { genome_t g; genome_init(&g, genome); return g.n; }

int igd_synth_db(const char *igd_path, int32_t nFiles, int32_t perFile, uint64_t seed, int32_t nbp_log,
                 int genome, int len_mode, int32_t lenA, int32_t lenB, int clustered, int gType)
{
    genome_t g; genome_init(&g, genome);
    const int64_t n = (int64_t)nFiles * perFile;
    igdc_interval *iv = (igdc_interval *)calloc((size_t)(n ? n : 1), sizeof(igdc_interval));
    int32_t *nr = (int32_t *)calloc((size_t)nFiles + 1, sizeof(int32_t));
    double *avg = (double *)calloc((size_t)nFiles + 1, sizeof(double));
    char **names = (char **)calloc((size_t)nFiles + 1, sizeof(char *));
    if (!iv || !nr || !avg || !names) return -1;
    const int nSpots = 2000;
    for (int32_t f = 0; f < nFiles; f++) {
        const uint64_t sf = seed + (uint64_t)f;
        double sum = 0;
        for (int32_t k = 0; k < perFile; k++) {
            const uint64_t b = (uint64_t)k * 8;
            int c = pick_contig(&g, mix64(sf, b));
            int32_t L;
            if (len_mode == 1) L = lenA + (int32_t)below(mix64(sf, b + 1), (int64_t)lenB - lenA + 1);
            else { int32_t e = exp800(mix64(sf, b + 1)); L = 200 + (e < 9800 ? e : 9800); }
            int64_t room = g.len[c] - L;
            int64_t s = below(mix64(sf, b + 2), room);
            if (clustered && (mix64(sf, b + 4) & 1)) {
                /* hot spot h: a fixed position of the whole genome; offset = sum of 4 uniforms */
                uint64_t h = mix64(sf, b + 5) % nSpots;
                int64_t gp = below(mix64(0xC0FFEEULL + seed, h), g.cum[g.n]);
                c = 0;
                while (c + 1 < g.n && gp >= g.cum[c + 1]) c++;
                int64_t center = gp - g.cum[c];
                uint64_t r = mix64(sf, b + 6);
                int64_t off = (int64_t)(r & 0xFFFF) + (int64_t)((r >> 16) & 0xFFFF) +
                              (int64_t)((r >> 32) & 0xFFFF) + (int64_t)(r >> 48) - 2 * 65535;
                s = center + off * 20000 / 37837;       /* sd of the 4-uniform sum = 37837 */
                room = g.len[c] - L;
                if (s < 0) s = 0;
                if (s > room) s = room;
            }
            igdc_interval *x = &iv[(int64_t)f * perFile + k];
            x->file = f; x->ctg = c; x->start = (int32_t)s; x->end = (int32_t)(s + L);
            x->value = (int32_t)(mix64(sf, b + 3) % 1001);
            sum += L;
        }
        nr[f] = perFile;
        avg[f] = perFile ? sum / perFile : 0;
        names[f] = (char *)malloc(24);
        snprintf(names[f], 24, "f%05d.bed", f);
    }
    int rc = igdc_write_igd(igd_path, 1 << nbp_log, gType, g.n, g.name, n, iv, nFiles,
                            (const char *const *)names, nr, avg);
    for (int32_t f = 0; f < nFiles; f++) free(names[f]);
    free(names); free(nr); free(avg); free(iv);
    return rc;
}

/* BED text of the database's source files (so that the REFERENCE's `igd create` can build the
 * same logical database for cross-checks): out_dir/f%05d.bed, 5 columns */
int igd_synth_db_beds(const char *out_dir, int32_t nFiles, int32_t perFile, uint64_t seed, int genome,
                      int len_mode, int32_t lenA, int32_t lenB)
{
    genome_t g; genome_init(&g, genome);
    for (int32_t f = 0; f < nFiles; f++) {
        char path[4096];
        snprintf(path, sizeof path, "%s/f%05d.bed", out_dir, f);
        FILE *fp = fopen(path, "w");
        if (!fp) return -1;
        const uint64_t sf = seed + (uint64_t)f;
        for (int32_t k = 0; k < perFile; k++) {
            const uint64_t b = (uint64_t)k * 8;
            int c = pick_contig(&g, mix64(sf, b));
            int32_t L;
            if (len_mode == 1) L = lenA + (int32_t)below(mix64(sf, b + 1), (int64_t)lenB - lenA + 1);
            else { int32_t e = exp800(mix64(sf, b + 1)); L = 200 + (e < 9800 ? e : 9800); }
            int64_t s = below(mix64(sf, b + 2), g.len[c] - L);
            fprintf(fp, "%s\t%lld\t%lld\tp%d\t%d\n", g.name[c], (long long)s, (long long)(s + L), k,
                    (int)(mix64(sf, b + 3) % 1001));
        }
        fclose(fp);
    }
    return 0;
}

static void radix_sort_keys(uint64_t *key, int32_t *pay, int64_t n)
{
    uint64_t *k2 = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n ? n : 1));
    int32_t *p2 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n ? n : 1));
    for (int pass = 0; pass < 4; pass++) {               /* 44 bits: contig(8) + start(32)... */
        const int sh = pass * 11;
        int64_t cnt[2049];
        memset(cnt, 0, sizeof cnt);
        for (int64_t i = 0; i < n; i++) cnt[((key[i] >> sh) & 2047) + 1]++;
        for (int i = 0; i < 2048; i++) cnt[i + 1] += cnt[i];
        for (int64_t i = 0; i < n; i++) {
            int64_t d = cnt[(key[i] >> sh) & 2047]++;
            k2[d] = key[i]; p2[d] = pay[i];
        }
        uint64_t *tk = key; key = k2; k2 = tk;
        int32_t *tp = pay; pay = p2; p2 = tp;
    }
    /* 4 passes: data is back in the caller's arrays */
    free(k2); free(p2);
}

int64_t igd_synth_queries(int64_t n, uint64_t seed, int genome, int32_t minLen, int32_t maxLen,
                          int sorted, int32_t unknown_every, int64_t extra_span,
                          int32_t *ichr, int32_t *qs, int32_t *qe)
{
    genome_t g; genome_init(&g, genome);
    uint64_t *key = sorted ? (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n ? n : 1)) : NULL;
    for (int64_t i = 0; i < n; i++) {
        const uint64_t b = (uint64_t)i * 4;
        int c = pick_contig(&g, mix64(seed, b));
        int32_t L = minLen + (int32_t)below(mix64(seed, b + 1), (int64_t)maxLen - minLen + 1);
        /* extra_span > 0 lets starts run past the contig end (queries beyond the last tile) */
        int64_t s = below(mix64(seed, b + 2), g.len[c] - L + extra_span);
        if (unknown_every > 0 && (i % unknown_every) == unknown_every - 1) c = -1;
        ichr[i] = c; qs[i] = (int32_t)s; qe[i] = (int32_t)(s + L);
        if (sorted) key[i] = ((uint64_t)(uint8_t)(c + 1) << 32) | (uint32_t)s;
    }
    if (sorted) {
        /* stable sort by (contig, start); the end travels as payload */
        radix_sort_keys(key, qe, n);
        for (int64_t i = 0; i < n; i++) {
            ichr[i] = (int32_t)(key[i] >> 32) - 1;
            qs[i] = (int32_t)(uint32_t)key[i];
        }
        free(key);
    }
    return n;
}

/* Positions [lo, hi) of the SORTED query set igd_synth_queries(n, seed, ..., sorted=1) would return,
 * without materialising the other n - (hi - lo) queries: BASELINE config 4 gives GPU r the r-th contiguous
 * slab of one position-sorted set of 10^8 queries.  Two generation passes (the generator is counter-based):
 * a histogram over the top bits of the sort key finds the key range that holds the slab, the second pass
 * keeps only queries in that range, which are then sorted exactly like the whole set (stable radix sort,
 * generation order among equal keys).  Returns hi - lo, or -1. */
int64_t igd_synth_queries_slab(int64_t n, uint64_t seed, int genome, int32_t minLen, int32_t maxLen,
                               int64_t lo, int64_t hi, int32_t *ichr, int32_t *qs, int32_t *qe)
{
    if (lo < 0 || hi > n || lo > hi) return -1;
    if (lo == hi) return 0;
    genome_t g; genome_init(&g, genome);
    const int SH = 16;                                     /* bucket = key >> 16: (contig+1) << 16 | start >> 16 */
    const int64_t nb = ((int64_t)(g.n + 1) << 16) + 1;
    int64_t *hist = (int64_t *)calloc((size_t)nb + 1, sizeof(int64_t));
    if (!hist) return -1;
    for (int64_t i = 0; i < n; i++) {
        const uint64_t b = (uint64_t)i * 4;
        const int c = pick_contig(&g, mix64(seed, b));
        const int32_t L = minLen + (int32_t)below(mix64(seed, b + 1), (int64_t)maxLen - minLen + 1);
        const int64_t s = below(mix64(seed, b + 2), g.len[c] - L);
        const uint64_t key = ((uint64_t)(uint8_t)(c + 1) << 32) | (uint32_t)s;
        hist[key >> SH]++;
    }
    int64_t b0 = 0, before = 0, run = 0;                   /* first bucket that reaches position lo */
    while (b0 < nb && run + hist[b0] <= lo) run += hist[b0++];
    before = run;
    int64_t b1 = b0;                                       /* last bucket needed for position hi-1 */
    while (b1 < nb && run + hist[b1] < hi) run += hist[b1++];
    int64_t m = 0;
    for (int64_t b = b0; b <= b1 && b < nb; b++) m += hist[b];
    free(hist);
    uint64_t *key = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(m ? m : 1));
    int32_t *end = (int32_t *)malloc(sizeof(int32_t) * (size_t)(m ? m : 1));
    if (!key || !end) { free(key); free(end); return -1; }
    int64_t k = 0;
    for (int64_t i = 0; i < n; i++) {
        const uint64_t b = (uint64_t)i * 4;
        const int c = pick_contig(&g, mix64(seed, b));
        const uint64_t kc = (uint64_t)(uint8_t)(c + 1) << 32;
        if ((int64_t)((kc | 0xFFFFFFFFull) >> SH) < b0 || (int64_t)(kc >> SH) > b1) continue;   /* contig outside the range */
        const int32_t L = minLen + (int32_t)below(mix64(seed, b + 1), (int64_t)maxLen - minLen + 1);
        const int64_t s = below(mix64(seed, b + 2), g.len[c] - L);
        const uint64_t ky = kc | (uint32_t)s;
        const int64_t bk = (int64_t)(ky >> SH);
        if (bk < b0 || bk > b1) continue;
        key[k] = ky; end[k] = (int32_t)(s + L); k++;
    }
    radix_sort_keys(key, end, k);
    for (int64_t i = lo; i < hi; i++) {
        const int64_t j = i - before;
        ichr[i - lo] = (int32_t)(key[j] >> 32) - 1;
        qs[i - lo] = (int32_t)(uint32_t)key[j];
        qe[i - lo] = end[j];
    }
    free(key); free(end);
    return hi - lo;
}

int igd_synth_write_bed(const char *path, int genome, int64_t n, const int32_t *ichr,
                        const int32_t *qs, const int32_t *qe)
{
    FILE *fp = fopen(path, "w");
    if (!fp) return -1;
    static char buf[1 << 20];
    setvbuf(fp, buf, _IOFBF, sizeof buf);
    for (int64_t i = 0; i < n; i++)
        fprintf(fp, "%s\t%d\t%d\n", igd_synth_contig_name(genome, ichr[i]), qs[i], qe[i]);
    fclose(fp);
    return 0;
}

#ifdef IGD_SYNTH_MAIN
static long long argll(int argc, char **argv, const char *flag, long long def)
{
    for (int i = 2; i + 1 < argc; i++)
        if (strcmp(argv[i], flag) == 0) return atoll(argv[i + 1]);
    return def;
}
static int has(int argc, char **argv, const char *flag)
{
    for (int i = 2; i < argc; i++)
        if (strcmp(argv[i], flag) == 0) return 1;
    return 0;
}
int main(int argc, char **argv)
{
    if (argc >= 3 && strcmp(argv[1], "db") == 0) {
        int small = has(argc, argv, "--small");
        return igd_synth_db(argv[2], (int32_t)argll(argc, argv, "--files", 1900),
                            (int32_t)argll(argc, argv, "--per-file", 26316),
                            (uint64_t)argll(argc, argv, "--seed", 1000),
                            (int32_t)argll(argc, argv, "--nbp-log", 14), small,
                            small ? 1 : 0, 50, 30000, has(argc, argv, "--clustered"),
                            has(argc, argv, "--gtype0") ? 0 : 1) ? 1 : 0;
    }
    if (argc >= 3 && strcmp(argv[1], "beds") == 0) {
        int small = has(argc, argv, "--small");
        return igd_synth_db_beds(argv[2], (int32_t)argll(argc, argv, "--files", 12),
                                 (int32_t)argll(argc, argv, "--per-file", 10000),
                                 (uint64_t)argll(argc, argv, "--seed", 1000), small, small ? 1 : 0,
                                 50, 30000) ? 1 : 0;
    }
    if (argc >= 3 && strcmp(argv[1], "queries") == 0) {
        int small = has(argc, argv, "--small");
        long long n = argll(argc, argv, "--n", 1000000);
        int32_t *c = malloc(4 * (size_t)n), *s = malloc(4 * (size_t)n), *e = malloc(4 * (size_t)n);
        igd_synth_queries(n, (uint64_t)argll(argc, argv, "--seed", 7), small,
                          (int32_t)argll(argc, argv, "--min-len", small ? 1 : 100),
                          (int32_t)argll(argc, argv, "--max-len", small ? 60000 : 1999),
                          !has(argc, argv, "--shuffled"),
                          (int32_t)argll(argc, argv, "--unknown-every", small ? 50 : 0),
                          argll(argc, argv, "--extra-span", small ? 100000 : 0), c, s, e);
        return igd_synth_write_bed(argv[2], small, n, c, s, e) ? 1 : 0;
    }
    fprintf(stderr,
            "usage: igd_synth db <out.igd> [--files F --per-file N --seed S --nbp-log B --small --clustered --gtype0]\n"
            "       igd_synth beds <out_dir> [--files F --per-file N --seed S --small]\n"
            "       igd_synth queries <out.bed> [--n Q --seed S --shuffled --small --min-len a --max-len b]\n");
    return 2;
}
#endif
