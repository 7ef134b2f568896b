/* igd_oracle.h -- CPU ORACLE for the IGD overlap-search hot path.
 *
 * >>> TEST INFRASTRUCTURE, NOT PRODUCT CODE. <<<
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load, link
 * or execute anything in oracle/.  The shipped path (igd_amd/csrc, include/) never
 * includes this header and never falls back to this code.
 *
 * What it is: a plain-C restatement of the reference algorithm of databio/IGD
 * (`/root/reference/src/igd_search.c`, `igd_base.c`), one function per reference
 * function, each citing the file:line it follows.  It keeps the reference's *structure*
 * (header tables only in RAM, one-tile cache, fseek+fread per tile change, 16/12-byte
 * AoS records, inline bisection + reverse linear scan, int64 hits[]), so that it can also
 * serve as a faithful single-thread CPU baseline ("port").
 *
 * Parity pin: the reference ships NO tests, golden vectors or fixtures for this path
 * (SURVEY.md section 4), so this oracle is pinned against OUTPUTS OF THE REFERENCE ITSELF:
 * oracle/_ref/igd (the real reference, compiled by oracle/Makefile from the sources where
 * they lie) run in the build container; its stdout is committed under tests/golden/ with
 * the generating script (tests/golden/make_golden.py), and tests/test_oracle_vs_ref.py
 * re-runs a differential fuzz whenever oracle/_ref/igd is present.
 *
 * Deliberate deviations from the reference (all are reference UB/crash, not behaviour):
 *   - qs <= -nbp makes n1 negative and the reference indexes nCnt[ichr][n1] out of
 *     bounds (src/igd_search.c:459,466); the oracle returns 0 hits for such a query.
 *   - byte offsets are computed in 64 bit (the reference multiplies in int32,
 *     src/igd_base.c:299-302).
 *   - long paths do not overflow (reference: char fname[64], src/igd_base.h:99).
 */
#ifndef IGD_ORACLE_H
#define IGD_ORACLE_H
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_db orc_db;

/* One emitted overlap of the `-f` path (src/igd_search.c:575-579, :608-612). */
typedef struct { int32_t idx, start, end; } orc_hit;

/* Work statistics, by-product of the scan: the terms of SURVEY.md section 8(d). */
typedef struct {
    int64_t queries;   /* queries that reached the tile logic (known contig)          */
    int64_t pairs;     /* (query,tile) pairs with cnt>0 && qe > g[0].start            */
    int64_t S;         /* sum of scan lengths  max(0, tL-tS+1)                         */
    int64_t H;         /* hits                                                        */
    int64_t B;         /* sum of ceil(log2(cnt+1)) bisection probes                   */
} orc_stats;

/* tile-visiting rules, SURVEY Appendix B.2 */
enum { ORC_RULE_NEST = 0, ORC_RULE_FLAT = 1 };

/* ---- database ------------------------------------------------------------------- */
orc_db *orc_open(const char *igd_path);          /* get_igdinfo + get_fileinfo         */
void    orc_close(orc_db *db);
void    orc_preload(orc_db *db);                 /* optional: whole data region in RAM */
int32_t orc_nfiles(const orc_db *db);
int32_t orc_nctg(const orc_db *db);
int32_t orc_nbp(const orc_db *db);
int32_t orc_gtype(const orc_db *db);
int32_t orc_ntile(const orc_db *db, int32_t ichr);
int32_t orc_ncnt(const orc_db *db, int32_t ichr, int32_t j);
const char *orc_ctg_name(const orc_db *db, int32_t ichr);
const char *orc_file_name(const orc_db *db, int32_t i);
int32_t orc_file_nr(const orc_db *db, int32_t i);
int32_t orc_get_id(const orc_db *db, const char *chrm);      /* src/igd_base.c:325-331 */
const orc_stats *orc_get_stats(const orc_db *db);
void    orc_reset_stats(orc_db *db);

/* ---- parsing -------------------------------------------------------------------- */
/* src/igd_base.c:53-72; mutates `line`; returns contig pointer or NULL (line skipped) */
char   *orc_parse_bed(char *line, int32_t *st, int32_t *en);
/* Reads a BED / BED.gz like the loop at src/igd_search.c:708-714 and returns the accepted
 * queries with a contig known to the db (others are skipped exactly as get_id<0 does).
 * Arrays are malloc'd; caller frees.  Returns count or -1 if the file cannot be opened. */
int64_t orc_read_queries(const orc_db *db, const char *qfile,
                         int32_t **ichr, int32_t **qs, int32_t **qe);

/* ---- per-query kernels (gType dispatch inside, like src/igd_search.c:1023-1053) --- */
int32_t orc_get_overlaps  (orc_db *db, const char *chrm, int32_t qs, int32_t qe, int64_t *hits);            /* :454-534 / :30-112 */
int32_t orc_get_overlaps_v(orc_db *db, const char *chrm, int32_t qs, int32_t qe, int32_t v, int64_t *hits); /* :623-694 */
/* `-f`: prints like :537-620 / :114-200 when out!=NULL; appends to (*buf) when buf!=NULL */
int32_t orc_get_overlaps_f(orc_db *db, const char *chrm, int32_t qs, int32_t qe, FILE *out);

/* ---- query-file loops ------------------------------------------------------------- */
int64_t orc_getOverlaps  (orc_db *db, const char *qfile, int64_t *hits);             /* :696-719 / :202-225 */
int64_t orc_getOverlaps_v(orc_db *db, const char *qfile, int64_t *hits, int32_t v);  /* :746-769 */
int64_t orc_getOverlaps_f(orc_db *db, const char *qfile, FILE *out);                 /* :721-744 / :227-250 */

/* ---- array batches (what the GPU engine is compared against) ---------------------- */
/* CLI dispatch (src/igd_search.c:1023-1030): gType 0 -> rule NEST, v ignored;
 * gType 1 && v>0 -> rule FLAT with value filter; else rule NEST.  hits += counts.
 * Returns the number of overlaps found (all modes; unlike the reference's nols). */
int64_t orc_search_batch(orc_db *db, const int32_t *ichr, const int32_t *qs, const int32_t *qe,
                         int64_t nq, int32_t v, int64_t *hits);
/* `-f` on arrays: qoff[0..nq] (exclusive scan of per-query counts), hits in reference
 * order.  Pass out=NULL/cap=0 to only count.  Returns total overlaps. */
int64_t orc_enumerate_batch(orc_db *db, const int32_t *ichr, const int32_t *qs, const int32_t *qe,
                            int64_t nq, int64_t *qoff, orc_hit *out, int64_t cap);

/* ---- dataset x dataset hit map (`-m`), SURVEY 8f row f3 ------------------------------ */
/* getMap src/igd_search.c:772-826 (use_v=0) / getMap_v :829-886 (value > v, strict):
 * hitmap is nFiles x nFiles, row-major, ADDED to.  Returns the number of pairs counted.
 * progress!=NULL receives the reference's progress lines ("%i\n" every 1000 tiles, :783-784). */
int64_t orc_getMap(orc_db *db, int use_v, int32_t v, uint32_t *hitmap, FILE *progress);

/* ---- Seqpare similarity (`search -q f.bed -s`), SURVEY 8f row f4 -------------------------- */
/* seqOverlaps src/igd_search.c:354-451 over seq_overlaps :253-352: sm[nFiles].  0 / -1 (file). */
int orc_seqOverlaps(orc_db *db, const char *qfile, double *sm);
/* seq_overlaps :253-352 for one interval: {idx_t, idx_g, idx_f, bits of sm} x min(n, cap) into out; returns n */
int64_t orc_seq_overlaps(orc_db *db, const char *chrm, int32_t qs, int32_t qe, int32_t *out, int64_t cap);

/* ---- `igd create`, SURVEY 8f row f4 (igd_oracle_create.c) ------------------------------ */
enum { ORC_CREATE_GLOB = 0, ORC_CREATE_LIST = 1, ORC_CREATE_GTYPE0 = 2, ORC_CREATE_BED4 = 3 };
/* src/igd_create.c:25-433 + igd_add/igd_saveT/igd_save (src/igd_base.c:118-169, :333-461):
 * writes <opath><name>.igd and <opath><name>_index.tsv; `out` receives the reference's stdout text. */
int  orc_create(const char *ipath, const char *opath, const char *name, int32_t nbp, int mode, FILE *out);
int  orc_igd_create(int argc, char **argv, FILE *out);            /* src/igd_create.c:436-501 */
/* radix_sort_intv (src/igd_base.h:196-249) applied to n (start, payload) pairs in place: the
 * exact, unstable order the reference leaves records of one tile in. */
void orc_tile_sort(int32_t *key, int32_t *src, int64_t n);

/* ---- `igd search` driver (stdout text identical to src/igd_search.c:889-1079) ----- */
int orc_igd_search(int argc, char **argv, FILE *out);

#ifdef __cplusplus
}
#endif
#endif
