/* igd_oracle.c -- CPU ORACLE (plain C restatement of the reference hot path).
 *
 * >>> TEST INFRASTRUCTURE, NOT PRODUCT CODE -- see igd_oracle.h for the rules. <<<
 *
 * Reference = /root/reference/src (databio/IGD v0.1.1).  Every function names the
 * reference lines it restates.  The gType-1 (16-byte) and gType-0 (12-byte) twins of the
 * reference are folded into one body parameterised by the record stride, since the twins
 * differ only in the struct they index (src/igd_search.c:30-112 vs :454-534).
 */
#define _GNU_SOURCE
#include "igd_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* ---------------------------------------------------------------------------------- */
/* record access: gdata_t {idx,start,end,value} src/igd_base.h:41-46,                  */
/*                gdata0_t{idx,start,end}       src/igd_base.h:48-52                   */
#define R_IDX(g, rs, i)   ((g)[(size_t)(i) * (rs) + 0])
#define R_START(g, rs, i) ((g)[(size_t)(i) * (rs) + 1])
#define R_END(g, rs, i)   ((g)[(size_t)(i) * (rs) + 2])
#define R_VALUE(g, rs, i) ((g)[(size_t)(i) * (rs) + 3])

struct orc_db {
    int32_t nbp, gType, nCtg, nFiles;
    int32_t rs;                 /* int32 words per record: 4 or 3                       */
    int32_t *nTile;             /* [nCtg]                                              */
    int32_t **nCnt;             /* [nCtg][nTile]                                       */
    int64_t **tIdx;             /* byte offset of each tile in the .igd                */
    char   (*cName)[40];
    char  **fileName;
    int32_t *fileNr;
    double  *fileMd;
    FILE   *fp;                 /* stays open for tile reads (reference: global fP)    */
    /* one-tile cache (reference globals gData/preIdx/preChr, src/igd.c:16-18)         */
    int32_t *tile;
    size_t   tile_cap;
    int32_t  preIdx, preChr;
    /* optional whole-file image (test convenience only)                               */
    int32_t *image;
    int64_t  image_base;        /* byte offset of image[0] in the file                 */
    /* contig dictionary: open addressing, exact match                                 */
    int32_t *dict;
    int32_t  dict_cap;
    orc_stats st;
    /* `-f` sink for orc_enumerate_batch                                               */
    orc_hit *sink;
    int64_t  sink_n, sink_cap;
    int      sink_on;
};

/* ------------------------------ contig dictionary ---------------------------------- */
static uint32_t str_hash(const char *s)
{
    uint32_t h = 2166136261u;
    for (; *s; ++s) h = (h ^ (unsigned char)*s) * 16777619u;
    return h;
}

static void dict_build(orc_db *db)
{
    int32_t cap = 16;
    while (cap < 4 * db->nCtg) cap <<= 1;
    db->dict_cap = cap;
    db->dict = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
    for (int32_t i = 0; i < cap; i++) db->dict[i] = -1;
    for (int32_t c = 0; c < db->nCtg; c++) {
        uint32_t p = str_hash(db->cName[c]) & (uint32_t)(cap - 1);
        /* a repeated name keeps one slot and the LATER index wins, as kh_put followed by
         * kh_val(h,k)=i does (src/igd_base.c:315-320) */
        while (db->dict[p] >= 0 && strcmp(db->cName[db->dict[p]], db->cName[c]) != 0)
            p = (p + 1) & (uint32_t)(cap - 1);
        db->dict[p] = c;
    }
}

/* src/igd_base.c:325-331 get_id: exact, case-sensitive; -1 when absent */
int32_t orc_get_id(const orc_db *db, const char *chrm)
{
    uint32_t p = str_hash(chrm) & (uint32_t)(db->dict_cap - 1);
    while (db->dict[p] >= 0) {
        if (strcmp(db->cName[db->dict[p]], chrm) == 0) return db->dict[p];
        p = (p + 1) & (uint32_t)(db->dict_cap - 1);
    }
    return -1;
}

/* ------------------------------ loaders -------------------------------------------- */
/* src/igd_base.c:235-267 get_fileinfo: skip the header line; every further fgets(1024)
 * line is one dataset: strtok on tabs -> index, fileName, nr=atol, md=(double)atol */
static int load_fileinfo(orc_db *db, const char *tsv)
{
    FILE *fp = fopen(tsv, "r");
    if (!fp) return -1;
    char buf[1024];
    int n = 0;
    if (!fgets(buf, sizeof buf, fp)) { fclose(fp); return -1; }
    while (fgets(buf, sizeof buf, fp)) n++;
    db->nFiles = n;
    db->fileName = (char **)calloc((size_t)n + 1, sizeof(char *));
    db->fileNr = (int32_t *)calloc((size_t)n + 1, sizeof(int32_t));
    db->fileMd = (double *)calloc((size_t)n + 1, sizeof(double));
    fseek(fp, 0, SEEK_SET);
    if (!fgets(buf, sizeof buf, fp)) { fclose(fp); return -1; }
    int i = 0;
    while (i < n && fgets(buf, sizeof buf, fp)) {
        char *save = NULL;
        char *f0 = strtok_r(buf, "\t", &save);
        char *f1 = strtok_r(NULL, "\t", &save);
        char *f2 = strtok_r(NULL, "\t", &save);
        char *f3 = strtok_r(NULL, "\t", &save);
        (void)f0;
        db->fileName[i] = strdup(f1 ? f1 : "");
        db->fileNr[i] = f2 ? (int32_t)atol(f2) : 0;
        db->fileMd[i] = f3 ? (double)atol(f3) : 0.0;
        i++;
    }
    fclose(fp);
    return 0;
}

/* src/igd_base.c:269-323 get_igdinfo (+ the _index.tsv naming of src/igd_search.c:918-923) */
orc_db *orc_open(const char *igd_path)
{
    FILE *fp = fopen(igd_path, "rb");
    if (!fp) return NULL;
    orc_db *db = (orc_db *)calloc(1, sizeof *db);
    int32_t h[3];
    if (fread(h, sizeof(int32_t), 3, fp) != 3) goto fail;
    db->nbp = h[0]; db->gType = h[1]; db->nCtg = h[2];
    db->rs = db->gType == 0 ? 3 : 4;
    if (db->nbp <= 0 || db->nCtg < 0) goto fail;
    int32_t m = db->nCtg;
    db->nTile = (int32_t *)malloc(sizeof(int32_t) * (size_t)(m + 1));
    if (m && fread(db->nTile, sizeof(int32_t), (size_t)m, fp) != (size_t)m) goto fail;
    int64_t loc = 12 + 44 * (int64_t)m;                    /* :288-289 header size     */
    for (int32_t i = 0; i < m; i++) loc += 4 * (int64_t)db->nTile[i];
    db->nCnt = (int32_t **)calloc((size_t)m + 1, sizeof(int32_t *));
    db->tIdx = (int64_t **)calloc((size_t)m + 1, sizeof(int64_t *));
    int64_t recbytes = 4 * (int64_t)db->rs;
    for (int32_t i = 0; i < m; i++) {                      /* :290-303                 */
        int32_t k = db->nTile[i];
        db->nCnt[i] = (int32_t *)calloc((size_t)k + 1, sizeof(int32_t));
        db->tIdx[i] = (int64_t *)calloc((size_t)k + 1, sizeof(int64_t));
        if (k && fread(db->nCnt[i], sizeof(int32_t), (size_t)k, fp) != (size_t)k) goto fail;
        for (int32_t j = 0; j < k; j++) {
            db->tIdx[i][j] = loc;
            loc += (int64_t)db->nCnt[i][j] * recbytes;
        }
    }
    db->cName = (char (*)[40])calloc((size_t)m + 1, 40);
    for (int32_t i = 0; i < m; i++) {                      /* :305-309                 */
        if (fread(db->cName[i], 40, 1, fp) != 1) goto fail;
        db->cName[i][39] = '\0';
    }
    db->fp = fp;
    db->preIdx = -8; db->preChr = -6;
    dict_build(db);
    {   /* "<db minus last .ext>_index.tsv" */
        size_t L = strlen(igd_path);
        char *tsv = (char *)malloc(L + 16);
        strcpy(tsv, igd_path);
        char *dot = strrchr(tsv, '.');
        if (dot) *dot = '\0';
        strcat(tsv, "_index.tsv");
        int rc = load_fileinfo(db, tsv);
        free(tsv);
        if (rc != 0) { orc_close(db); return NULL; }
    }
    return db;
fail:
    fclose(fp);
    db->fp = NULL;
    orc_close(db);
    return NULL;
}

void orc_close(orc_db *db)
{
    if (!db) return;
    if (db->fp) fclose(db->fp);
    for (int32_t i = 0; i < db->nCtg; i++) {
        if (db->nCnt) free(db->nCnt[i]);
        if (db->tIdx) free(db->tIdx[i]);
    }
    for (int32_t i = 0; i < db->nFiles; i++) free(db->fileName[i]);
    free(db->fileName); free(db->fileNr); free(db->fileMd);
    free(db->nCnt); free(db->tIdx); free(db->nTile); free(db->cName);
    free(db->tile); free(db->image); free(db->dict); free(db->sink);
    free(db);
}

void orc_preload(orc_db *db)
{
    if (db->image || db->nCtg == 0) return;
    int64_t base = db->tIdx[0][0];
    fseek(db->fp, 0, SEEK_END);
    int64_t end = ftell(db->fp);
    if (end <= base) return;
    db->image = (int32_t *)malloc((size_t)(end - base));
    fseek(db->fp, base, SEEK_SET);
    if (fread(db->image, 1, (size_t)(end - base), db->fp) != (size_t)(end - base)) {
        free(db->image); db->image = NULL; return;
    }
    db->image_base = base;
}

int32_t orc_nfiles(const orc_db *db) { return db->nFiles; }
int32_t orc_nctg(const orc_db *db) { return db->nCtg; }
int32_t orc_nbp(const orc_db *db) { return db->nbp; }
int32_t orc_gtype(const orc_db *db) { return db->gType; }
int32_t orc_ntile(const orc_db *db, int32_t c) { return db->nTile[c]; }
int32_t orc_ncnt(const orc_db *db, int32_t c, int32_t j) { return db->nCnt[c][j]; }
const char *orc_ctg_name(const orc_db *db, int32_t c) { return db->cName[c]; }
const char *orc_file_name(const orc_db *db, int32_t i) { return db->fileName[i]; }
int32_t orc_file_nr(const orc_db *db, int32_t i) { return db->fileNr[i]; }
const orc_stats *orc_get_stats(const orc_db *db) { return &db->st; }
void orc_reset_stats(orc_db *db) { memset(&db->st, 0, sizeof db->st); }

/* ------------------------------ tile fetch ----------------------------------------- */
/* The reference re-reads a tile only when (contig,tile) changes: fseek + free + malloc +
 * fread, src/igd_search.c:469-476 (and :501-508).  Same here (buffer grown, not freed). */
static const int32_t *fetch_tile(orc_db *db, int32_t ichr, int32_t j, int32_t cnt)
{
    if (db->image)
        return db->image + (db->tIdx[ichr][j] - db->image_base) / 4;
    if (j != db->preIdx || ichr != db->preChr) {
        size_t need = (size_t)cnt * (size_t)db->rs;
        if (need > db->tile_cap) {
            free(db->tile);
            db->tile = (int32_t *)malloc(need * sizeof(int32_t));
            db->tile_cap = need;
        }
        fseek(db->fp, db->tIdx[ichr][j], SEEK_SET);
        if (fread(db->tile, sizeof(int32_t), need, db->fp) != need) memset(db->tile, 0, need * 4);
        db->preIdx = j;
        db->preChr = ichr;
    }
    return db->tile;
}

/* ------------------------------ searches inside one tile --------------------------- */
/* inline bisection of get_overlaps, src/igd_search.c:479-487 (again :512-520):
 * precondition qe > g[0].start; returns the LAST index with start < qe */
static int32_t bisect_inline(const int32_t *g, int rs, int32_t cnt, int32_t qe)
{
    int32_t tL = 0, tR = cnt - 1;
    while (tL < tR - 1) {
        int32_t tM = (tL + tR) / 2;
        if (R_START(g, rs, tM) < qe) tL = tM; else tR = tM;
    }
    if (R_START(g, rs, tR) < qe) tL = tR;
    return tL;
}

/* bSearch, src/igd_base.c:74-94: last index in [t0,tc] with start < qe, -1 if none */
static int32_t bsearch_last(const int32_t *g, int rs, int32_t t0, int32_t tc, int32_t qe)
{
    int32_t tL = t0, tR = tc;
    if (R_START(g, rs, tR) < qe) return tR;
    if (R_START(g, rs, tL) >= qe) return -1;
    while (tL < tR - 1) {
        int32_t tM = (tL + tR) / 2;
        if (R_START(g, rs, tM) >= qe) tR = tM - 1; else tL = tM;
    }
    if (R_START(g, rs, tR) < qe) return tR;
    if (R_START(g, rs, tL) < qe) return tL;
    return -1;
}

static int ceil_log2_plus1(int32_t cnt)       /* ceil(log2(cnt+1)) */
{
    int b = 0;
    while (((int64_t)1 << b) < (int64_t)cnt + 1) b++;
    return b;
}

static void sink_push(orc_db *db, int32_t idx, int32_t s, int32_t e)
{
    if (db->sink_n == db->sink_cap) {
        db->sink_cap = db->sink_cap ? db->sink_cap * 2 : 1024;
        db->sink = (orc_hit *)realloc(db->sink, sizeof(orc_hit) * (size_t)db->sink_cap);
    }
    db->sink[db->sink_n].idx = idx;
    db->sink[db->sink_n].start = s;
    db->sink[db->sink_n].end = e;
    db->sink_n++;
}

/* One visited tile.  `first`: this is tile n1 (scan down to index 0); otherwise records
 * with start < bd are skipped (tS walk, src/igd_search.c:510-511 / :679-680).
 * use_v: the get_overlaps_v flavour (:645-656, :673-686): tE by back-walk when cnt<16,
 * else bSearch; extra predicate value >= v.
 * hits!=NULL: count into hits[idx];  out/sink: the `-f` flavour (:575-579, :608-612). */
static int32_t visit_tile(orc_db *db, int32_t ichr, int32_t j, int first, int32_t bd,
                          int32_t qs, int32_t qe, int use_v, int32_t v,
                          int64_t *hits, FILE *out, int32_t *nprint)
{
    int32_t cnt = db->nCnt[ichr][j];
    if (cnt <= 0) return 0;
    const int rs = db->rs;
    const int32_t *g = fetch_tile(db, ichr, j, cnt);
    if (!(qe > R_START(g, rs, 0))) return 0;                     /* :478 / :509 / :644  */
    int32_t tS = 0;
    if (!first)
        while (tS < cnt && R_START(g, rs, tS) < bd) tS++;
    int32_t tE;
    if (use_v) {
        if (cnt < 16) {                                          /* :645-648            */
            tE = cnt - 1;
            while (R_START(g, rs, tE) >= qe) tE--;
        } else
            tE = bsearch_last(g, rs, 0, cnt - 1, qe);            /* :650                */
    } else
        tE = bisect_inline(g, rs, cnt, qe);
    int32_t n = 0;
    for (int32_t i = tE; i >= tS; i--) {
        if (R_END(g, rs, i) > qs && (!use_v || R_VALUE(g, rs, i) >= v)) {
            n++;
            if (hits) hits[R_IDX(g, rs, i)]++;
            if (out)
                fprintf(out, "%i\t %i\t %i\t %s\n", (*nprint)++, R_START(g, rs, i),
                        R_END(g, rs, i), db->fileName[R_IDX(g, rs, i)]);
            if (db->sink_on) sink_push(db, R_IDX(g, rs, i), R_START(g, rs, i), R_END(g, rs, i));
        }
    }
    db->st.pairs++;
    db->st.S += (tE - tS + 1) > 0 ? (tE - tS + 1) : 0;
    db->st.H += n;
    db->st.B += ceil_log2_plus1(cnt);
    return n;
}

/* Tile walk shared by every per-query kernel.
 *  rule NEST: src/igd_search.c:454-534 get_overlaps (also :30-112, :114-200, :537-620):
 *             the loop over n1+1..n2 sits INSIDE `if(nCnt[n1]>0)` (:468 ... :532).
 *  rule FLAT: src/igd_search.c:623-694 get_overlaps_v: that loop is OUTSIDE (:659).
 * Returns the number of overlaps found in this query. */
static int32_t walk_tiles(orc_db *db, int32_t ichr, int32_t qs, int32_t qe, int rule,
                          int use_v, int32_t v, int64_t *hits, FILE *out,
                          const char *chrm_for_print)
{
    const int32_t nbp = db->nbp;
    int32_t n1 = qs / nbp, n2 = (qe - 1) / nbp;                  /* :459                */
    int32_t mTile = db->nTile[ichr] - 1;
    if (n1 > mTile) return 0;                                    /* :461-462            */
    if (n1 < 0) return 0;            /* reference UB (negative index); see header       */
    db->st.queries++;
    if (out && chrm_for_print)                                   /* :548 / :126         */
        fprintf(out, "Query %s, %i, %i: \n", chrm_for_print, qs, qe);
    if (n2 > mTile) n2 = mTile;                                  /* :464                */
    int32_t total = 0, nprint = 0;
    if (rule == ORC_RULE_NEST && db->nCnt[ichr][n1] <= 0) return 0;
    total += visit_tile(db, ichr, n1, 1, 0, qs, qe, use_v, v, hits, out, &nprint);
    if (n2 > n1) {
        int32_t bd = nbp * (n1 + 1);                             /* :496 / :660         */
        for (int32_t j = n1 + 1; j <= n2; j++) {
            total += visit_tile(db, ichr, j, 0, bd, qs, qe, use_v, v, hits, out, &nprint);
            bd = (int32_t)((uint32_t)bd + (uint32_t)nbp);        /* :529 (wraps unused) */
        }
    }
    return total;
}

/* ------------------------------ per-query kernels ---------------------------------- */
/* get_overlaps :454-534 (gType 1) / get_overlaps0 :30-112 (gType 0).
 * NOTE: the reference returns nols which it never increments here (always 0). */
int32_t orc_get_overlaps(orc_db *db, const char *chrm, int32_t qs, int32_t qe, int64_t *hits)
{
    int32_t ichr = orc_get_id(db, chrm);
    if (ichr < 0) return 0;
    walk_tiles(db, ichr, qs, qe, ORC_RULE_NEST, 0, 0, hits, NULL, NULL);
    return 0;
}

/* get_overlaps_v :623-694; returns the real count (nols++ at :653,:683) */
int32_t orc_get_overlaps_v(orc_db *db, const char *chrm, int32_t qs, int32_t qe, int32_t v,
                           int64_t *hits)
{
    int32_t ichr = orc_get_id(db, chrm);
    if (ichr < 0) return 0;
    return walk_tiles(db, ichr, qs, qe, ORC_RULE_FLAT, 1, v, hits, NULL, NULL);
}

/* get_overlaps_f1 :537-620 / get_overlaps_f0 :114-200 */
int32_t orc_get_overlaps_f(orc_db *db, const char *chrm, int32_t qs, int32_t qe, FILE *out)
{
    int32_t ichr = orc_get_id(db, chrm);
    if (ichr < 0) return 0;
    return walk_tiles(db, ichr, qs, qe, ORC_RULE_NEST, 0, 0, NULL, out, chrm);
}

/* ------------------------------ parsing -------------------------------------------- */
/* parse_bed src/igd_base.c:53-72 */
char *orc_parse_bed(char *s, int32_t *st_, int32_t *en_)
{
    char *p = s, *q = s, *ctg = NULL;
    int32_t i = 0, st = -1, en = -1;
    for (;; ++q) {
        if (*q == '\t' || *q == '\0') {
            int c = *q;
            *q = '\0';
            if (i == 0) ctg = p;
            else if (i == 1) st = (int32_t)atol(p);
            else if (i == 2) en = (int32_t)atol(p);
            ++i;
            p = q + 1;
            if (c == '\0') break;
        }
    }
    *st_ = st; *en_ = en;
    if (i >= 3 && ctg[0] == 'c' && ctg[1] == 'h' && ctg[2] == 'r' && strlen(ctg) < 40 && en > 0)
        return ctg;
    return NULL;
}

/* line reader with the behaviour of ks_getuntil(KS_SEP_LINE), src/kseq.h:82-130:
 * '\n'-separated, transparently gunzips, drops ONE trailing '\r' when the line is
 * longer than one char (:127), last line needs no newline. */
typedef struct {
    gzFile f;
    char *buf; int begin, end, eof;
    char *line; size_t len, cap;
} lreader;

static int lr_open(lreader *r, const char *path)
{
    memset(r, 0, sizeof *r);
    r->f = gzopen(path, "r");
    if (!r->f) return -1;
    r->buf = (char *)malloc(0x10000);
    return 0;
}
static void lr_close(lreader *r) { gzclose(r->f); free(r->buf); free(r->line); }
static int lr_next(lreader *r)
{
    r->len = 0;
    if (r->begin >= r->end && r->eof) return -1;
    for (;;) {
        if (r->begin >= r->end) {
            if (r->eof) break;
            r->begin = 0;
            r->end = gzread(r->f, r->buf, 0x10000);
            if (r->end < 0x10000) r->eof = 1;
            if (r->end <= 0) { r->end = 0; break; }
        }
        int i = r->begin;
        while (i < r->end && r->buf[i] != '\n') i++;
        size_t add = (size_t)(i - r->begin);
        if (r->cap < r->len + add + 1) {
            r->cap = (r->len + add + 1) * 2;
            r->line = (char *)realloc(r->line, r->cap);
        }
        memcpy(r->line + r->len, r->buf + r->begin, add);
        r->len += add;
        r->begin = i + 1;
        if (i < r->end) break;
    }
    if (!r->line) { r->cap = 16; r->line = (char *)calloc(1, r->cap); }
    if (r->len > 1 && r->line[r->len - 1] == '\r') r->len--;
    r->line[r->len] = '\0';
    return (int)r->len;
}

int64_t orc_read_queries(const orc_db *db, const char *qfile, int32_t **ichr_, int32_t **qs_,
                         int32_t **qe_)
{
    lreader r;
    if (lr_open(&r, qfile) != 0) return -1;
    int64_t n = 0, cap = 1024;
    int32_t *c = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
    int32_t *s = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
    int32_t *e = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
    while (lr_next(&r) >= 0) {
        int32_t st, en;
        char *chrm = orc_parse_bed(r.line, &st, &en);
        if (!chrm) continue;
        int32_t id = orc_get_id(db, chrm);
        if (id < 0) continue;
        if (n == cap) {
            cap *= 2;
            c = (int32_t *)realloc(c, sizeof(int32_t) * (size_t)cap);
            s = (int32_t *)realloc(s, sizeof(int32_t) * (size_t)cap);
            e = (int32_t *)realloc(e, sizeof(int32_t) * (size_t)cap);
        }
        c[n] = id; s[n] = st; e[n] = en; n++;
    }
    lr_close(&r);
    *ichr_ = c; *qs_ = s; *qe_ = e;
    return n;
}

/* ------------------------------ query-file loops ----------------------------------- */
/* getOverlaps :696-719 / getOverlaps0 :202-225: invalidate the cache (:707), then per
 * accepted line get_overlaps; returns the sum of the per-call returns (i.e. 0). */
int64_t orc_getOverlaps(orc_db *db, const char *qfile, int64_t *hits)
{
    lreader r;
    if (lr_open(&r, qfile) != 0) return 0;
    int64_t ols = 0;
    db->preChr = -6; db->preIdx = -8;
    while (lr_next(&r) >= 0) {
        int32_t st, en;
        char *chrm = orc_parse_bed(r.line, &st, &en);
        if (chrm) ols += orc_get_overlaps(db, chrm, st, en, hits);
    }
    lr_close(&r);
    return ols;
}

/* getOverlaps_v :746-769 */
int64_t orc_getOverlaps_v(orc_db *db, const char *qfile, int64_t *hits, int32_t v)
{
    lreader r;
    if (lr_open(&r, qfile) != 0) return 0;
    int64_t ols = 0;
    db->preChr = -6; db->preIdx = -8;
    while (lr_next(&r) >= 0) {
        int32_t st, en;
        char *chrm = orc_parse_bed(r.line, &st, &en);
        if (chrm) ols += orc_get_overlaps_v(db, chrm, st, en, v, hits);
    }
    lr_close(&r);
    return ols;
}

/* getOverlaps_f1 :721-744 / getOverlaps_f0 :227-250 */
int64_t orc_getOverlaps_f(orc_db *db, const char *qfile, FILE *out)
{
    lreader r;
    if (lr_open(&r, qfile) != 0) return 0;
    int64_t ols = 0;
    db->preChr = -6; db->preIdx = -8;
    while (lr_next(&r) >= 0) {
        int32_t st, en;
        char *chrm = orc_parse_bed(r.line, &st, &en);
        if (chrm) ols += orc_get_overlaps_f(db, chrm, st, en, out);
    }
    lr_close(&r);
    return ols;
}

/* ------------------------------ array batches -------------------------------------- */
int64_t orc_search_batch(orc_db *db, const int32_t *ichr, const int32_t *qs, const int32_t *qe,
                         int64_t nq, int32_t v, int64_t *hits)
{
    int use_v = (db->gType != 0 && v > 0);                       /* :1024-1029          */
    int rule = use_v ? ORC_RULE_FLAT : ORC_RULE_NEST;
    int64_t total = 0;
    for (int64_t i = 0; i < nq; i++) {
        if (ichr[i] < 0 || ichr[i] >= db->nCtg) continue;
        total += walk_tiles(db, ichr[i], qs[i], qe[i], rule, use_v, v, hits, NULL, NULL);
    }
    return total;
}

int64_t orc_enumerate_batch(orc_db *db, const int32_t *ichr, const int32_t *qs, const int32_t *qe,
                            int64_t nq, int64_t *qoff, orc_hit *out, int64_t cap)
{
    db->sink_on = 1;
    db->sink_n = 0;
    for (int64_t i = 0; i < nq; i++) {
        if (qoff) qoff[i] = db->sink_n;
        if (ichr[i] < 0 || ichr[i] >= db->nCtg) continue;
        walk_tiles(db, ichr[i], qs[i], qe[i], ORC_RULE_NEST, 0, 0, NULL, NULL, NULL);
    }
    if (qoff) qoff[nq] = db->sink_n;
    db->sink_on = 0;
    int64_t n = db->sink_n;
    if (out) memcpy(out, db->sink, sizeof(orc_hit) * (size_t)(n < cap ? n : cap));
    return n;
}

/* ------------------------------ hit map (-m) ----------------------------------------- */
/* getMap :772-826 / getMap_v :829-886.  Tile by tile, every record j of the tile is a query
 * against the tile itself: partners i with start_i < end_j (index from the <16 back-walk or
 * bSearch over [tS, cnt-1], :806-810), scanned downwards while the running maximum of ends
 * maxE[i] > start_j (:811), counted when end_i > start_j [and both values > v].  A record that
 * starts before the tile (qs < bd) only pairs with records that start inside it (:803-804). */
int64_t orc_getMap(orc_db *db, int use_v, int32_t v, uint32_t *hitmap, FILE *progress)
{
    const int rs = db->rs;
    int64_t nols = 0;
    int m = 0;
    for (int32_t ichr = 0; ichr < db->nCtg; ichr++) {
        for (int32_t n1 = 0; n1 < db->nTile[ichr]; n1++) {
            const int32_t bd = (int32_t)((uint32_t)db->nbp * (uint32_t)n1);
            const int32_t cnt = db->nCnt[ichr][n1];
            m++;
            if (progress && m % 1000 == 0) fprintf(progress, "%i\n", m);
            if (cnt <= 0) continue;
            db->preIdx = -8; db->preChr = -6;                    /* the reference re-reads every tile */
            const int32_t *g = fetch_tile(db, ichr, n1, cnt);
            int32_t *maxE = (int32_t *)malloc(sizeof(int32_t) * (size_t)cnt);
            int32_t tmax = R_END(g, rs, 0);
            for (int32_t i = 0; i < cnt; i++) {
                if (R_END(g, rs, i) > tmax) tmax = R_END(g, rs, i);
                maxE[i] = tmax;
            }
            for (int32_t j = 0; j < cnt; j++) {
                if (use_v && !(R_VALUE(g, rs, j) > v)) continue;
                const int32_t qe = R_END(g, rs, j), qs = R_START(g, rs, j);
                if (!(qe > R_START(g, rs, 0))) continue;
                const int32_t jj = R_IDX(g, rs, j);
                int32_t tS = 0;
                if (qs < bd)
                    while (tS < cnt && R_START(g, rs, tS) < bd) tS++;
                int32_t i;
                if (cnt < 16) {
                    i = cnt - 1;
                    while (R_START(g, rs, i) >= qe) i--;
                } else
                    i = tS <= cnt - 1 ? bsearch_last(g, rs, tS, cnt - 1, qe) : -1;
                while (i >= tS && maxE[i] > qs) {
                    if (R_END(g, rs, i) > qs && (!use_v || R_VALUE(g, rs, i) > v)) {
                        nols++;
                        hitmap[(size_t)jj * (size_t)db->nFiles + (size_t)R_IDX(g, rs, i)]++;
                    }
                    i--;
                }
            }
            free(maxE);
        }
    }
    return nols;
}

/* ------------------------------ `igd search` driver -------------------------------- */
/* ------------------------------ Seqpare (`search -q f.bed -s`) ---------------------------
 * SURVEY 8f row f4.  seq_overlaps src/igd_search.c:253-352, seqOverlaps :354-451,
 * readBED / ailist_add src/igd_base.c:601-649, output :1054-1061.  gType-1 databases only (the
 * reference reads 16-byte records unconditionally).
 *
 * Two library-dependent details of the reference are fixed here the way glibc <= 2.36 behaves
 * (its qsort is a stable merge sort whenever the temporary buffer can be allocated):
 *   - queries of a contig are ordered by start, ties in file order   (qsort(compare_qstart), :370)
 *   - overlaps of a query by dataset index, ties in discovery order  (qsort(compare_fidx),  :382)
 * The reference's quirk idx_t = n1 (the QUERY's first tile, also for records found in later
 * tiles, :291,:337) is kept: a "column" is (index inside its tile, first tile of the query). */
typedef struct { int32_t idx_t, idx_g, idx_f; float sm; } sq_ovl;
typedef struct { int32_t start, end; } sq_iv;
typedef struct { char *name; sq_iv *iv; int64_t n, cap; } sq_ctg;

static void sq_push(sq_ovl **L, int32_t *nn, int32_t *mm, int32_t it, int32_t ig, int32_t f, float sm)
{
    if (*nn == *mm) { *mm = *mm ? 2 * *mm : 1024; *L = (sq_ovl *)realloc(*L, sizeof(sq_ovl) * (size_t)*mm); }
    sq_ovl *p = &(*L)[(*nn)++];
    p->idx_t = it; p->idx_g = ig; p->idx_f = f; p->sm = sm;
}

/* seq_overlaps, src/igd_search.c:253-352 */
static void sq_overlaps(orc_db *db, const char *chrm, int32_t qs, int32_t qe, sq_ovl **L, int32_t *nn, int32_t *mm)
{
    const float qlen = (float)(qe - qs);
    const int32_t ichr = orc_get_id(db, chrm);
    if (ichr < 0) return;
    int32_t n1 = qs / db->nbp, n2 = (qe - 1) / db->nbp;
    const int32_t mTile = db->nTile[ichr] - 1;
    if (n1 > mTile || n1 < 0) return;
    if (n2 > mTile) n2 = mTile;
    int32_t cnt = db->nCnt[ichr][n1];
    if (cnt <= 0) return;                                   /* everything is nested in if(tmpi>0), :266 */
    const int rs = db->rs;
    const int32_t *g = fetch_tile(db, ichr, n1, cnt);
    if (qe > R_START(g, rs, 0)) {
        const int32_t tL = bisect_inline(g, rs, cnt, qe);
        for (int32_t i = tL; i >= 0; i--)
            if (R_END(g, rs, i) > qs) {
                const int32_t re = R_END(g, rs, i), r0 = R_START(g, rs, i);
                const float st = (float)((qe < re ? qe : re) - (qs > r0 ? qs : r0));
                const float rlen = (float)(re - r0);
                sq_push(L, nn, mm, n1, i, R_IDX(g, rs, i), st / (qlen + rlen - st));
            }
    }
    int32_t bd = db->nbp * (n1 + 1);
    for (int32_t j = n1 + 1; j <= n2; j++, bd += db->nbp) {
        cnt = db->nCnt[ichr][j];
        if (cnt <= 0) continue;
        g = fetch_tile(db, ichr, j, cnt);
        if (qe <= R_START(g, rs, 0)) continue;
        int32_t tS = 0;
        while (tS < cnt && R_START(g, rs, tS) < bd) tS++;
        const int32_t tL = bisect_inline(g, rs, cnt, qe);
        for (int32_t i = tL; i >= tS; i--)
            if (R_END(g, rs, i) > qs) {
                const int32_t re = R_END(g, rs, i), r0 = R_START(g, rs, i);
                const float st = (float)((qe < re ? qe : re) - (qs > r0 ? qs : r0));
                const float rlen = (float)(re - r0);
                sq_push(L, nn, mm, n1, i, R_IDX(g, rs, i), st / (qlen + rlen - st));    /* idx_t = n1, :337 */
            }
    }
}

/* seq_overlaps for ONE interval, for the tests: up to `cap` entries {idx_t, idx_g, idx_f, float bits of sm} into out[4 * k];
 * returns the number of overlaps (which may exceed cap). */
int64_t orc_seq_overlaps(orc_db *db, const char *chrm, int32_t qs, int32_t qe, int32_t *out, int64_t cap)
{
    sq_ovl *L = NULL;
    int32_t nn = 0, mm = 0;
    sq_overlaps(db, chrm, qs, qe, &L, &nn, &mm);
    for (int32_t k = 0; k < nn && k < cap; k++) {
        out[4 * k] = L[k].idx_t; out[4 * k + 1] = L[k].idx_g; out[4 * k + 2] = L[k].idx_f;
        memcpy(&out[4 * k + 3], &L[k].sm, 4);
    }
    free(L);
    return nn;
}

static void sq_stable_sort_iv(sq_iv *a, int64_t n)          /* by start, ties keep their order */
{
    if (n < 2) return;
    sq_iv *t = (sq_iv *)malloc(sizeof(sq_iv) * (size_t)n), *src = a, *dst = t;
    for (int64_t w = 1; w < n; w <<= 1) {
        for (int64_t lo = 0; lo < n; lo += 2 * w) {
            int64_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n, i = lo, j = mid, k = lo;
            while (i < mid && j < hi) dst[k++] = (src[j].start < src[i].start) ? src[j++] : src[i++];
            while (i < mid) dst[k++] = src[i++];
            while (j < hi) dst[k++] = src[j++];
        }
        sq_iv *x = src; src = dst; dst = x;
    }
    if (src != a) memcpy(a, src, sizeof(sq_iv) * (size_t)n);
    free(t);
}

static void sq_stable_sort_ovl(sq_ovl *a, int64_t n)        /* by idx_f, ties keep their order */
{
    if (n < 2) return;
    sq_ovl *t = (sq_ovl *)malloc(sizeof(sq_ovl) * (size_t)n), *src = a, *dst = t;
    for (int64_t w = 1; w < n; w <<= 1) {
        for (int64_t lo = 0; lo < n; lo += 2 * w) {
            int64_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n, i = lo, j = mid, k = lo;
            while (i < mid && j < hi) dst[k++] = (src[j].idx_f < src[i].idx_f) ? src[j++] : src[i++];
            while (i < mid) dst[k++] = src[i++];
            while (j < hi) dst[k++] = src[j++];
        }
        sq_ovl *x = src; src = dst; dst = x;
    }
    if (src != a) memcpy(a, src, sizeof(sq_ovl) * (size_t)n);
    free(t);
}

/* seqOverlaps, src/igd_search.c:354-451.  sm[nFiles].  Returns 0, -1 if the file cannot be read. */
int orc_seqOverlaps(orc_db *db, const char *qfile, double *sm)
{
    lreader r;
    if (lr_open(&r, qfile) != 0) return -1;
    sq_ctg *ctg = NULL;
    int32_t nctg = 0, mctg = 0;
    while (lr_next(&r) >= 0) {                                /* readBED, src/igd_base.c:628-649 */
        int32_t st, en;
        char *name = orc_parse_bed(r.line, &st, &en);
        if (!name) continue;
        if ((uint32_t)st > (uint32_t)en) continue;             /* ailist_add: uint32 s > e, :603 */
        int32_t k = 0;
        while (k < nctg && strcmp(ctg[k].name, name) != 0) k++;
        if (k == nctg) {
            if (nctg == mctg) { mctg = mctg ? 2 * mctg : 32; ctg = (sq_ctg *)realloc(ctg, sizeof(sq_ctg) * (size_t)mctg); }
            ctg[nctg].name = strdup(name); ctg[nctg].iv = NULL; ctg[nctg].n = ctg[nctg].cap = 0;
            nctg++;
        }
        sq_ctg *c = &ctg[k];
        if (c->n == c->cap) { c->cap = c->cap ? 2 * c->cap : 64; c->iv = (sq_iv *)realloc(c->iv, sizeof(sq_iv) * (size_t)c->cap); }
        c->iv[c->n].start = st; c->iv[c->n].end = en; c->n++;
    }
    lr_close(&r);
    const int32_t nfiles = db->nFiles;
    int64_t Nq = 0;
    db->preChr = -6; db->preIdx = -8;
    for (int32_t m = 0; m < nfiles; m++) sm[m] = 0.0;
    for (int32_t ci = 0; ci < nctg; ci++) {
        sq_ctg *c = &ctg[ci];
        const int64_t nq = c->n;
        Nq += nq;
        sq_stable_sort_iv(c->iv, nq);
        sq_ovl **olps = (sq_ovl **)calloc((size_t)(nq ? nq : 1), sizeof(sq_ovl *));
        int32_t *nh = (int32_t *)calloc((size_t)(nq ? nq : 1), sizeof(int32_t));
        sq_ovl *L = NULL; int32_t nn = 0, mm = 0;
        for (int64_t j = 0; j < nq; j++) {
            nn = 0;
            sq_overlaps(db, c->name, c->iv[j].start, c->iv[j].end, &L, &nn, &mm);
            if (nn > 0) {
                sq_stable_sort_ovl(L, nn);
                olps[j] = (sq_ovl *)malloc(sizeof(sq_ovl) * (size_t)nn);
                memcpy(olps[j], L, sizeof(sq_ovl) * (size_t)nn);
            }
            nh[j] = nn;
        }
        free(L);
        int32_t *kst0 = (int32_t *)calloc((size_t)(nq ? nq : 1), sizeof(int32_t));
        int32_t *kst = (int32_t *)calloc((size_t)(nq ? nq : 1), sizeof(int32_t));
        int32_t *nst0 = (int32_t *)calloc((size_t)(nq ? nq : 1), sizeof(int32_t));
        for (int32_t m = 0; m < nfiles; m++) {
            float maxf = 0.0f;
            int64_t maxj = 0; int32_t maxk = 0;
            for (int64_t j = 0; j < nq; j++) {                /* 1. the best pair of dataset m, :392-409 */
                int32_t k = kst[j];
                while (k < nh[j] && olps[j][k].idx_f < m) k++;
                kst0[j] = k;
                while (k < nh[j] && olps[j][k].idx_f == m) {
                    if (olps[j][k].sm > maxf) { maxf = olps[j][k].sm; maxk = k; maxj = j; }
                    k++;
                }
                kst[j] = k;
                nst0[j] = k - kst0[j];
            }
            while (maxf > 0.0f) {                             /* 2. take it, drop its row and column, :412-430 */
                sm[m] += maxf;
                nst0[maxj] = 0;
                const int32_t it = olps[maxj][maxk].idx_t, ig = olps[maxj][maxk].idx_g;
                maxf = 0.0f;
                for (int64_t j = 0; j < nq; j++) {
                    if (nst0[j] <= 0) continue;
                    for (int32_t k = kst0[j]; k < kst0[j] + nst0[j]; k++) {
                        if (olps[j][k].idx_g == ig && olps[j][k].idx_t == it) olps[j][k].sm = 0.0f;
                        else if (olps[j][k].sm > maxf) { maxf = olps[j][k].sm; maxk = k; maxj = j; }
                    }
                }
            }
        }
        free(nst0); free(kst); free(kst0); free(nh);
        for (int64_t j = 0; j < nq; j++) free(olps[j]);
        free(olps);
    }
    for (int32_t m = 0; m < nfiles; m++) sm[m] = sm[m] / ((double)Nq + db->fileNr[m] - sm[m]);   /* :446-449 */
    for (int32_t k = 0; k < nctg; k++) { free(ctg[k].name); free(ctg[k].iv); }
    free(ctg);
    return 0;
}

/* src/igd_search.c:889-1079.  Same flag loop (:931-971), same dispatch (:975-1053), same
 * text.  `-m` and `-s` are outside the hot path and are not restated. */
int orc_igd_search(int argc, char **argv, FILE *out)
{
    if (argc < 4) { fprintf(stderr, "usage: igd search <db.igd> [options]\n"); return 0; }
    const char *igdName = argv[2];
    size_t L = strlen(igdName);
    if (L < 4 || strcmp(".igd", igdName + L - 4) != 0) {
        fprintf(out, "%s is not an igd database", igdName);
        return 0;
    }
    FILE *probe = fopen(igdName, "rb");
    if (!probe) { fprintf(out, "%s does not exist", igdName); return 0; }
    fclose(probe);
    orc_db *db = orc_open(igdName);
    if (!db) { fprintf(out, "cannot load %s\n", igdName); return 0; }
    int32_t v = 0, qs = 1, qe = 2;
    int mode = -1, p_mode = 0;
    const char *chrm = NULL, *qfName = "";
    char outName[64] = "";
    for (int i = 3; i < argc; i++) {
        if (strcmp(argv[i], "-q") == 0) {
            if (i + 1 < argc) { qfName = argv[i + 1]; mode = 1; }
            else { fprintf(out, "No query file.\n"); orc_close(db); return 0; }
        } else if (strcmp(argv[i], "-r") == 0) {
            if (i + 3 < argc) { mode = 2; chrm = argv[i + 1]; qs = atoi(argv[i + 2]); qe = atoi(argv[i + 3]); }
        } else if (strcmp(argv[i], "-v") == 0) {
            if (i + 1 < argc) v = atoi(argv[i + 1]);
        } else if (strcmp(argv[i], "-m") == 0) mode = 0;
        else if (strcmp(argv[i], "-o") == 0) {
            if (i + 1 < argc) { strncpy(outName, argv[i + 1], 63); outName[63] = '\0'; }
        }
        else if (strcmp(argv[i], "-s") == 0 && mode != 2) mode = 3;
        else if (strcmp(argv[i], "-f") == 0) p_mode = 1;
    }
    int64_t *hits = (int64_t *)calloc((size_t)db->nFiles + 1, sizeof(int64_t));
    if (p_mode == 1) {                                           /* :975-995            */
        if (mode == 1)
            fprintf(out, "Total overlaps: %lld\n", (long long)orc_getOverlaps_f(db, qfName, out));
        else if (mode == 2)
            fprintf(out, "Total overlaps: %lld\n", (long long)orc_get_overlaps_f(db, chrm, qs, qe, out));
        else
            fprintf(out, "Not supported -f option\n");
    } else if (mode == 0) {                                      /* :996-1022           */
        const size_t nf = (size_t)db->nFiles;
        uint32_t *hm = (uint32_t *)calloc(nf * nf + 1, sizeof(uint32_t));
        orc_getMap(db, v > 0, v, hm, out);
        if (strlen(outName) < 2) strcpy(outName, "Hitsmap");
        FILE *fo = fopen(outName, "w");
        if (!fo) fprintf(out, "Can't open file %s\n", outName);
        else {
            fprintf(fo, "%u\t%u\t%u\n", (unsigned)nf, (unsigned)nf, (unsigned)v);
            for (size_t a = 0; a < nf; a++) {
                for (size_t b = 0; b < nf; b++) fprintf(fo, "%u\t", hm[a * nf + b]);
                fprintf(fo, "\n");
            }
            fclose(fo);
        }
        free(hm);
    } else if (mode == 1) {                                      /* :1023-1040          */
        if (db->gType == 0 || v <= 0) orc_getOverlaps(db, qfName, hits);
        else orc_getOverlaps_v(db, qfName, hits, v);
        fprintf(out, "index\t number of regions\t number of hits\t File_name\n");
        int64_t total = 0;
        for (int32_t i = 0; i < db->nFiles; i++) {
            if (hits[i] > 0)
                fprintf(out, "%i\t%i\t%lld\t%s\n", i, db->fileNr[i], (long long)hits[i], db->fileName[i]);
            total += hits[i];
        }
        fprintf(out, "Total: %lld\n", (long long)total);
    } else if (mode == 2) {                                      /* :1041-1053          */
        if (db->gType == 0 || v <= 0) orc_get_overlaps(db, chrm, qs, qe, hits);
        else orc_get_overlaps_v(db, chrm, qs, qe, v, hits);
        fprintf(out, "index\t number of regions\t number of hits\t File_name\n");
        for (int32_t i = 0; i < db->nFiles; i++)
            fprintf(out, "%i\t%i\t%lld\t%s\n", i, db->fileNr[i], (long long)hits[i], db->fileName[i]);
    } else if (mode == 3) {                                      /* :1054-1061          */
        double *smv = (double *)calloc((size_t)db->nFiles + 1, sizeof(double));
        orc_seqOverlaps(db, qfName, smv);
        fprintf(out, "index\t number of regions\t similarity\t dataset name\n");
        for (int32_t i = 0; i < db->nFiles; i++)
            fprintf(out, "%i\t%i\t%10.6f\t%s\n", i, db->fileNr[i], smv[i], db->fileName[i]);
        free(smv);
    } else {
        fprintf(out, "oracle: missing -q/-r\n");
    }
    free(hits);
    orc_close(db);
    return 0;
}
