/* igd_create.c -- host side of `igd create`: BED text -> interval arrays -> (GPU) -> .igd + _index.tsv
 * (SURVEY.md section 8f, row f4).
 *
 * Reference counterparts (databio/IGD, /root/reference):
 *   igd_create        src/igd_create.c:436-501   option parsing, path fix-ups, "exists!" check
 *   create_igd        src/igd_create.c:25-121    glob of BED files (default)
 *   create_igd_f      src/igd_create.c:124-243   -f: a text file listing BED files
 *   create_igd0       src/igd_create.c:246-343   -s 0: 12-byte records, no value
 *   create_igd_bed4   src/igd_create.c:346-433   -s 2: one BED4+ file, dataset name in column 4
 *   str_splits        src/igd_base.c:37-51       tab splitting with its creeping column limit
 *   create_iGD        src_py/igd_create.c:19-143, IGDr/src/igd_create.c:19-170 (same loop, 256-byte lines)
 * The work between "intervals in input order" and "sorted tile records" -- igd_add, igd_saveT,
 * igd_save, radix_sort_intv -- runs on the GPU (igd_hip_create, igd_create.hip); there is no CPU
 * version of it in this library, and without a GPU `create` fails.
 *
 * What stays on the host and how:
 *   - files are parsed in parallel, one file per thread at a time; the global order the reference
 *     sees (files in glob/list order, lines in file order) is restored when the per-file arrays are
 *     concatenated, and contig numbers follow first appearance in that order;
 *   - lines are cut the way gzgets(buf, 1024|256) cuts them: a longer line becomes several "lines";
 *   - str_splits overwrites its column limit with the number of columns of the line it just split,
 *     so a line after a short one is split into fewer columns than it has (src/igd_base.c:49-50).
 *     This sequential dependence matters only when column counts are mixed; the parallel parse
 *     detects that case (any line under 5 columns while another has 5 or more, or any under 3) and
 *     the input is then re-read by ONE thread carrying the limit exactly like the reference.
 * Deviations (reference UB): lines that end up with fewer than 3 columns are skipped (the reference
 * reads stale pointers); start < 0 is dropped; the -f mode's uninitialised value is 0; the 40-byte
 * contig-name fields are zero-padded; no data0/ temp files; fewer than 10 input files work (the
 * reference divides by n_files/10) and print no progress dots.
 */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <glob.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <zlib.h>

#include "igd_core.h"
#include "igd_create_host.h"

static double now_s(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

/* ---- small string dictionary: name -> dense id in first-seen order --------------------------- */
typedef struct {
    char **name; int32_t *len; int32_t n, cap;
    int32_t *slot; int32_t nslot;
} strdict;

static uint32_t hash_bytes(const char *s, size_t n)
{
    uint32_t h = 2166136261u;
    for (size_t i = 0; i < n; i++) h = (h ^ (unsigned char)s[i]) * 16777619u;
    return h;
}

static void dict_rehash(strdict *d, int32_t nslot)
{
    free(d->slot);
    d->nslot = nslot;
    d->slot = (int32_t *)malloc(sizeof(int32_t) * (size_t)nslot);
    for (int32_t i = 0; i < nslot; i++) d->slot[i] = -1;
    for (int32_t i = 0; i < d->n; i++) {
        uint32_t p = hash_bytes(d->name[i], (size_t)d->len[i]) & (uint32_t)(nslot - 1);
        while (d->slot[p] >= 0) p = (p + 1) & (uint32_t)(nslot - 1);
        d->slot[p] = i;
    }
}

static int32_t dict_id(strdict *d, const char *s, size_t len)
{
    if (d->nslot == 0) dict_rehash(d, 64);
    uint32_t p = hash_bytes(s, len) & (uint32_t)(d->nslot - 1);
    for (;;) {
        const int32_t i = d->slot[p];
        if (i < 0) break;
        if ((size_t)d->len[i] == len && memcmp(d->name[i], s, len) == 0) return i;
        p = (p + 1) & (uint32_t)(d->nslot - 1);
    }
    if (d->n == d->cap) {
        d->cap = d->cap ? 2 * d->cap : 32;
        d->name = (char **)realloc(d->name, sizeof(char *) * (size_t)d->cap);
        d->len = (int32_t *)realloc(d->len, sizeof(int32_t) * (size_t)d->cap);
    }
    d->name[d->n] = (char *)malloc(len + 1);
    memcpy(d->name[d->n], s, len);
    d->name[d->n][len] = '\0';
    d->len[d->n] = (int32_t)len;
    d->slot[p] = d->n++;
    if (d->n * 2 > d->nslot) dict_rehash(d, d->nslot * 2);
    return d->n - 1;
}

static void dict_free(strdict *d)
{
    for (int32_t i = 0; i < d->n; i++) free(d->name[i]);
    free(d->name); free(d->len); free(d->slot);
    memset(d, 0, sizeof *d);
}

/* ---- whole file in memory, plain or gzip (the reference reads through gzopen/gzgets) ---------- */
static int load_file(const char *path, char **buf, size_t *len)
{
    gzFile z = gzopen(path, "r");
    if (!z) return 1;
    gzbuffer(z, 1 << 20);
    size_t cap = 1 << 22, n = 0;
    char *b = (char *)malloc(cap);
    for (;;) {
        if (cap - n < (1 << 20)) { cap *= 2; b = (char *)realloc(b, cap); }
        const int got = gzread(z, b + n, (unsigned)(cap - n > (1u << 30) ? (1u << 30) : cap - n));
        if (got <= 0) break;
        n += (size_t)got;
    }
    gzclose(z);
    *buf = b; *len = n;
    return 0;
}

/* atol() of the text at p (bounded by e), narrowed to int32 like `int32_t st = atol(..)` */
static int32_t atol32(const char *p, const char *e)
{
    while (p < e && (*p == ' ' || (*p >= '\t' && *p <= '\r'))) p++;
    int neg = 0;
    if (p < e && (*p == '+' || *p == '-')) { neg = (*p == '-'); p++; }
    unsigned long long v = 0;
    int sat = 0;
    for (; p < e && *p >= '0' && *p <= '9'; p++) {
        if (v > (0x7fffffffffffffffULL - (unsigned)(*p - '0')) / 10) sat = 1;
        if (!sat) v = v * 10 + (unsigned)(*p - '0');
    }
    long long r;
    if (sat) r = neg ? (long long)(-0x7fffffffffffffffLL - 1) : 0x7fffffffffffffffLL;
    else r = neg ? -(long long)v : (long long)v;
    return (int32_t)r;
}

/* ---- intervals of one input file ------------------------------------------------------------- */
#define COLCAP 8                         /* column counts are only compared with 3 and 5          */
typedef struct {
    int32_t *ctg, *start, *end, *value, *file;
    int64_t n, cap;
    strdict names;                       /* contigs, local first-seen order                        */
    int32_t nr;                          /* "Number of regions" (GLOB/LIST/GTYPE0: this file)      */
    double avg;
    int minC, maxC;                      /* over the pieces seen (column count capped at COLCAP)   */
    int failed;
} part;

static void part_push(part *P, int32_t c, int32_t s, int32_t e, int32_t v, int32_t f)
{
    if (P->n == P->cap) {
        P->cap = P->cap ? 2 * P->cap : 4096;
        P->ctg = (int32_t *)realloc(P->ctg, sizeof(int32_t) * (size_t)P->cap);
        P->start = (int32_t *)realloc(P->start, sizeof(int32_t) * (size_t)P->cap);
        P->end = (int32_t *)realloc(P->end, sizeof(int32_t) * (size_t)P->cap);
        P->value = (int32_t *)realloc(P->value, sizeof(int32_t) * (size_t)P->cap);
        P->file = (int32_t *)realloc(P->file, sizeof(int32_t) * (size_t)P->cap);
    }
    P->ctg[P->n] = c; P->start[P->n] = s; P->end[P->n] = e; P->value[P->n] = v; P->file[P->n] = f;
    P->n++;
}

static void part_free(part *P)
{
    free(P->ctg); free(P->start); free(P->end); free(P->value); free(P->file);
    dict_free(&P->names);
    memset(P, 0, sizeof *P);
}

/* str_splits on the piece [s,e): at most `limit` fields (the reference's *nmax + 1); the last one
 * keeps any further tabs.  f[i] = start of field i for i < min(ns, 5); returns ns (<= COLCAP). */
static int split_piece(const char *s, const char *e, int limit, const char **f)
{
    int ns = 1;
    f[0] = s;
    if (limit > COLCAP) limit = COLCAP;
    const char *q = s;
    while (ns < limit) {
        q = (const char *)memchr(q, '\t', (size_t)(e - q));
        if (!q) break;
        q++;
        if (ns < 5) f[ns] = q;
        ns++;
    }
    return ns;
}

typedef struct {
    int mode, linebuf;
    int exact;                           /* carry the column limit from piece to piece           */
    int nCols;                           /* the reference's nCols (only when exact)               */
    strdict *datasets;                   /* BED4: dataset names -> file index                     */
    int32_t **ds_nr; double **ds_avg; int32_t *ds_cap;
} parse_ctx;

/* the interval of igd_add (src/igd_base.c:118-121): contigs are registered only by kept intervals */
static void add_interval(part *P, const char *name, size_t nlen, int32_t st, int32_t en, int32_t va, int32_t f)
{
    if (st >= en || st < 0) return;
    part_push(P, dict_id(&P->names, name, nlen), st, en, va, f);
}

static void parse_piece(parse_ctx *X, part *P, const char *s, const char *e, int32_t fileIdx)
{
    if (X->mode == IGDC_CREATE_LIST) {                      /* parse_bed + the filter of :187-188 */
        const char *t1 = (const char *)memchr(s, '\t', (size_t)(e - s));
        if (!t1) return;
        const char *t2 = (const char *)memchr(t1 + 1, '\t', (size_t)(e - t1 - 1));
        if (!t2) return;
        const size_t nl = (size_t)(t1 - s);
        const int32_t st = atol32(t1 + 1, e), en = atol32(t2 + 1, e);
        if (!(nl >= 3 && s[0] == 'c' && s[1] == 'h' && s[2] == 'r' && nl < 40 && en > 0)) return;
        if (!(st >= 0 && en < 321000000)) return;
        P->nr++;
        P->avg += (double)(int32_t)((uint32_t)en - (uint32_t)st);
        add_interval(P, s, nl, st, en, 0, fileIdx);
        return;
    }
    const char *f[5];
    const int ns = split_piece(s, e, X->exact ? X->nCols + 1 : COLCAP, f);
    if (X->exact) X->nCols = ns;
    if (ns < P->minC) P->minC = ns;
    if (ns > P->maxC) P->maxC = ns;
    if (X->mode == IGDC_CREATE_BED4) {
        if (ns < 5) return;
        const size_t dl = (size_t)(f[4] - 1 - f[3]);
        const int32_t idx = dict_id(X->datasets, f[3], dl);
        if (idx >= *X->ds_cap) {
            const int32_t nc = *X->ds_cap ? 2 * *X->ds_cap : 256;
            *X->ds_nr = (int32_t *)realloc(*X->ds_nr, sizeof(int32_t) * (size_t)nc);
            *X->ds_avg = (double *)realloc(*X->ds_avg, sizeof(double) * (size_t)nc);
            for (int32_t i = *X->ds_cap; i < nc; i++) { (*X->ds_nr)[i] = 0; (*X->ds_avg)[i] = 0.0; }
            *X->ds_cap = nc;
        }
        const int32_t st = atol32(f[1], e), en = atol32(f[2], e);
        (*X->ds_nr)[idx]++;
        (*X->ds_avg)[idx] += (double)(int32_t)((uint32_t)en - (uint32_t)st);
        add_interval(P, s, (size_t)(f[1] - 1 - s), st, en, atol32(f[4], e), idx);
        return;
    }
    if (ns < 3) return;                                     /* deviation: see the header           */
    const int32_t st = atol32(f[1], e), en = atol32(f[2], e);
    const int32_t va = (X->mode == IGDC_CREATE_GLOB && ns > 4) ? atol32(f[4], e) : 0;
    P->nr++;
    P->avg += (double)(int32_t)((uint32_t)en - (uint32_t)st);
    add_interval(P, s, (size_t)(f[1] - 1 - s), st, en, va, fileIdx);
}

static int parse_file(parse_ctx *X, part *P, const char *path, int32_t fileIdx)
{
    char *buf; size_t len;
    if (load_file(path, &buf, &len) != 0) { P->failed = 1; return -1; }
    const char *p = buf, *end = buf + len;
    const size_t chunk = (size_t)X->linebuf - 1;
    while (p < end) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl + 1 : end;                 /* gzgets keeps the '\n'                */
        while (p < le) {
            const char *q = (size_t)(le - p) > chunk ? p + chunk : le;
            parse_piece(X, P, p, q, fileIdx);
            p = q;
        }
    }
    free(buf);
    return 0;
}

/* ---- thread pool over files -------------------------------------------------------------------- */
typedef struct {
    char **files; int32_t nf; part *parts; int mode, linebuf;
    volatile int32_t next;
} pool;

static void *pool_run(void *arg)
{
    pool *Q = (pool *)arg;
    for (;;) {
        const int32_t f = __sync_fetch_and_add(&Q->next, 1);
        if (f >= Q->nf) break;
        parse_ctx X;
        memset(&X, 0, sizeof X);
        X.mode = Q->mode; X.linebuf = Q->linebuf;
        Q->parts[f].minC = COLCAP; Q->parts[f].maxC = 0;
        parse_file(&X, &Q->parts[f], Q->files[f], f);
    }
    return NULL;
}

/* per-file arrays -> one array per column at the file's offset, contig numbers made global */
typedef struct {
    part *parts; int32_t nparts; const int64_t *poff; int32_t **map;
    int32_t *ctg, *start, *end, *value, *file;
    volatile int32_t next;
} gather;

static void *gather_run(void *arg)
{
    gather *G = (gather *)arg;
    for (;;) {
        const int32_t f = __sync_fetch_and_add(&G->next, 1);
        if (f >= G->nparts) break;
        part *P = &G->parts[f];
        const int64_t o0 = G->poff[f];
        const int32_t *map = G->map[f];
        for (int64_t i = 0; i < P->n; i++) G->ctg[o0 + i] = map[P->ctg[i]];
        memcpy(G->start + o0, P->start, sizeof(int32_t) * (size_t)P->n);
        memcpy(G->end + o0, P->end, sizeof(int32_t) * (size_t)P->n);
        memcpy(G->value + o0, P->value, sizeof(int32_t) * (size_t)P->n);
        memcpy(G->file + o0, P->file, sizeof(int32_t) * (size_t)P->n);
        free(P->ctg); free(P->start); free(P->end); free(P->value); free(P->file);
        P->ctg = P->start = P->end = P->value = P->file = NULL;
    }
    return NULL;
}

static int n_threads(int32_t nf)
{
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    const char *e = getenv("IGD_PARSE_THREADS");
    if (e && atoi(e) > 0) n = atoi(e);
    if (n < 1) n = 1;
    if (n > 64) n = 64;
    if (n > nf) n = nf;
    return (int)n;
}

/* ---- writing ----------------------------------------------------------------------------------- */
/* _index.tsv, src/igd_create.c:93-110 */
static void write_index(const char *path, char **files, int32_t nf, const int32_t *nr, const double *avg,
                        int64_t *nT, double *l_avg)
{
    *nT = 0; *l_avg = 0.0;
    FILE *fpi = fopen(path, "w");
    if (!fpi) { printf("Can't open file %s", path); return; }
    fprintf(fpi, "Index\tFile\tNumber of regions\tAvg size\n");
    for (int32_t i = 0; i < nf; i++) {
        const char *t = strrchr(files[i], '/');
        t = t ? t + 1 : files[i];
        *nT += nr[i];
        *l_avg += avg[i];
        fprintf(fpi, "%i\t%s\t%i\t%f\n", i, t, nr[i], avg[i] / nr[i]);
    }
    fclose(fpi);
}

static void *warm_gpu(void *arg)
{
    (void)arg;
    (void)igd_hip_device_count();          /* first HIP call: loads and initialises the runtime */
    return NULL;
}

/* ------------------------------------------------------------------------------------------------ */
int igdc_create(const igdc_create_opts *o)
{
    const int cli = o->msg == IGDC_MSG_CLI, py = o->msg == IGDC_MSG_PY, rr = o->msg == IGDC_MSG_R;
    char **files = NULL;
    int32_t nf = 0;
    glob_t g;
    int globbed = 0, rc = 0;
    part *parts = NULL;
    int32_t nparts = 0;
    strdict datasets;
    memset(&datasets, 0, sizeof datasets);
    int32_t *ds_nr = NULL; double *ds_avg = NULL; int32_t ds_cap = 0;
    mkdir(o->opath, 0777);
    const int timing = getenv("IGD_TIMING") != NULL;
    double t0 = now_s(), t1;
#define PHASE(what) do { if (timing) { t1 = now_s(); fprintf(stderr, "[igd create] %-28s %9.3f ms\n", what, 1e3 * (t1 - t0)); t0 = t1; } } while (0)

    /* 1. the input files */
    if (o->mode == IGDC_CREATE_BED4) {
        if (cli) printf("igd_create 1\n");
        nf = 1;
        files = (char **)malloc(sizeof(char *));
        files[0] = strdup(o->ipath);
    } else if (o->mode == IGDC_CREATE_LIST) {               /* src/igd_create.c:131-163 */
        if (cli) printf("Create igd from %s: \n", o->ipath);
        FILE *fl = fopen(o->ipath, "r");
        if (!fl) { printf("Can't open file %s", o->ipath); return 1; }
        char buf[1024];
        int32_t cap = 0;
        while (fgets(buf, 1024, fl) != NULL) {
            buf[strcspn(buf, "\n")] = 0;
            gzFile z = gzopen(buf, "r");
            if (!z) continue;
            char first[1024];
            first[0] = 0;
            if (gzgets(z, first, 1024) == NULL) first[0] = 0;
            gzclose(z);
            int32_t st, en;
            if (igdc_parse_bed(first, &st, &en, 1)) {       /* kept only if its first line is a valid BED line */
                if (nf == cap) { cap = cap ? 2 * cap : 64; files = (char **)realloc(files, sizeof(char *) * (size_t)cap); }
                files[nf++] = strdup(buf);
            }
        }
        fclose(fl);
        if (nf < 1) { printf("Too few files (add to path /*): %i\n", nf); free(files); return 1; }
    } else {
        if (o->mode == IGDC_CREATE_GTYPE0 || py) { if (cli || py) printf("igd_create 0\n"); }
        else if (cli) printf("Create igd from %s: \n", o->ipath);
        if (glob(o->ipath, 0, NULL, &g) != 0) {
            printf(o->mode == IGDC_CREATE_GTYPE0 || py ? "wrong dir path: %s" : "wrong dir path: %s\n", o->ipath);
            return 1;
        }
        globbed = 1;
        files = g.gl_pathv;
        nf = (int32_t)g.gl_pathc;
        if ((o->mode == IGDC_CREATE_GTYPE0 && cli) || py || rr) printf("igd_create 1: %i\n", nf);
    }

    /* the HIP runtime starts up (~70 ms) while the files are parsed */
    pthread_t warm;
    const int warming = pthread_create(&warm, NULL, warm_gpu, NULL) == 0;

    /* 2. parse: one part per file, in parallel; BED4 is one file, one thread */
    nparts = nf;
    parts = (part *)calloc((size_t)nparts, sizeof(part));
    parse_ctx X;
    memset(&X, 0, sizeof X);
    X.mode = o->mode; X.linebuf = o->linebuf;
    X.datasets = &datasets; X.ds_nr = &ds_nr; X.ds_avg = &ds_avg; X.ds_cap = &ds_cap;
    if (o->mode == IGDC_CREATE_BED4) {
        X.exact = 1; X.nCols = 32;                          /* src/igd_create.c:349 */
        parts[0].minC = COLCAP;
        if (parse_file(&X, &parts[0], files[0], 0) != 0) rc = 1;
    } else {
        pool Q;
        Q.files = files; Q.nf = nf; Q.parts = parts; Q.mode = o->mode; Q.linebuf = o->linebuf; Q.next = 0;
        const int nt = n_threads(nf);
        pthread_t th[64];
        int up[64];                                         /* a worker that could not be started is simply missing: the pool is pulled from */
        for (int t = 1; t < nt; t++) up[t] = pthread_create(&th[t], NULL, pool_run, &Q) == 0;
        pool_run(&Q);
        for (int t = 1; t < nt; t++) if (up[t]) pthread_join(th[t], NULL);
        int minC = COLCAP, maxC = 0;
        for (int32_t f = 0; f < nf; f++) {
            if (parts[f].failed) rc = 1;                   /* gzopen failed: the reference returns */
            if (parts[f].minC < minC) minC = parts[f].minC;
            if (parts[f].maxC > maxC) maxC = parts[f].maxC;
        }
        const int mixed = o->mode != IGDC_CREATE_LIST && maxC > 0 &&
                          (minC < 3 || (o->mode == IGDC_CREATE_GLOB && minC < 5 && maxC >= 5));
        if (rc == 0 && mixed) {                             /* the column limit creeps: redo in order */
            X.exact = 1; X.nCols = 16;                      /* src/igd_create.c:46 */
            for (int32_t f = 0; f < nf; f++) {
                part_free(&parts[f]);
                parts[f].minC = COLCAP;
                parse_file(&X, &parts[f], files[f], f);
            }
        }
    }

    if (warming) pthread_join(warm, NULL);
    PHASE("parse (host threads) || HIP start-up");
    int32_t *nr = NULL; double *avg = NULL;
    char **idxNames = files;
    int32_t nIdx = nf;
    int64_t n = 0, *poff = NULL;
    int32_t *ctg = NULL, *start = NULL, *end = NULL, *value = NULL, *file = NULL;
    strdict contigs;
    memset(&contigs, 0, sizeof contigs);
    igd_hip_created C;
    memset(&C, 0, sizeof C);
    if (rc != 0) goto out;

    /* 3. global contig numbers (first appearance in input order), one array per column */
    poff = (int64_t *)malloc(sizeof(int64_t) * ((size_t)nparts + 1));
    for (int32_t f = 0; f < nparts; f++) { poff[f] = n; n += parts[f].n; }
    poff[nparts] = n;
    ctg = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n ? n : 1));
    start = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n ? n : 1));
    end = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n ? n : 1));
    value = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n ? n : 1));
    file = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n ? n : 1));
    {
        gather G;
        G.parts = parts; G.nparts = nparts; G.poff = poff; G.next = 0;
        G.ctg = ctg; G.start = start; G.end = end; G.value = value; G.file = file;
        G.map = (int32_t **)calloc((size_t)(nparts ? nparts : 1), sizeof(int32_t *));
        for (int32_t f = 0; f < nparts; f++) {              /* sequential: this IS the first-seen order */
            part *P = &parts[f];
            G.map[f] = (int32_t *)malloc(sizeof(int32_t) * (size_t)(P->names.n ? P->names.n : 1));
            for (int32_t k = 0; k < P->names.n; k++)
                G.map[f][k] = dict_id(&contigs, P->names.name[k], (size_t)P->names.len[k]);
        }
        const int nt = n_threads(nparts);
        pthread_t th[64];
        int up[64];
        for (int t = 1; t < nt; t++) up[t] = pthread_create(&th[t], NULL, gather_run, &G) == 0;
        gather_run(&G);
        for (int t = 1; t < nt; t++) if (up[t]) pthread_join(th[t], NULL);
        for (int32_t f = 0; f < nparts; f++) free(G.map[f]);
        free(G.map);
    }
    if (o->mode == IGDC_CREATE_BED4) {
        nIdx = datasets.n; idxNames = datasets.name;
        nr = (int32_t *)calloc((size_t)(nIdx ? nIdx : 1), sizeof(int32_t));
        avg = (double *)calloc((size_t)(nIdx ? nIdx : 1), sizeof(double));
        for (int32_t i = 0; i < nIdx; i++) { nr[i] = ds_nr[i]; avg[i] = ds_avg[i]; }
    } else {
        nr = (int32_t *)calloc((size_t)(nf ? nf : 1), sizeof(int32_t));
        avg = (double *)calloc((size_t)(nf ? nf : 1), sizeof(double));
        for (int32_t f = 0; f < nf; f++) { nr[f] = parts[f].nr; avg[f] = parts[f].avg; }
    }
    if (cli && o->mode != IGDC_CREATE_GTYPE0 && o->mode != IGDC_CREATE_BED4) {
        const int32_t nf10 = nf / 10;                       /* progress dots, src/igd_create.c:81 */
        for (int32_t ig = 1; nf10 > 0 && ig <= nf; ig++) if (ig % nf10 == 0) printf(".");
    }

    PHASE("concatenate + contig ids");
    /* 4. the GPU: replicate into tiles, order every tile like the reference, gather the records */
    {
        igd_hip_create_desc D;
        D.nbp = o->nbp; D.gType = o->mode == IGDC_CREATE_GTYPE0 ? 0 : 1; D.nCtg = contigs.n; D.n = n;
        D.ctg = ctg; D.start = start; D.end = end; D.value = o->mode == IGDC_CREATE_GTYPE0 ? NULL : value; D.file = file;
        D.ctgName = (const char *const *)contigs.name;
        const size_t L = strlen(o->opath) + strlen(o->name) + 16;
        char *path = (char *)malloc(L);
        snprintf(path, L, "%s%s.igd", o->opath, o->name);
        D.out_fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);   /* the engine streams header + tiles into it */
        if (D.out_fd < 0) { printf("Can't open file %s", path); free(path); rc = 1; goto out; }
        rc = igd_hip_create(&D, o->device, &C);
        close(D.out_fd);
        if (rc != IGD_HIP_OK) unlink(path);                  /* no half-written database left behind */
        free(path);
        if (rc != IGD_HIP_OK) {
            fprintf(stderr, "igd create: the GPU engine failed (%d): %s\n"
                            "igd create: this build has no CPU path.\n", rc, igd_hip_last_error());
            goto out;
        }
    }
    PHASE("igd_hip_create (GPU + .igd)");
    if (cli && (o->mode == IGDC_CREATE_GLOB || o->mode == IGDC_CREATE_LIST || (o->mode == IGDC_CREATE_BED4 && datasets.n > 0)))
        printf("nCtgs, nRegions, nTiles: %i\t %lld\t %lld\n", contigs.n, (long long)C.nRecords, (long long)C.nTiles);
    if (cli && (o->mode == IGDC_CREATE_GLOB || o->mode == IGDC_CREATE_LIST)) printf("\n");
    if ((cli && o->mode == IGDC_CREATE_BED4) || py) printf("igd_create 2\n");

    /* 5. files */
    {
        const size_t L = strlen(o->opath) + strlen(o->name) + 16;
        char *path = (char *)malloc(L);
        int64_t nT; double l_avg;
        snprintf(path, L, "%s%s_index.tsv", o->opath, o->name);
        write_index(path, idxNames, nIdx, nr, avg, &nT, &l_avg);
        if ((cli && (o->mode == IGDC_CREATE_GTYPE0 || o->mode == IGDC_CREATE_BED4)) || py) printf("igd_create 3\n");
        if ((cli && (o->mode == IGDC_CREATE_GTYPE0 || o->mode == IGDC_CREATE_BED4)) || py) printf("igd_create 4\n");
        else if (cli) printf("Save igd database to %s%s.igd\n", o->opath, o->name);
        if (cli) printf("Total intervals, l_avg:  %lld %12.3f\n", (long long)nT, l_avg / nT);
        if (rr) printf("igd_create done!\n");
        free(path);
        PHASE("write _index.tsv");
    }
out:
    igd_hip_created_free(&C);
    free(ctg); free(start); free(end); free(value); free(file); free(poff); free(nr); free(avg);
    for (int32_t f = 0; f < nparts; f++) part_free(&parts[f]);
    free(parts);
    dict_free(&contigs);
    dict_free(&datasets);
    free(ds_nr); free(ds_avg);
    if (globbed) globfree(&g);
    else { for (int32_t i = 0; i < nf; i++) free(files[i]); free(files); }
    return rc;
}

/* `igd create`, src/igd_create.c:436-501 */
int igd_create(int argc, char **argv)
{
    if (argc < 5) {
        fprintf(stderr,
                "%s, v%s\n"
                "usage:   %s create <input dir> <output dir> <output igd name> [options] \n"
                "             -s  <Type of data structure> \n"
                "                   0 for [index, start, end]\n"
                "                   1 for [index, start, end, value], default\n"
                "                   2 for single bed4 file\n"
                "             -b  <Tile size in power of 2 (default 14)> \n"
                "             -f  <input is a file that lists the paths of the bed files> \n",
                "igd", "0.1 (MI355X)", "igd");
        return 0;
    }
    const size_t li = strlen(argv[2]), lo = strlen(argv[3]);
    char *ipath = (char *)malloc(li + 4), *opath = (char *)malloc(lo + 4);
    strcpy(ipath, argv[2]);
    strcpy(opath, argv[3]);
    const char *dbname = argv[4];
    int dtype = 1, ftype = 0;
    int32_t nbp = 16384;
    for (int i = 5; i < argc; i++) {
        if (strcmp(argv[i], "-s") == 0 && i + 1 < argc) dtype = atoi(argv[i + 1]);
        if (strcmp(argv[i], "-b") == 0 && i + 1 < argc) {
            const int n = atoi(argv[i + 1]);
            if (n > 10 && n < 20) nbp = 1 << n;
        }
        if (strcmp(argv[i], "-f") == 0) ftype = 1;
    }
    if (lo == 0 || opath[lo - 1] != '/') strcat(opath, "/");
    if (ftype == 0 && dtype != 2 && li > 0) {
        if (ipath[li - 1] == '/') strcat(ipath, "*");
        else if (ipath[li - 1] != '*') strcat(ipath, "/*");
    }
    const size_t L = strlen(opath) + strlen(dbname) + 8;
    char *probe = (char *)malloc(L);
    snprintf(probe, L, "%s%s.igd", opath, dbname);
    struct stat st;
    if (stat(probe, &st) == 0) printf("The igd database file %s exists!\n", probe);
    else {
        igdc_create_opts o;
        o.ipath = ipath; o.opath = opath; o.name = dbname; o.nbp = nbp;
        o.mode = dtype == 0 ? IGDC_CREATE_GTYPE0 : dtype == 2 ? IGDC_CREATE_BED4 : ftype == 1 ? IGDC_CREATE_LIST : IGDC_CREATE_GLOB;
        o.msg = IGDC_MSG_CLI;
        o.linebuf = dtype == 0 ? 256 : 1024;
        const char *dv = getenv("IGD_DEVICE");
        o.device = dv ? atoi(dv) : 0;
        const int rc = igdc_create(&o);
        if (rc < 0) { free(probe); free(ipath); free(opath); return 1; }
    }
    free(probe); free(ipath); free(opath);
    return 0;
}
