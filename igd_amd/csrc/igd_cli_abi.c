/* igd_cli_abi.c -- the CLI/libigd flavour of the reference ABI (include/igd_search.h,
 * include/igd_base.h) on top of the host core (igd_core.c) and the HIP engine (igd_hip.h).
 *
 * Every function keeps the name, prototype, return value and silent-failure behaviour of
 * its counterpart in /root/reference/src/igd_search.c / igd_base.c (lines cited at each
 * definition).  What differs is how the answer is computed: the whole tile region is put on
 * the GPU once and every call -- single query or query file -- is one batch for the engine.
 * There is no CPU search here.  A library must not end its host process (it may be a Python or R
 * interpreter): if no HIP device is usable, the call prints why on stderr, returns the way the
 * reference's silent failures return (src/igd_search.c:457,462,701-702: 0 / hits untouched) and
 * igd_engine_status() is non-zero from then on; only `igd_search` -- the body of the command line
 * tool -- turns that into a non-zero return value instead of printing a table of zeros.
 */
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdlib.h>
#include <unistd.h>
#include <string.h>
#include <sysexits.h>

#include <time.h>

#include "igd_search.h"
#include "igd_core.h"
#include "igd_create_host.h"
#include "../../include/igd_create.h"

/* IGD_TIMING=1: wall-clock phases of `igd search` on stderr (stdout stays the reference's) */
static double now_s(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
static int timing_on(void) { const char *e = getenv("IGD_TIMING"); return e && *e && *e != '0'; }
static void phase(const char *name, double *t0)
{
    if (!timing_on()) return;
    double t = now_s();
    fprintf(stderr, "[igd timing] %-28s %8.1f ms\n", name, 1e3 * (t - *t0));
    *t0 = t;
}

/* process-wide state of this flavour (reference: src/igd.c:14-19) */
void     *hc = NULL;
iGD_t    *IGD = NULL;
gdata_t  *gData = NULL;
gdata0_t *gData0 = NULL;
int32_t   preIdx = 0, preChr = 0, tile_size = 16384;
FILE     *fP = NULL;

/* our side of an iGD_t handed out by get_igdinfo */
static igdc_db *g_core = NULL;       /* header tables + dictionary + device handle        */
static iGD_t   *g_core_of = NULL;    /* the iGD_t that g_core mirrors                      */
static char    *g_core_path = NULL;

/* The database the search functions work on.  In the reference everything is one program and
 * `IGD` is THE global (src/igd.c:15).  When this library is a shared object and the calling
 * program defines its own `IGD` (as src/igd.c does) without exporting it, the program's variable
 * and ours are different objects and ours stays NULL: then the handle last returned by
 * get_igdinfo() is the database -- which is what the program stored in its `IGD` anyway. */
static iGD_t *cur_igd(void)
{
    return (IGD && IGD == g_core_of) ? IGD : g_core_of;
}

static int device_from_env(void)
{
    const char *e = getenv("IGD_DEVICE");
    return e && *e ? atoi(e) : 0;
}

static int g_fail_rc = 0;            /* code of the first engine failure of this process (0: none) */

int igd_engine_status(void) { return g_fail_rc; }

static void engine_failed(const char *where, int rc)
{
    if (rc == IGD_HIP_ERR_ARG || rc == IGD_HIP_ERR_NOMEM)
        fprintf(stderr, "igd: %s: the GPU engine refused the request (code %d): %s\n", where, rc, igd_hip_last_error());
    else
        fprintf(stderr, "igd: %s: GPU engine unavailable (code %d): %s\n"
                        "igd: this build has no CPU search path.\n", where, rc, igd_hip_last_error());
    if (!g_fail_rc) g_fail_rc = rc ? rc : IGD_HIP_ERR_DEVICE;
}

/* The engine for the current IGD, created at the first search call: this is the moment the
 * reference would do its first fseek/fread on fP (src/igd_search.c:469-476). */
static igd_hip_db *engine(void)
{
    iGD_t *G = cur_igd();
    if (!G || !g_core) {
        fprintf(stderr, "igd: search called before get_igdinfo()\n");
        if (!g_fail_rc) g_fail_rc = IGD_HIP_ERR_ARG;
        return NULL;
    }
    if (g_core->dev && igd_hip_nfiles(g_core->dev) == G->nFiles) return g_core->dev;
    g_core->nFiles = G->nFiles;          /* hits[] is sized from the TSV (:923-925) */
    double t0 = now_s();
    /* the path is preferred when known (pipelined pread + upload); fP serves callers that only
     * opened the stream themselves.  IGD_DEVICES=0,1,..: the database goes to every listed GPU and
     * query files are searched in contiguous slabs, one per device (igdc_search_multi) */
    int devs[IGDC_MAX_DEVICES];
    const int nd = g_core_path ? igdc_devices_from_env(devs, IGDC_MAX_DEVICES) : 0;
    /* (IGD_MULTI_REDUCE=rccl with ONE listed device still goes through the group: a one-rank communicator and all-reduce --
     * how the RCCL call-site is exercised on a one-GPU box, tests/test_gpu_multidev.py) */
    const char *mr = getenv("IGD_MULTI_REDUCE");
    const int group = nd > 1 || (nd == 1 && mr && !strcmp(mr, "rccl"));
    int rc = group ? igdc_attach_path_multi(g_core, g_core_path, devs, nd)
           : g_core_path ? igdc_attach_path(g_core, g_core_path, nd == 1 ? devs[0] : device_from_env())
                         : igdc_attach_fp(g_core, fP, device_from_env());
    if (rc != IGD_HIP_OK) { engine_failed("open", rc); return NULL; }
    phase("database -> GPU", &t0);
    return g_core->dev;
}

/* ------------------------------- base ------------------------------------------------- */
char *parse_bed(char *s, int32_t *st_, int32_t *en_)                /* src/igd_base.c:53-72 */
{
    return igdc_parse_bed(s, st_, en_, 1);
}

int32_t bSearch(gdata_t *g, int32_t t0, int32_t tc, int32_t qe)    /* src/igd_base.c:74-94 */
{
    /* last index in [t0,tc] whose start < qe; -1 when there is none */
    if (tc < t0 || g[t0].start >= qe) return -1;
    int32_t lo = t0, hi = tc;            /* invariant: g[lo].start < qe */
    while (lo < hi) {
        int32_t mid = lo + (hi - lo + 1) / 2;
        if (g[mid].start < qe) lo = mid; else hi = mid - 1;
    }
    return lo;
}

int32_t get_id(const char *chrm)                                   /* src/igd_base.c:325-331 */
{
    return igdc_get_id(hc ? (const igdc_db *)hc : g_core, chrm);
}

info_t *get_fileinfo(char *ifName, int32_t *nFiles)                /* src/igd_base.c:235-267 */
{
    igdc_db tmp;
    memset(&tmp, 0, sizeof tmp);
    if (igdc_load_index(&tmp, ifName) != 0) {
        printf("file not found:%s\n", ifName);
        return NULL;
    }
    info_t *fi = (info_t *)malloc(sizeof(info_t) * (size_t)(tmp.nFiles + 1));
    for (int32_t i = 0; i < tmp.nFiles; i++) {
        fi[i].fileName = tmp.fileName[i];      /* ownership moves to the caller, as strdup'd */
        fi[i].nr = tmp.fileNr[i];
        fi[i].md = tmp.fileMd[i];
    }
    *nFiles = tmp.nFiles;
    free(tmp.fileName); free(tmp.fileNr); free(tmp.fileMd);
    return fi;
}

iGD_t *get_igdinfo(char *igdFile)                                  /* src/igd_base.c:269-323 */
{
    igdc_db *core = igdc_open(igdFile);
    if (!core) {
        printf("Can't open file %s", igdFile);
        return NULL;
    }
    /* hand out the tables in the allocation shape the reference's callers free
     * (src/igd_search.c:1067-1076: nTile, each nCnt[i]/tIdx[i], the arrays, then IGD) */
    iGD_t *g = (iGD_t *)calloc(1, sizeof *g);
    const int32_t m = core->nCtg;
    g->nbp = core->nbp; g->gType = core->gType; g->nCtg = m;
    g->nTile = (int32_t *)malloc(sizeof(int32_t) * (size_t)(m + 1));
    g->nCnt = (int32_t **)malloc(sizeof(int32_t *) * (size_t)(m + 1));
    g->tIdx = (int64_t **)malloc(sizeof(int64_t *) * (size_t)(m + 1));
    g->cName = (char **)malloc(sizeof(char *) * (size_t)(m + 1));
    const int64_t recBytes = core->gType == 0 ? 12 : 16;
    int64_t loc = core->dataOff;                               /* tile offsets: the running sum of src/igd_base.c:288-303 */
    for (int32_t c = 0; c < m; c++) {
        const int32_t k = core->nTile[c];
        g->nTile[c] = k;
        g->nCnt[c] = (int32_t *)malloc(((size_t)k + 1) * sizeof(int32_t));
        g->tIdx[c] = (int64_t *)malloc(((size_t)k + 1) * sizeof(int64_t));
        memcpy(g->nCnt[c], core->nCnt[c], sizeof(int32_t) * (size_t)k);
        g->nCnt[c][k] = 0; g->tIdx[c][k] = 0;
        for (int32_t j = 0; j < k; j++) { g->tIdx[c][j] = loc; loc += recBytes * (int64_t)core->nCnt[c][j]; }
        g->cName[c] = (char *)malloc(40);
        memcpy(g->cName[c], core->cName[c], 40);
    }
    if (g_core) igdc_close(g_core);
    free(g_core_path);
    g_core = core;
    g_core_of = g;
    g_core_path = strdup(igdFile);
    hc = core;                         /* the dictionary lives in the core */
    tile_size = core->nbp;
    return g;
}

/* ------------------------------- one query -------------------------------------------- */
/* Single intervals (`-r`, get_overlaps*): while no engine is resident the host reads the interval's own tiles, like the
 * reference does (:469-476), instead of uploading the whole database for one wave of work (igdc_walk_one, igd_core.h) */
static int one_query_fd(void)
{
    if (g_core && g_core->dev) return -1;                     /* the database is on the GPU already: ask it */
    if (g_core_path) return open(g_core_path, O_RDONLY);
    return fP ? dup(fileno(fP)) : -1;
}

static int32_t one_query(const char *chrm, int32_t qs, int32_t qe, int32_t v, int rule, int64_t *hits)
{
    int32_t ichr = get_id(chrm);
    if (ichr < 0) return 0;                                   /* :456-457 */
    if (g_core && cur_igd()) {
        const int fd = one_query_fd();
        if (fd >= 0) {
            g_core->nFiles = cur_igd()->nFiles;
            const int64_t n = igdc_walk_one(g_core, fd, ichr, qs, qe, v, v != IGD_HIP_NO_VALUE_FILTER, rule, hits, NULL, NULL);
            close(fd);
            if (n >= 0) return (int32_t)n;
        }
    }
    igd_hip_db *dev = engine();
    if (!dev) return 0;
    int64_t total = 0;
    int rc = igd_hip_search(dev, &ichr, &qs, &qe, 1, v, rule, hits, &total);
    if (rc != IGD_HIP_OK) { engine_failed("search", rc); return 0; }
    return (int32_t)total;
}

int32_t get_overlaps(char *chrm, int32_t qs, int32_t qe, int64_t *hits)      /* :454-534 */
{
    one_query(chrm, qs, qe, IGD_HIP_NO_VALUE_FILTER, IGD_HIP_RULE_NEST, hits);
    return 0;                           /* the reference's nols is never incremented (:533) */
}

int32_t get_overlaps0(char *chrm, int32_t qs, int32_t qe, int64_t *hits)     /* :30-112 */
{
    one_query(chrm, qs, qe, IGD_HIP_NO_VALUE_FILTER, IGD_HIP_RULE_NEST, hits);
    return 0;
}

int32_t get_overlaps_v(char *chrm, int32_t qs, int32_t qe, int32_t v, int64_t *hits) /* :623-694 */
{
    /* the driver calls this only for v>0 (:1027); any v keeps the predicate value>=v */
    return one_query(chrm, qs, qe, v, IGD_HIP_RULE_FLAT, hits);
}

/* ------------------------------- query files ------------------------------------------ */
/* the query file is parsed (host threads) while the database goes to the GPU (first search only) */
typedef struct { const char *qFile; igdc_queries q; int rc; } parse_job;
static void *parse_run(void *arg)
{
    parse_job *J = (parse_job *)arg;
    J->rc = igdc_read_queries(g_core, J->qFile, 1, &J->q);
    return NULL;
}

/* a file of at most igdc_host_limit() queries, while no engine is resident: counted on the host (igd_hostpath.c) */
static igdc_map *host_map_lim(int64_t nq, int64_t lim)
{
    if (!g_core || g_core->dev || nq > lim) return NULL;
    const int fd = g_core_path ? open(g_core_path, O_RDONLY) : (fP ? dup(fileno(fP)) : -1);
    if (fd < 0) return NULL;
    g_core->nFiles = cur_igd()->nFiles;          /* hits[] is sized from the TSV (:923-925) */
    igdc_map *m = igdc_map_open(g_core, fd);
    close(fd);
    return m;
}

static int64_t file_query(const char *qFile, int32_t v, int rule, int64_t *hits)
{
    if (!g_core || !cur_igd()) { engine(); return 0; }
    parse_job J;
    J.qFile = qFile; J.rc = -1;
    double t0 = now_s();
    pthread_t th;
    /* a file that is probably small is parsed first and the engine is only started if it turns out not to be */
    const int threaded = !(g_core->dev) && !igdc_host_probably_small(qFile) && pthread_create(&th, NULL, parse_run, &J) == 0;
    if (threaded) { engine(); pthread_join(th, NULL); }
    else parse_run(&J);
    if (J.rc != 0) return 0;                                         /* :701-702 */
    igdc_queries q = J.q;
    phase(threaded ? "database -> GPU  ||  read + parse queries" : "read + parse queries", &t0);
    int64_t total = 0;
    if (q.unsorted && igdc_queries_group_contigs(&q, g_core->nCtg))       /* a sorted BED, chromosomes in another order */
        phase("contig runs put into the database's order", &t0);
    else if (q.unsorted && timing_on())                                   /* one out-of-place line is enough */
        fprintf(stderr, "[igd timing] the query file is not position-sorted: the engine groups it (bucket path)\n");
    igdc_map *hm = q.n > 0 ? host_map_lim(q.n, igdc_host_limit()) : NULL;
    int onHost = 0;
    if (hm) {
        onHost = igdc_search_host(g_core, hm, q.ichr, q.qs, q.qe, q.n, v, rule, hits, &total) == 0;   // (fails only on a read error: hits[] untouched)
        igdc_map_close(hm);
        if (onHost) phase("search on the host (small file)", &t0);
        else total = 0;
    }
    if (!onHost && q.n > 0) {
        igd_hip_db *dev = engine();
        t0 = now_s();
        /* position-sorted BED (the common case): tell the engine, it verifies on the device */
        int rc = !dev ? IGD_HIP_OK
               : g_core->grp ? igdc_search_multi(g_core, q.ichr, q.qs, q.qe, q.n, v, rule, igdc_queries_flags(&q, g_core->nbp), hits, &total)
               : igd_hip_search_ex(dev, q.ichr, q.qs, q.qe, q.n, v, rule, igdc_queries_flags(&q, g_core->nbp), hits, &total);
        if (rc != IGD_HIP_OK) { engine_failed("search", rc); total = 0; }
        phase("search (H2D + kernels + D2H)", &t0);
    }
    igdc_queries_free(&q);
    return total;
}

int64_t getOverlaps(char *qFile, int64_t *hits)                              /* :696-719 */
{
    file_query(qFile, IGD_HIP_NO_VALUE_FILTER, IGD_HIP_RULE_NEST, hits);
    return 0;                           /* sum of get_overlaps returns = 0 */
}

int64_t getOverlaps0(char *qFile, int64_t *hits)                             /* :202-225 */
{
    file_query(qFile, IGD_HIP_NO_VALUE_FILTER, IGD_HIP_RULE_NEST, hits);
    return 0;
}

int64_t getOverlaps_v(char *qFile, int64_t *hits, int32_t v)                 /* :746-769 */
{
    return file_query(qFile, v, IGD_HIP_RULE_FLAT, hits);
}

/* ------------------------------- Seqpare (-s) ------------------------------------------ */
/* seqOverlaps, src/igd_search.c:354-451.  The query file is read like readBED reads it
 * (src/igd_base.c:628-649: parse_bed's accept rule, ailist_add drops uint32 start > end), contigs in
 * first-seen order, each contig's queries ordered by start with ties in file order (the reference's
 * qsort(compare_qstart) is glibc's stable merge sort).  Enumeration, grouping, the greedy matching
 * and the ordered double sums run on the GPU (igd_hip_seqpare); here only sm/(Nq + nr - sm). */
typedef struct { char *name; int32_t id; int32_t *qs, *qe; int64_t n, cap; } sq_ctg;

static void sq_sort(int32_t *qs, int32_t *qe, int64_t n)
{
    if (n < 2) return;
    int32_t *ts = (int32_t *)malloc(sizeof(int32_t) * (size_t)n), *te = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    int32_t *as = qs, *ae = qe, *bs = ts, *be = te;
    for (int64_t w = 1; w < n; w <<= 1) {
        for (int64_t lo = 0; lo < n; lo += 2 * w) {
            const int64_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            int64_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi) {
                if (as[j] < as[i]) { bs[k] = as[j]; be[k++] = ae[j++]; }
                else { bs[k] = as[i]; be[k++] = ae[i++]; }
            }
            while (i < mid) { bs[k] = as[i]; be[k++] = ae[i++]; }
            while (j < hi) { bs[k] = as[j]; be[k++] = ae[j++]; }
        }
        int32_t *x = as; as = bs; bs = x;
        x = ae; ae = be; be = x;
    }
    if (as != qs) { memcpy(qs, as, sizeof(int32_t) * (size_t)n); memcpy(qe, ae, sizeof(int32_t) * (size_t)n); }
    free(ts); free(te);
}

void seqOverlaps(char *qFile, double *sm)
{
    iGD_t *G = cur_igd();
    if (!G) return;
    const int32_t nfiles = G->nFiles;
    for (int32_t m = 0; m < nfiles; m++) sm[m] = 0.0;
    igdc_lines *r = igdc_lines_open(qFile);
    if (!r) return;                                           /* the reference dereferences NULL here */
    sq_ctg *ctg = NULL;
    int32_t nctg = 0, mctg = 0;
    int64_t Nq = 0;
    char *line;
    while ((line = igdc_lines_next(r, NULL)) != NULL) {
        int32_t st, en;
        char *name = igdc_parse_bed(line, &st, &en, 1);
        if (!name || (uint32_t)st > (uint32_t)en) continue;
        int32_t k = nctg - 1;                                  /* BED files are grouped by contig */
        while (k >= 0 && strcmp(ctg[k].name, name) != 0) k--;
        if (k < 0) {
            if (nctg == mctg) { mctg = mctg ? 2 * mctg : 32; ctg = (sq_ctg *)realloc(ctg, sizeof(sq_ctg) * (size_t)mctg); }
            k = nctg++;
            ctg[k].name = strdup(name); ctg[k].id = get_id(name);
            ctg[k].qs = ctg[k].qe = NULL; ctg[k].n = ctg[k].cap = 0;
        }
        sq_ctg *c = &ctg[k];
        if (c->n == c->cap) {
            c->cap = c->cap ? 2 * c->cap : 64;
            c->qs = (int32_t *)realloc(c->qs, sizeof(int32_t) * (size_t)c->cap);
            c->qe = (int32_t *)realloc(c->qe, sizeof(int32_t) * (size_t)c->cap);
        }
        c->qs[c->n] = st; c->qe[c->n] = en; c->n++;
        Nq++;
    }
    igdc_lines_close(r);
    int64_t n = 0;
    for (int32_t k = 0; k < nctg; k++) if (ctg[k].id >= 0) n += ctg[k].n;
    int32_t *ichr = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n ? n : 1)), *qs = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n ? n : 1));
    int32_t *qe = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n ? n : 1)), *grp = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n ? n : 1));
    int64_t at = 0;
    int32_t ng = 0;
    for (int32_t k = 0; k < nctg; k++) {
        if (ctg[k].id < 0) continue;                          /* unknown contig: counts in Nq, overlaps nothing */
        sq_sort(ctg[k].qs, ctg[k].qe, ctg[k].n);
        for (int64_t i = 0; i < ctg[k].n; i++, at++) { ichr[at] = ctg[k].id; qs[at] = ctg[k].qs[i]; qe[at] = ctg[k].qe[i]; grp[at] = ng; }
        ng++;
    }
    double *sums = (double *)calloc((size_t)nfiles + 1, sizeof(double));
    /* The (query contig, dataset) groups are independent, so a query file beyond one engine batch (2^24
     * queries, 2^32 overlaps; the reference has no limit) goes contig range by contig range; the engine
     * continues the running double sums (igd_hip_seqpare_add), so the order of additions stays the reference's. */
    igd_hip_db *dev = n > 0 ? engine() : NULL;
    int failed = n > 0 && !dev;
    for (int32_t g0 = 0; g0 < ng && !failed;) {
        int64_t a0 = 0, a1;
        while (a0 < n && grp[a0] < g0) a0++;
        int32_t g1 = g0;
        a1 = a0;
        while (g1 < ng) {                                      /* as many whole contigs as fit one batch */
            int64_t e = a1;
            while (e < n && grp[e] == g1) e++;
            if (g1 > g0 && e - a0 > igd_hip_max_batch()) break;
            a1 = e; g1++;
        }
        for (;;) {
            for (int64_t i = a0; i < a1; i++) grp[i] -= g0;    /* group numbers of the call start at 0 */
            const int rc = igd_hip_seqpare_add(dev, ichr + a0, qs + a0, qe + a0, a1 - a0, grp + a0, g1 - g0, sums);
            for (int64_t i = a0; i < a1; i++) grp[i] += g0;
            if (rc == IGD_HIP_OK) break;
            if (rc == IGD_HIP_ERR_ARG && g1 - g0 > 1) {          /* too many overlaps for one call: fewer contigs */
                g1 = g0 + (g1 - g0) / 2;
                a1 = a0;
                while (a1 < n && grp[a1] < g1) a1++;
                continue;
            }
            engine_failed("seqpare", rc);
            failed = 1;
            break;
        }
        g0 = g1;
    }
    for (int32_t m = 0; m < nfiles; m++) sm[m] = sums[m] / ((double)Nq + G->finfo[m].nr - sums[m]);   /* :446-449 */
    free(sums); free(ichr); free(qs); free(qe); free(grp);
    for (int32_t k = 0; k < nctg; k++) { free(ctg[k].name); free(ctg[k].qs); free(ctg[k].qe); }
    free(ctg);
}

/* ------------------------------- full enumeration (-f) -------------------------------- */
typedef struct { char *buf; size_t n, cap; } obuf;
static void ob_str(obuf *o, const char *s, size_t L) { memcpy(o->buf + o->n, s, L); o->n += L; }

/* Prints what get_overlaps_f1/_f0 print for each query of the batch, in order
 * ("Query %s, %i, %i: \n" at :548, one "%i\t %i\t %i\t %s\n" per overlap at :577,:610).  The engine
 * streams the overlaps in chunks of contiguous query ranges (igd_hip_enumerate_stream); the text of a
 * chunk (35 bytes per overlap: > 1 GB for 10^6 queries) is formatted by several threads, each into its
 * own buffer for a contiguous range of queries, and written out in order -- while the next chunk is
 * being filled on the GPU and copied over PCIe. */
typedef struct {
    const igdc_queries *q; char **names; const iGD_t *G; const size_t *flen;
    int64_t q0;                       /* first query of the engine call this block belongs to   */
    const int64_t *qoff; const igd_hip_hit *hit;   /* hit[h - hbase] = overlap h of the call    */
    const igd_hip_hit8 *hit8; int bits;            /* ... or the packed stream (8 bytes per overlap; bits = those of idx) */
    int64_t hbase;
    int64_t i0, i1;                   /* queries [i0, i1) of that call                           */
    obuf o;
} fmt_job;

static const char DIGIT2[201] =
    "00010203040506070809101112131415161718192021222324252627282930313233343536373839"
    "40414243444546474849505152535455565758596061626364656667686970717273747576777879"
    "8081828384858687888990919293949596979899";
static inline void ob_uint(obuf *o, uint32_t u)
{
    char t[12];
    int k = 12;
    while (u >= 100) { const uint32_t r = u % 100; u /= 100; k -= 2; memcpy(t + k, DIGIT2 + 2 * r, 2); }
    if (u >= 10) { k -= 2; memcpy(t + k, DIGIT2 + 2 * u, 2); }
    else t[--k] = (char)('0' + u);
    memcpy(o->buf + o->n, t + k, (size_t)(12 - k));
    o->n += (size_t)(12 - k);
}
static inline void ob_int(obuf *o, int32_t x)
{
    if (x < 0) { o->buf[o->n++] = '-'; ob_uint(o, 0u - (uint32_t)x); }
    else ob_uint(o, (uint32_t)x);
}

static void *fmt_run(void *arg)
{
    fmt_job *J = (fmt_job *)arg;
    obuf *o = &J->o;
    const iGD_t *G = J->G;
    for (int64_t i = J->i0; i < J->i1; i++) {
        const int32_t c = J->q->ichr[J->q0 + i], qs = J->q->qs[J->q0 + i], qe = J->q->qe[J->q0 + i];
        const int32_t n1 = qs / G->nbp;
        if (n1 > G->nTile[c] - 1 || n1 < 0) continue;           /* :544-545 */
        const char *nm = J->names[J->q0 + i];
        ob_str(o, "Query ", 6);
        ob_str(o, nm, strlen(nm));
        ob_str(o, ", ", 2); ob_int(o, qs); ob_str(o, ", ", 2); ob_int(o, qe);
        ob_str(o, ": \n", 3);
        uint32_t k = 0;
        if (J->hit8) {                                            /* start | (end - start) << bits | idx: expanded while it is printed */
            const igd_hip_hit8 *h = J->hit8 + (J->qoff[i] - J->hbase), *he = J->hit8 + (J->qoff[i + 1] - J->hbase);
            const int bits = J->bits;
            const uint32_t fmask = bits ? (1u << bits) - 1u : 0u;
            for (; h < he; h++, k++) {
                const uint32_t f = h->lenidx & fmask;
                const int32_t st = (int32_t)h->start, en = (int32_t)(h->start + (h->lenidx >> bits));
                ob_uint(o, k); ob_str(o, "\t ", 2); ob_int(o, st); ob_str(o, "\t ", 2);
                ob_int(o, en); ob_str(o, "\t ", 2); ob_str(o, G->finfo[f].fileName, J->flen[f]); o->buf[o->n++] = '\n';
            }
            continue;
        }
        const igd_hip_hit *h = J->hit + (J->qoff[i] - J->hbase), *he = J->hit + (J->qoff[i + 1] - J->hbase);
        for (; h < he; h++, k++) {
            const int32_t f = h->idx;
            ob_uint(o, k); ob_str(o, "\t ", 2); ob_int(o, h->start); ob_str(o, "\t ", 2);
            ob_int(o, h->end); ob_str(o, "\t ", 2); ob_str(o, G->finfo[f].fileName, J->flen[f]); o->buf[o->n++] = '\n';
        }
    }
    return NULL;
}

typedef struct {
    const igdc_queries *q; char **names; const iGD_t *G; const size_t *flen; size_t maxL;
    int64_t q0; int nt;
    obuf keep[64];                    /* the formatting threads' text buffers, kept from chunk to chunk: fresh 100 MB
                                         allocations per chunk would spend the time in page faults, not in formatting */
} print_ctx;

/* igd_hip_enum_sink / igd_hip_enum_sink8: one chunk = queries [b0,b1) of the call, its overlaps in pinned memory */
static int print_chunk_any(void *ctx, int64_t b0, int64_t b1, const int64_t *qoff, const igd_hip_hit *hit, const igd_hip_hit8 *hit8, int bits);
static int print_chunk(void *ctx, int64_t b0, int64_t b1, const int64_t *qoff, const igd_hip_hit *hit)
{
    return print_chunk_any(ctx, b0, b1, qoff, hit, NULL, 0);
}
static int print_chunk8(void *ctx, int64_t b0, int64_t b1, const int64_t *qoff, const igd_hip_hit8 *hit8, int bits)
{
    return print_chunk_any(ctx, b0, b1, qoff, NULL, hit8, bits);
}
static int print_chunk_any(void *ctx, int64_t b0, int64_t b1, const int64_t *qoff, const igd_hip_hit *hit, const igd_hip_hit8 *hit8, int bits)
{
    print_ctx *P = (print_ctx *)ctx;
    const int nt = P->nt;
    fmt_job job[64];
    pthread_t th[64];
    int started[64];
    int used = 0;
    int64_t i0 = b0;
    for (int t = 0; t < nt && i0 < b1; t++) {                    /* equal shares of the chunk's overlaps (+queries) */
        int64_t i1 = b1;
        if (t + 1 < nt) {
            const int64_t want = qoff[b0] + b0 + (qoff[b1] - qoff[b0] + (b1 - b0)) * (t + 1) / nt;
            int64_t lo = i0, hi = b1;                            /* largest i1 with qoff[i1] + i1 <= want */
            while (lo < hi) {
                const int64_t mid = lo + (hi - lo + 1) / 2;
                if (qoff[mid] + mid <= want) lo = mid; else hi = mid - 1;
            }
            i1 = lo > i0 ? lo : i0 + 1;
        }
        fmt_job *J = &job[used];
        J->q = P->q; J->names = P->names; J->G = P->G; J->flen = P->flen; J->q0 = P->q0; J->qoff = qoff; J->hit = hit; J->hit8 = hit8; J->bits = bits;
        J->hbase = qoff[b0];
        J->i0 = i0; J->i1 = i1;
        const size_t need = (size_t)(i1 - i0) * 96 + (size_t)(qoff[i1] - qoff[i0]) * (40 + P->maxL) + 64;
        if (need > P->keep[used].cap) {
            free(P->keep[used].buf);
            P->keep[used].cap = need + need / 4;
            P->keep[used].buf = (char *)malloc(P->keep[used].cap);
        }
        J->o = P->keep[used];
        J->o.n = 0;
        if (!J->o.buf) { P->keep[used].cap = 0; for (int k = 0; k < used; k++) if (started[k]) pthread_join(th[k], NULL); return 1; }
        /* a thread that cannot be started is simply run here */
        started[used] = used > 0 && pthread_create(&th[used], NULL, fmt_run, J) == 0;
        if (used > 0 && !started[used]) fmt_run(J);
        used++;
        i0 = i1;
    }
    if (used > 0) fmt_run(&job[0]);
    for (int t = 0; t < used; t++) {
        if (started[t]) pthread_join(th[t], NULL);
        if (job[t].o.n) fwrite(job[t].o.buf, 1, job[t].o.n, stdout);
    }
    return 0;
}

static int64_t enumerate_and_print(const igdc_queries *q, char **names)
{
    if (q->n == 0) return 0;
    igdc_map *hm = host_map_lim(q->n, igdc_host_limit_enum());
    igd_hip_db *dev = hm ? NULL : engine();
    if (!hm && !dev) return 0;
    iGD_t *G = cur_igd();
    int64_t *qoff = (int64_t *)malloc(sizeof(int64_t) * (size_t)(q->n + 1));
    int64_t total = 0, grand = 0;
    size_t *flen = (size_t *)malloc(sizeof(size_t) * (size_t)(G->nFiles + 1));
    size_t maxL = 0;
    for (int32_t f = 0; f < G->nFiles; f++) { flen[f] = strlen(G->finfo[f].fileName); if (flen[f] > maxL) maxL = flen[f]; }
    long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    const char *ev = getenv("IGD_PRINT_THREADS");
    int nt = ev && atoi(ev) > 0 ? atoi(ev) : (ncpu > 32 ? 32 : (ncpu < 1 ? 1 : (int)ncpu));
    if (nt > 64) nt = 64;
    const int64_t step = igd_hip_max_batch();
    fflush(stdout);
    print_ctx P;
    P.q = q; P.names = names; P.G = G; P.flen = flen; P.maxL = maxL; P.nt = nt;
    memset(P.keep, 0, sizeof P.keep);
    int onHost = 0;
    if (hm) {                                                    /* small file: the overlaps come from the host, the text as always */
        igd_hip_hit *hit = NULL;
        P.q0 = 0;
        if (igdc_enumerate_host(g_core, hm, q->ichr, q->qs, q->qe, q->n, qoff, &hit, &total) == 0) {
            print_chunk(&P, 0, q->n, qoff, hit);
            grand = total;
            onHost = 1;
        }
        free(hit);
        igdc_map_close(hm);
        if (!onHost) dev = engine();                             /* (a read error on the host path: the engine reads the file its own way) */
    }
    if (!onHost && dev)
    for (int64_t q0 = 0; q0 < q->n; q0 += step) {
        int64_t m = q->n - q0 < step ? q->n - q0 : step;
        P.q0 = q0;
        /* 8 bytes per overlap over PCIe when the database's records fit them (igd_hip_hit8), else 16 */
        int rc = (!getenv("IGD_ENUM_HIT16") && igd_hip_hit8_idx_bits(dev) >= 0)
                     ? igd_hip_enumerate_stream8(dev, q->ichr + q0, q->qs + q0, q->qe + q0, m, qoff, print_chunk8, &P, &total)
                     : igd_hip_enumerate_stream(dev, q->ichr + q0, q->qs + q0, q->qe + q0, m, qoff, print_chunk, &P, &total);
        if (rc != IGD_HIP_OK) { engine_failed("enumerate", rc); break; }
        grand += total;
    }
    for (int t = 0; t < 64; t++) free(P.keep[t].buf);
    free(flen);
    free(qoff);
    return grand;
}

static int64_t file_enumerate(const char *qFile)
{
    if (!g_core || !cur_igd()) { engine(); return 0; }
    igdc_lines *r = igdc_lines_open(qFile);
    if (!r) return 0;
    igdc_queries q;
    memset(&q, 0, sizeof q);
    char **names = NULL;
    int64_t ncap = 0;
    char *line;
    while ((line = igdc_lines_next(r, NULL)) != NULL) {
        int32_t st, en;
        char *chrm = igdc_parse_bed(line, &st, &en, 1);
        if (!chrm) continue;
        int32_t id = igdc_get_id(g_core, chrm);
        if (id < 0) continue;
        if (q.n == ncap) {
            ncap = ncap ? ncap * 2 : 4096;
            names = (char **)realloc(names, sizeof(char *) * (size_t)ncap);
        }
        names[q.n] = g_core->cName[id];    /* the accepted name equals the stored one */
        igdc_queries_push(&q, id, st, en);
    }
    igdc_lines_close(r);
    int64_t total = enumerate_and_print(&q, names);
    igdc_queries_free(&q);
    free(names);
    return total;
}

int64_t getOverlaps_f1(char *qFile) { return file_enumerate(qFile); }        /* :721-744 */
int64_t getOverlaps_f0(char *qFile) { return file_enumerate(qFile); }        /* :227-250 */

typedef struct { int32_t k; const iGD_t *G; } emit_ctx;
static void emit_line(void *ctx, int32_t idx, int32_t start, int32_t end, int32_t in_tile, int32_t tile)
{
    (void)in_tile; (void)tile;
    emit_ctx *E = (emit_ctx *)ctx;
    printf("%i\t %i\t %i\t %s\n", E->k++, start, end, E->G->finfo[idx].fileName);          /* :577,:610 */
}

static int32_t one_enumerate(char *chrm, int32_t qs, int32_t qe)
{
    int32_t ichr = get_id(chrm);
    if (ichr < 0) return 0;
    if (g_core && cur_igd()) {                                /* `-r ... -f`: the interval's own tiles, on the host */
        const iGD_t *G = cur_igd();
        const int32_t n1 = qs / G->nbp;
        const int fd = one_query_fd();
        if (fd >= 0) {
            if (n1 > G->nTile[ichr] - 1 || n1 < 0) { close(fd); return 0; }              /* :544-545 */
            g_core->nFiles = G->nFiles;
            printf("Query %s, %i, %i: \n", chrm, qs, qe);                                 /* :548 */
            emit_ctx E;
            E.k = 0; E.G = G;
            const int64_t n = igdc_walk_one(g_core, fd, ichr, qs, qe, 0, 0, IGD_HIP_RULE_NEST, NULL, emit_line, &E);
            close(fd);
            if (n >= 0) return (int32_t)n;
        }
    }
    igdc_queries q;
    memset(&q, 0, sizeof q);
    igdc_queries_push(&q, ichr, qs, qe);
    char *name = chrm;
    int64_t total = enumerate_and_print(&q, &name);
    igdc_queries_free(&q);
    return (int32_t)total;
}

/* seq_overlaps, src/igd_search.c:253-352: Seqpare's per-query helper -- every overlap of ONE interval appended to the
 * caller's list with its similarity st / (qlen + rlen - st) in single precision (the reference's order of operations) and
 * the reference's identity of a record: (index inside its tile, FIRST tile of the query) -- idx_t = n1 also for records
 * met in later tiles (:291,:337).  One interval: answered on the host from the interval's own tiles, like get_overlaps;
 * rule NEST (everything is nested in `if(tmpi>0)`, :266), 16-byte records (the reference reads gdata_t unconditionally).
 * The list grows by the reference's EXPAND rule (src/igd_base.h:262-265), so `mm` matches as well. */
typedef struct { overlaps_t *olp; float qlen; int32_t qs, qe, n1; int failed; } seq_ctx;
static void emit_seq(void *ctx, int32_t idx, int32_t start, int32_t end, int32_t in_tile, int32_t tile)
{
    seq_ctx *S = (seq_ctx *)ctx;
    overlaps_t *o = S->olp;
    (void)tile;
    if (S->failed) return;
    if (o->nn == o->mm) {
        const int32_t m = o->mm ? o->mm + (2 + o->mm / 8) : 16;
        overlap_t *p = (overlap_t *)realloc(o->olist, sizeof(overlap_t) * (size_t)m);
        if (!p) { S->failed = 1; return; }
        o->olist = p; o->mm = m;
    }
    const float st = (float)((S->qe < end ? S->qe : end) - (S->qs > start ? S->qs : start));
    const float rlen = (float)(end - start);
    overlap_t *p = &o->olist[o->nn++];
    p->idx_g = in_tile; p->idx_f = idx; p->idx_t = S->n1;
    p->sm = st / (S->qlen + rlen - st);
}

void seq_overlaps(char *chrm, int32_t qs, int32_t qe, overlaps_t *olp)
{
    const int32_t ichr = get_id(chrm);
    if (ichr < 0 || !olp || !g_core || !cur_igd()) return;                  /* :257-258 */
    const iGD_t *G = cur_igd();
    if (G->gType == 0) return;                                              /* 12-byte records: the reference would misread them */
    int fd = g_core_path ? open(g_core_path, O_RDONLY) : (fP ? dup(fileno(fP)) : -1);
    if (fd < 0) return;
    seq_ctx S;
    S.olp = olp; S.qlen = (float)(qe - qs); S.qs = qs; S.qe = qe; S.n1 = qs / G->nbp; S.failed = 0;
    g_core->nFiles = G->nFiles > 0 ? G->nFiles : INT32_MAX;  /* (no hits[] is indexed here; a caller may not have read the index file) */
    (void)igdc_walk_one(g_core, fd, ichr, qs, qe, 0, 0, IGD_HIP_RULE_NEST, NULL, emit_seq, &S);
    g_core->nFiles = G->nFiles;
    close(fd);
}

int32_t get_overlaps_f1(char *chrm, int32_t qs, int32_t qe) { return one_enumerate(chrm, qs, qe); } /* :537-620 */
int32_t get_overlaps_f0(char *chrm, int32_t qs, int32_t qe) { return one_enumerate(chrm, qs, qe); } /* :114-200 */

/* ------------------------------- hit map (-m) ----------------------------------------- */
static int64_t hit_map(uint32_t **hitmap, int use_v, int32_t v)
{
    igd_hip_db *dev = engine();
    if (!dev) return 0;
    const int32_t n = cur_igd()->nFiles;
    uint32_t *flat = (uint32_t *)calloc((size_t)n * (size_t)n + 1, sizeof(uint32_t));
    int64_t total = 0;
    int rc = igd_hip_hitmap(dev, use_v, v, flat, &total);
    if (rc != IGD_HIP_OK) { engine_failed("hitmap", rc); free(flat); return 0; }
    for (int32_t a = 0; a < n; a++)
        for (int32_t b = 0; b < n; b++) hitmap[a][b] += flat[(size_t)a * (size_t)n + (size_t)b];
    free(flat);
    /* the reference prints a progress counter every 1000 tiles while it works (:783-784) */
    int64_t tiles = 0;
    for (int32_t c = 0; c < cur_igd()->nCtg; c++) tiles += cur_igd()->nTile[c];
    for (int64_t m = 1000; m <= tiles; m += 1000) printf("%i\n", (int)m);
    return total;
}
int64_t getMap(uint32_t **hitmap) { return hit_map(hitmap, 0, 0); }                  /* :772-826 */
int64_t getMap_v(uint32_t **hitmap, int32_t v) { return hit_map(hitmap, 1, v); }      /* :829-886 */

/* ------------------------------- `igd search` ----------------------------------------- */
static int usage_search(void)
{
    fprintf(stderr,
            "igd (MI355X build), search usage:\n"
            "  igd search <igd database file> [options]\n"
            "    -q <query file>            BED or BED.gz\n"
            "    -r <chrN start end>        a single region\n"
            "    -v <signal value 0-1000>   keep records with value >= v\n"
            "    -f                         print every overlap (with -q or -r)\n"
            "    -m                         dataset x dataset hit map, written to -o <name> (default Hitsmap)\n"
            "    -c                         accepted, no effect\n"
            "    -s                         Seqpare similarity of the query file with every dataset\n"
            "  environment: IGD_DEVICE=<n> selects the GPU (default 0); IGD_DEVICES=0,1,.. searches a query file on\n"
            "               several GPUs (database replicated, contiguous query slabs, per-dataset counts summed)\n");
    return EX_OK;
}

int igd_search(int argc, char **argv)                                        /* :889-1079 */
{
    if (argc < 4) return usage_search();
    char *igdName = argv[2];
    size_t L = strlen(igdName);
    if (L < 4 || strcmp(igdName + L - 4, ".igd") != 0) {                      /* :894-898 */
        printf("%s is not an igd database", igdName);
        return EX_OK;
    }
    FILE *probe = fopen(igdName, "rb");
    if (!probe) {                                                             /* :899-903 */
        printf("%s does not exist", igdName);
        return EX_OK;
    }
    fclose(probe);

    double t0 = now_s();
    IGD = get_igdinfo(igdName);
    if (!IGD) return EX_OK;
    phase("header", &t0);
    char *tsv = igdc_index_path(igdName);
    {   /* fname = path without extension; the reference strcpy's into 64 bytes (:916-922) */
        size_t stem = strlen(tsv) - strlen("_index.tsv");
        if (stem > sizeof IGD->fname - 1) stem = sizeof IGD->fname - 1;
        memcpy(IGD->fname, igdName, stem);
        IGD->fname[stem] = '\0';
    }
    IGD->finfo = get_fileinfo(tsv, &IGD->nFiles);
    free(tsv);
    if (!IGD->finfo) return EX_OK;
    const int32_t nfiles = IGD->nFiles;
    int64_t *hits = (int64_t *)calloc((size_t)nfiles + 1, sizeof(int64_t));

    int32_t v = 0, qs = 1, qe = 2;
    int mode = -1, full = 0;
    char *chrm = NULL, *qfName = (char *)"";
    char out[64] = "";
    for (int i = 3; i < argc; i++) {                                          /* :931-971 */
        const char *a = argv[i];
        if (strcmp(a, "-q") == 0) {
            if (i + 1 >= argc) { printf("No query file.\n"); return EX_OK; }
            qfName = argv[i + 1];
            mode = 1;
        } else if (strcmp(a, "-r") == 0) {
            if (i + 3 < argc) {
                mode = 2;
                chrm = argv[i + 1];
                qs = atoi(argv[i + 2]);
                qe = atoi(argv[i + 3]);
            }
        } else if (strcmp(a, "-v") == 0) {
            if (i + 1 < argc) v = atoi(argv[i + 1]);
        } else if (strcmp(a, "-m") == 0) {
            mode = 0;
        } else if (strcmp(a, "-s") == 0 && mode != 2) {
            mode = 3;
        } else if (strcmp(a, "-f") == 0) {
            full = 1;
        } else if (strcmp(a, "-o") == 0) {
            if (i + 1 < argc) { strncpy(out, argv[i + 1], sizeof out - 1); out[sizeof out - 1] = '\0'; }
        }
    }

    fP = fopen(igdName, "rb");                                                /* :974 */
    if (full) {                                                               /* :975-995 */
        if (mode == 1) {
            int64_t total = IGD->gType == 0 ? getOverlaps_f0(qfName) : getOverlaps_f1(qfName);
            if (!g_fail_rc) printf("Total overlaps: %lld\n", (long long)total);
        } else if (mode == 2) {
            int64_t total = IGD->gType == 0 ? get_overlaps_f0(chrm, qs, qe) : get_overlaps_f1(chrm, qs, qe);
            if (!g_fail_rc) printf("Total overlaps: %lld\n", (long long)total);
        } else {
            printf("Not supported -f option\n");
            return EX_OK;
        }
    } else if (mode == 1) {                                                   /* :1023-1040 */
        if (IGD->gType == 0) getOverlaps0(qfName, hits);
        else if (v > 0) getOverlaps_v(qfName, hits, v);
        else getOverlaps(qfName, hits);
        if (!g_fail_rc) printf("index\t number of regions\t number of hits\t File_name\n");
        int64_t total = 0;
        for (int32_t i = 0; i < nfiles && !g_fail_rc; i++) {
            if (hits[i] > 0)
                printf("%i\t%i\t%lld\t%s\n", i, IGD->finfo[i].nr, (long long)hits[i], IGD->finfo[i].fileName);
            total += hits[i];
        }
        if (!g_fail_rc) printf("Total: %lld\n", (long long)total);
    } else if (mode == 2) {                                                   /* :1041-1053 */
        if (IGD->gType == 0) get_overlaps0(chrm, qs, qe, hits);
        else if (v > 0) get_overlaps_v(chrm, qs, qe, v, hits);
        else get_overlaps(chrm, qs, qe, hits);
        if (!g_fail_rc) printf("index\t number of regions\t number of hits\t File_name\n");
        for (int32_t i = 0; i < nfiles && !g_fail_rc; i++)
            printf("%i\t%i\t%lld\t%s\n", i, IGD->finfo[i].nr, (long long)hits[i], IGD->finfo[i].fileName);
    } else if (mode == 0) {                                                   /* :996-1022 */
        if (IGD->gType != 1) {
            printf("igd: -m needs a database created with 16-byte records (gType 1)\n");
        } else {
            uint32_t **hitmap = (uint32_t **)malloc(sizeof(uint32_t *) * (size_t)(nfiles + 1));
            for (int32_t i = 0; i < nfiles; i++) hitmap[i] = (uint32_t *)calloc((size_t)nfiles + 1, sizeof(uint32_t));
            if (v > 0) getMap_v(hitmap, v); else getMap(hitmap);
            if (strlen(out) < 2) strcpy(out, "Hitsmap");
            FILE *fo = g_fail_rc ? NULL : fopen(out, "w");
            if (g_fail_rc) ;                                /* no matrix of zeros after an engine failure */
            else if (!fo) printf("Can't open file %s\n", out);
            else {
                static char obuf[1 << 20];
                setvbuf(fo, obuf, _IOFBF, sizeof obuf);
                fprintf(fo, "%u\t%u\t%u\n", (unsigned)nfiles, (unsigned)nfiles, (unsigned)v);
                for (int32_t i = 0; i < nfiles; i++) {
                    for (int32_t j = 0; j < nfiles; j++) fprintf(fo, "%u\t", hitmap[i][j]);
                    fprintf(fo, "\n");
                }
                fclose(fo);
            }
            for (int32_t i = 0; i < nfiles; i++) free(hitmap[i]);
            free(hitmap);
        }
    } else if (mode == 3) {                                                   /* :1054-1061 */
        if (IGD->gType != 1) printf("igd: -s needs a database created with 16-byte records (gType 1)\n");
        else {
            double *sm = (double *)malloc(sizeof(double) * (size_t)(nfiles + 1));
            seqOverlaps(qfName, sm);
            if (!g_fail_rc) printf("index\t number of regions\t similarity\t dataset name\n");
            for (int32_t i = 0; i < nfiles && !g_fail_rc; i++)
                printf("%i\t%i\t%10.6f\t%s\n", i, IGD->finfo[i].nr, sm[i], IGD->finfo[i].fileName);
            free(sm);
        }
    } else {
        free(hits);
        return usage_search();
    }

    /* release everything, in the shape get_igdinfo/get_fileinfo handed it out (:1066-1078) */
    if (fP) { fclose(fP); fP = NULL; }
    free(IGD->nTile);
    for (int32_t c = 0; c < IGD->nCtg; c++) {
        free(IGD->nCnt[c]); free(IGD->tIdx[c]); free(IGD->cName[c]);
    }
    for (int32_t i = 0; i < nfiles; i++) free(IGD->finfo[i].fileName);
    free(IGD->nCnt); free(IGD->tIdx); free(IGD->cName); free(IGD->finfo);
    free(IGD);
    IGD = NULL;
    free(hits);
    if (g_core) { igdc_close(g_core); g_core = NULL; g_core_of = NULL; hc = NULL; }
    free(g_core_path); g_core_path = NULL;
    /* the reference's exit code is always 0; an engine failure (message already on stderr) is the one
     * thing this tool reports through it, instead of printing a table of zeros */
    return g_fail_rc ? EX_UNAVAILABLE : EX_OK;
}

/* ---- create_igd*, src/igd_create.h:10-14 --------------------------------------------------- */
static void create_with(char *iPath, char *oPath, char *igdName, int mode)
{
    igdc_create_opts o;
    o.ipath = iPath; o.opath = oPath; o.name = igdName;
    o.nbp = tile_size > 0 ? tile_size : 16384;            /* igd_init, src/igd_base.c:522 */
    o.mode = mode; o.msg = IGDC_MSG_CLI;
    o.linebuf = mode == IGDC_CREATE_GTYPE0 ? 256 : 1024;
    const char *dv = getenv("IGD_DEVICE");
    o.device = dv ? atoi(dv) : 0;
    const int rc = igdc_create(&o);
    if (rc < 0) {
        fprintf(stderr, "igd create: no usable GPU (%d): %s\n", rc, igd_hip_last_error());
        if (!g_fail_rc) g_fail_rc = rc;
    }
}
void create_igd(char *iPath, char *oPath, char *igdName) { create_with(iPath, oPath, igdName, IGDC_CREATE_GLOB); }
void create_igd0(char *iPath, char *oPath, char *igdName) { create_with(iPath, oPath, igdName, IGDC_CREATE_GTYPE0); }
void create_igd_f(char *iPath, char *oPath, char *igdName) { create_with(iPath, oPath, igdName, IGDC_CREATE_LIST); }
void create_igd_bed4(char *iPath, char *oPath, char *igdName) { create_with(iPath, oPath, igdName, IGDC_CREATE_BED4); }
