/* igd_hostpath.c -- SMALL query files answered on the host (product code; nothing of oracle/ is used or linked).
 *
 * Why this exists.  The reference starts cheaply: get_igdinfo reads the header only (src/igd_base.c:269-323) and
 * get_overlaps freads just the tiles a query touches (src/igd_search.c:469-476), so `igd search -q` on a file of 10^3
 * queries takes it 6 ms.  Bringing up the HIP runtime and uploading the whole tile region costs a fixed ~0.18 s
 *  whatever the file holds -- 30 x the reference's time at 10^3 queries.  The GPU pays off
 * above ~10^5 queries.  So the three flavours' query-FILE entry points (getOverlaps*, search_n, `-q`, `-f`) count
 * files of at most igdc_host_limit() queries here, like the single-interval entry points already do
 * (igdc_walk_one, igd_core.c), and take the engine for everything larger.
 *
 * What this is NOT: a fallback.  The choice depends on the number of queries and the host's thread count
 * (igdc_host_limit: the same file can take different paths on different hosts, with identical output), never on whether
 * a GPU is present; a file above the limit has no CPU path and fails loudly without a usable device (tests/test_host.py),
 * IGD_HOST_MAX_QUERIES=0 sends every file to the engine (all `-m gpu` parity tests run that way), and the engine's
 * own entry points (include/igd_hip.h, igd_amd.Database, bench.py) never come here.
 *
 * Algorithm = the reference's per query (src/igd_search.c:454-534 rule NEST, :623-694 rule FLAT): in every visited
 * tile bisect the start-sorted records for the last one with start < qe (:479-487 / bSearch src/igd_base.c:74-94),
 * walk back from there testing end > qs [&& value >= v] (:489-493), in later tiles stop at the first record that
 * starts before the tile (:510-511).  Tiles are read with pread() into per-thread buffers (igdc_map holds the file
 * descriptor only; the page cache serves repeats), the
 * queries are split into contiguous ranges over a few threads with private hits[] that are added up at the end
 * (hits[] is a sum over queries).
 */
#define _GNU_SOURCE
#include "igd_core.h"

#include <fcntl.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

/* Where the engine takes over.  The host route exists for SMALL query files -- a search of 10^3 .. 10^5 lines should not pay
 * the accelerator's start-up (HIP runtime + 851 MB through PCIe: 0.18 s at best, 0.4-0.5 s in many runs, LABNOTES R5-3) --
 * not to carry the product's headline workloads on CPU threads.  Round 6: the limit is back at
 * the conservative round-4 value, 25 000 queries per usable thread (<= 400 000), the same as `-f`'s.  Round 5's model
 * (engine F + n * 7 ns against host 0.03 s + n * 0.55 us / T, crossing at ~280 000 * T) assumed the engine route's BEST start-up
 * and threads that scale; on a host where they do not (1.4 us per query whatever T) it sent 2e6-query files to a 2.8 s host
 * route.  With this limit the worst case of the model's error is ~0.1 s either way, and a 10^6-query file -- BASELINE config 2
 * -- is counted by the MI355X.  IGD_HOST_MAX_QUERIES overrides it (0: every file goes to the GPU).  This file is frozen:
 * bench.py times every size on BOTH routes and labels who counted. */
static int64_t host_threads_usable(void)
{
    long t = sysconf(_SC_NPROCESSORS_ONLN);
    if (t > 16) t = 16;
    if (t < 2) t = 2;
    return t;
}
static int64_t host_limit_env(void)
{
    const char *e = getenv("IGD_HOST_MAX_QUERIES");
    if (e && *e) {
        const long long x = atoll(e);
        return x < 0 ? 0 : (int64_t)x;
    }
    return -1;
}
int64_t igdc_host_limit(void)
{
    const int64_t e = host_limit_env();
    if (e >= 0) return e;
    return 25000 * host_threads_usable();
}
int64_t igdc_host_limit_enum(void)
{
    const int64_t e = host_limit_env();
    return e >= 0 ? e : 25000 * host_threads_usable();
}

/* a query file of `bytes` bytes can be expected to hold at most the limit's number of lines (a BED3 line of a human
 * genome is 18-26 bytes; gzip shrinks it about four times): parse it BEFORE the engine is started, then decide */
int igdc_host_probably_small(const char *qfile)
{
    const int64_t lim = igdc_host_limit();
    if (lim <= 0 || !qfile) return 0;
    struct stat st;
    if (stat(qfile, &st) != 0 || !S_ISREG(st.st_mode)) return 0;
    const size_t L = strlen(qfile);
    const int gz = L > 3 && strcmp(qfile + L - 3, ".gz") == 0;
    return (int64_t)st.st_size <= lim * (gz ? 6 : 24);
}

/* The tile region is read with pread into a per-thread buffer that keeps the last tile (position-sorted query files ask
 * for the same tile again and again; the reference keeps one tile too, src/igd_search.c:469-476).  Measured against a
 * read-only mapping of the file: the mapping pays a page fault per touched 4 KiB page -- 49 ms for 10^4 queries on five
 * threads, which contend for the address space's lock -- where one pread per tile pays a system call (8 ms). */
struct igdc_map {
    int fd;
    int64_t bytes;
};

igdc_map *igdc_map_open(const igdc_db *db, int fd)
{
    if (!db || fd < 0) return NULL;
    const int64_t recBytes = db->gType == 0 ? 12 : 16;
    const int64_t need = db->dataOff + recBytes * db->nRecords;
    struct stat st;
    if (fstat(fd, &st) != 0 || (int64_t)st.st_size < need || need <= 0) return NULL;
    igdc_map *m = (igdc_map *)malloc(sizeof *m);
    if (!m) return NULL;
    m->fd = dup(fd);
    m->bytes = need;
    if (m->fd < 0) { free(m); return NULL; }
    return m;
}

void igdc_map_close(igdc_map *m)
{
    if (!m) return;
    close(m->fd);
    free(m);
}

typedef struct { int32_t *buf; size_t cap; int32_t ichr, tile; int failed; } tilebuf;
static const int32_t *tile_records(const igdc_db *db, const igdc_map *m, tilebuf *tb, int32_t ichr, int32_t j, int32_t cnt)
{
    if (tb->ichr == ichr && tb->tile == j) return tb->buf;
    const size_t bytes = (size_t)cnt * (db->gType == 0 ? 12u : 16u);
    if (bytes > tb->cap) {
        free(tb->buf);
        tb->cap = bytes + bytes / 2 + 4096;
        tb->buf = (int32_t *)malloc(tb->cap);
        if (!tb->buf) { tb->cap = 0; tb->failed = 1; tb->ichr = -1; return NULL; }
    }
    size_t done = 0;
    const int64_t off = igdc_tile_off(db, ichr, j);
    while (done < bytes) {
        const ssize_t got = pread(m->fd, (char *)tb->buf + done, bytes - done, (off_t)(off + (int64_t)done));
        if (got <= 0) { tb->failed = 1; tb->ichr = -1; return NULL; }
        done += (size_t)got;
    }
    tb->ichr = ichr; tb->tile = j;
    return tb->buf;
}

/* number of records of a start-sorted tile with start < qe (the reference's bisections, :479-487, bSearch) */
static inline int32_t below(const int32_t *rec, int w, int32_t cnt, int32_t qe)
{
    int32_t lo = 0, hi = cnt;
    while (lo < hi) {
        const int32_t mid = lo + ((hi - lo) >> 1);
        if (rec[(size_t)mid * (size_t)w + 1] < qe) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/* one query; hits may be NULL (count only); emit appends (idx,start,end) triples in the reference's -f order */
typedef struct { igd_hip_hit *v; int64_t n, cap; int failed; } hitvec;
static inline void hv_push(hitvec *o, int32_t q, const int32_t *r)
{
    if (o->n == o->cap) {
        const int64_t nc = o->cap ? 2 * o->cap : 4096;
        igd_hip_hit *nv = (igd_hip_hit *)realloc(o->v, sizeof(igd_hip_hit) * (size_t)nc);
        if (!nv) { o->failed = 1; return; }
        o->v = nv; o->cap = nc;
    }
    igd_hip_hit *h = &o->v[o->n++];
    h->q = q; h->idx = r[0]; h->start = r[1]; h->end = r[2];
}

static inline int64_t one_query(const igdc_db *db, const igdc_map *m, tilebuf *tb, int32_t ichr, int32_t qs, int32_t qe, int32_t v,
                                int use_v, int rule, int64_t *hits, hitvec *emit, int32_t qno)
{
    if (ichr < 0 || ichr >= db->nCtg) return 0;                            /* :456-457 */
    const int32_t nbp = db->nbp, mT = db->nTile[ichr] - 1;
    const int32_t n1 = qs / nbp;                                           /* C division, as :459 */
    int32_t n2 = (int32_t)((uint32_t)qe - 1u) / nbp;                       /* (qe-1)/nbp with the reference's wrap */
    if (n1 < 0 || n1 > mT) return 0;
    if (n2 > mT) n2 = mT;
    if (rule == IGD_HIP_RULE_NEST && db->nCnt[ichr][n1] == 0) return 0;    /* :468 */
    const int w = db->gType == 0 ? 3 : 4;
    const int32_t nf = db->nFiles;
    int64_t total = 0;
    for (int32_t j = n1; j <= (n2 > n1 ? n2 : n1); j++) {
        const int32_t cnt = db->nCnt[ichr][j];
        if (cnt <= 0) continue;
        const int32_t *rec = tile_records(db, m, tb, ichr, j, cnt);
        if (!rec) return total;
        const int64_t lob = j == n1 ? INT64_MIN : (int64_t)(int32_t)((uint32_t)nbp * (uint32_t)j);
        for (int32_t i = below(rec, w, cnt, qe) - 1; i >= 0; i--) {        /* the reverse scans of :489-493, :522-526 */
            const int32_t *r = rec + (size_t)i * (size_t)w;
            if ((int64_t)r[1] < lob) break;                                /* met in an earlier tile (:510-511) */
            if (r[2] > qs && (!use_v || r[3] >= v)) {
                if (r[0] < 0 || r[0] >= nf) continue;                      /* the reference indexes hits[] unchecked (:491) */
                if (hits) hits[r[0]]++;
                if (emit) hv_push(emit, qno, r);
                total++;
            }
        }
    }
    return total;
}

typedef struct {
    const igdc_db *db; const igdc_map *m;
    const int32_t *ichr, *qs, *qe;
    int64_t lo, hi;
    int32_t v; int use_v, rule;
    int64_t *hits, total;
    int64_t *qcnt;          /* enumeration: per-query counts (may be NULL) */
    hitvec out; int want_out;
    int io_failed;
} host_job;

static void *host_run(void *arg)
{
    host_job *J = (host_job *)arg;
    int64_t tot = 0;
    tilebuf tb;
    memset(&tb, 0, sizeof tb);
    tb.ichr = -1;
    for (int64_t i = J->lo; i < J->hi; i++) {
        const int64_t n = one_query(J->db, J->m, &tb, J->ichr[i], J->qs[i], J->qe[i], J->v, J->use_v, J->rule, J->hits,
                                    J->want_out ? &J->out : NULL, (int32_t)i);
        if (J->qcnt) J->qcnt[i] = n;
        tot += n;
    }
    J->total = tot;
    J->io_failed = tb.failed;
    free(tb.buf);
    return NULL;
}

static int host_threads(int64_t nq)
{
    const char *e = getenv("IGD_HOST_THREADS");
    long t = e && *e ? atol(e) : 0;
    if (t <= 0) {
        t = sysconf(_SC_NPROCESSORS_ONLN);
        if (t > 16) t = 16;
    }
    const int64_t byWork = (nq + 2047) / 2048;      /* a thread is worth starting for ~2000 queries (~3 ms of counting) */
    if (t > byWork) t = (long)byWork;
    if (t > 64) t = 64;
    return t < 1 ? 1 : (int)t;
}

static int run_jobs(host_job *job, int T)
{
    pthread_t th[64];
    int started[64];
    for (int k = 1; k < T; k++) {
        started[k] = pthread_create(&th[k], NULL, host_run, &job[k]) == 0;
        if (!started[k]) host_run(&job[k]);
    }
    host_run(&job[0]);
    for (int k = 1; k < T; k++) if (started[k]) pthread_join(th[k], NULL);
    return 0;
}

int igdc_search_host(const igdc_db *db, const igdc_map *m, const int32_t *ichr, const int32_t *qs, const int32_t *qe,
                     int64_t nq, int32_t v, int rule, int64_t *hits, int64_t *total)
{
    if (!db || !m || !hits || nq < 0) return -1;
    const int use_v = v != IGD_HIP_NO_VALUE_FILTER && db->gType == 1;      /* gType 0 stores no value (:1024-1025) */
    const int T = host_threads(nq);
    host_job job[64];
    /* every thread counts into a private vector; they are added to the caller's hits[] only if every tile could be read */
    int64_t *priv = (int64_t *)calloc((size_t)T * (size_t)(db->nFiles + 1), sizeof(int64_t));
    if (!priv) return -1;
    for (int k = 0; k < T; k++) {
        memset(&job[k], 0, sizeof job[k]);
        job[k].db = db; job[k].m = m; job[k].ichr = ichr; job[k].qs = qs; job[k].qe = qe;
        job[k].lo = nq * k / T; job[k].hi = nq * (k + 1) / T;
        job[k].v = v; job[k].use_v = use_v; job[k].rule = rule;
        job[k].hits = priv + (size_t)k * (size_t)(db->nFiles + 1);
    }
    run_jobs(job, T);
    int64_t tot = 0;
    int bad = 0;
    for (int k = 0; k < T; k++) bad |= job[k].io_failed;
    for (int k = 0; k < T; k++) {
        tot += job[k].total;
        if (!bad) for (int32_t f = 0; f < db->nFiles; f++) hits[f] += job[k].hits[f];
    }
    free(priv);
    if (bad) return -1;
    if (total) *total = tot;
    return 0;
}

int igdc_enumerate_host(const igdc_db *db, const igdc_map *m, const int32_t *ichr, const int32_t *qs, const int32_t *qe,
                        int64_t nq, int64_t *qoff, igd_hip_hit **out, int64_t *total)
{
    if (!db || !m || !qoff || !out || nq < 0 || nq > INT32_MAX) return -1;
    *out = NULL;
    const int T = host_threads(nq);
    host_job job[64];
    for (int k = 0; k < T; k++) {
        memset(&job[k], 0, sizeof job[k]);
        job[k].db = db; job[k].m = m; job[k].ichr = ichr; job[k].qs = qs; job[k].qe = qe;
        job[k].lo = nq * k / T; job[k].hi = nq * (k + 1) / T;
        job[k].v = IGD_HIP_NO_VALUE_FILTER; job[k].use_v = 0; job[k].rule = IGD_HIP_RULE_NEST;    /* -f: rule NEST, no filter */
        job[k].qcnt = qoff;                 /* counts first, turned into offsets below */
        job[k].want_out = 1;
    }
    run_jobs(job, T);
    int64_t tot = 0;
    int failed = 0;
    for (int k = 0; k < T; k++) { tot += job[k].out.n; failed |= job[k].out.failed | job[k].io_failed; }
    igd_hip_hit *all = failed ? NULL : (igd_hip_hit *)malloc(sizeof(igd_hip_hit) * (size_t)(tot ? tot : 1));
    if (!all) failed = 1;
    int64_t at = 0;
    for (int k = 0; k < T; k++) {
        if (!failed && job[k].out.n) memcpy(all + at, job[k].out.v, sizeof(igd_hip_hit) * (size_t)job[k].out.n);
        at += job[k].out.n;
        free(job[k].out.v);
    }
    if (failed) return -1;
    int64_t run = 0;
    for (int64_t i = 0; i < nq; i++) { const int64_t c = qoff[i]; qoff[i] = run; run += c; }
    qoff[nq] = run;
    *out = all;
    if (total) *total = tot;
    return 0;
}

/* The handle flavours' batches (Python search_n / search_1, R search_nr / getOverlaps): on the host while the batch is
 * small and no engine is resident, otherwise on the engine, which is attached at the first batch that needs it -- the
 * moment the reference would do its first fseek/fread (src/igd_search.c:469-476); open_iGD reads the header only, like
 * the reference's (src_py/igd_base.c, IGDr/src/igd_base.c open_iGD). */
int igdc_search_auto(igdc_db *db, const char *path, int device, const int32_t *ichr, const int32_t *qs, const int32_t *qe,
                     int64_t nq, int32_t v, int rule, int flags, int64_t *hits, int64_t *total)
{
    if (total) *total = 0;
    if (!db || !hits || nq < 0) return IGD_HIP_ERR_ARG;
    if (nq == 0) return IGD_HIP_OK;
    if (!db->dev && nq <= igdc_host_limit() && path) {
        const int fd = open(path, O_RDONLY);
        igdc_map *m = fd >= 0 ? igdc_map_open(db, fd) : NULL;
        if (fd >= 0) close(fd);
        if (m) {
            const int rc = igdc_search_host(db, m, ichr, qs, qe, nq, v, rule, hits, total);
            igdc_map_close(m);
            if (rc == 0) return IGD_HIP_OK;
        }
    }
    if (!db->dev) {
        const int rc = path ? igdc_attach_path(db, path, device) : IGD_HIP_ERR_ARG;
        if (rc != IGD_HIP_OK) return rc;
    }
    return igd_hip_search_ex(db->dev, ichr, qs, qe, nq, v, rule, flags, hits, total);
}
