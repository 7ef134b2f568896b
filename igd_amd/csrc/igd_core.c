/* igd_core.c -- host core: .igd / _index.tsv loaders, contig dictionary, BED reader,
 * GPU attach, minimal .igd writer.  See igd_core.h for the reference counterparts.
 * No search is done here: every overlap count comes from the HIP engine (igd_hip.h). */
#define _GNU_SOURCE
#include "igd_core.h"

#include <errno.h>
#include <fcntl.h>
#include <limits.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

/* ---------------------------------------------------------------------------------------
 * contig dictionary: exact, case-sensitive name -> index (what the reference gets from
 * khash, src/igd_base.c:313-331).  FNV-1a + linear probing; names are the NUL-terminated
 * prefix of the 40-byte on-disk field (bytes after the NUL are garbage in real files). */
static uint32_t name_hash(const char *s)
{
    uint32_t h = 2166136261u;
    while (*s) h = (h ^ (unsigned char)*s++) * 16777619u;
    return h;
}

static int dict_build(igdc_db *db)
{
    int32_t cap = 8;
    while (cap < 4 * (db->nCtg + 1)) cap <<= 1;
    db->dict = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
    if (!db->dict) return -1;
    db->dictCap = cap;
    for (int32_t i = 0; i < cap; i++) db->dict[i] = -1;
    for (int32_t c = 0; c < db->nCtg; c++) {
        uint32_t p = name_hash(db->cName[c]) & (uint32_t)(cap - 1);
        while (db->dict[p] >= 0 && strcmp(db->cName[db->dict[p]], db->cName[c]) != 0)
            p = (p + 1) & (uint32_t)(cap - 1);
        db->dict[p] = c;   /* a repeated name resolves to its LAST index, as kh_put+kh_val= does */
    }
    return 0;
}

int32_t igdc_get_id(const igdc_db *db, const char *chrm)
{
    if (!db || !db->dict || !chrm) return -1;
    uint32_t p = name_hash(chrm) & (uint32_t)(db->dictCap - 1);
    for (;;) {
        int32_t c = db->dict[p];
        if (c < 0) return -1;
        if (strcmp(db->cName[c], chrm) == 0) return c;
        p = (p + 1) & (uint32_t)(db->dictCap - 1);
    }
}

/* --------------------------------------------------------------------------------------- */
static int read_exact(FILE *fp, void *dst, size_t bytes)
{
    return bytes == 0 || fread(dst, 1, bytes, fp) == bytes ? 0 : -1;
}

igdc_db *igdc_open(const char *igd_path)
{
    FILE *fp = fopen(igd_path, "rb");
    if (!fp) return NULL;
    igdc_db *db = (igdc_db *)calloc(1, sizeof *db);
    if (!db) { fclose(fp); return NULL; }
    int32_t head[3];
    if (read_exact(fp, head, sizeof head) != 0) goto bad;
    db->nbp = head[0]; db->gType = head[1]; db->nCtg = head[2];
    if (db->nbp <= 0 || db->nCtg < 0 || (db->gType != 0 && db->gType != 1)) goto bad;
    const int32_t m = db->nCtg;
    db->nTile = (int32_t *)calloc((size_t)m + 1, sizeof(int32_t));
    if (!db->nTile || read_exact(fp, db->nTile, sizeof(int32_t) * (size_t)m) != 0) goto bad;
    int64_t nT = 0;
    for (int32_t c = 0; c < m; c++) {
        if (db->nTile[c] < 0) goto bad;
        nT += db->nTile[c];
    }
    db->nTileTotal = nT;
    db->dataOff = 12 + 44 * (int64_t)m + 4 * nT;
    db->nCntFlat = (int32_t *)calloc((size_t)nT + 1, sizeof(int32_t));
    db->tBase = (int64_t *)calloc((size_t)(nT >> 6) + 2, sizeof(int64_t));
    db->nCnt = (int32_t **)calloc((size_t)m + 1, sizeof(int32_t *));
    db->cName = (char **)calloc((size_t)m + 1, sizeof(char *));
    if (!db->nCntFlat || !db->tBase || !db->nCnt || !db->cName) goto bad;
    if (read_exact(fp, db->nCntFlat, sizeof(int32_t) * (size_t)nT) != 0) goto bad;
    const int64_t recBytes = db->gType == 0 ? 12 : 16;
    int64_t loc = db->dataOff, t = 0;
    for (int32_t c = 0; c < m; c++) {
        db->nCnt[c] = db->nCntFlat + t;
        for (int32_t j = 0; j < db->nTile[c]; j++, t++) {
            if (db->nCntFlat[t] < 0) goto bad;
            if ((t & 63) == 0) db->tBase[t >> 6] = loc;
            loc += recBytes * (int64_t)db->nCntFlat[t];
            db->nRecords += db->nCntFlat[t];
        }
    }
    for (int32_t c = 0; c < m; c++) {
        db->cName[c] = (char *)calloc(1, 41);
        if (!db->cName[c] || read_exact(fp, db->cName[c], 40) != 0) goto bad;
    }
    fclose(fp);
    if (dict_build(db) != 0) { igdc_close(db); return NULL; }
    return db;
bad:
    fclose(fp);
    igdc_close(db);
    return NULL;
}

char *igdc_index_path(const char *igd_path)
{
    size_t L = strlen(igd_path);
    char *p = (char *)malloc(L + 16);
    if (!p) return NULL;
    memcpy(p, igd_path, L + 1);
    char *dot = strrchr(p, '.');
    if (dot) *dot = '\0';
    strcat(p, "_index.tsv");
    return p;
}

/* one dataset per line after the header line: index \t name \t nr \t avg  (nr and avg read
 * with atol, so avg loses its fraction: src/igd_base.c:258-261) */
int igdc_load_index(igdc_db *db, const char *tsv_path)
{
    FILE *fp = fopen(tsv_path, "r");
    if (!fp) return -1;
    char line[1024];
    int32_t n = 0;
    if (!fgets(line, sizeof line, fp)) { fclose(fp); return -1; }
    while (fgets(line, sizeof line, fp)) n++;
    for (int32_t i = 0; i < db->nFiles; i++) free(db->fileName[i]);
    free(db->fileName); free(db->fileNr); free(db->fileMd);
    db->fileName = (char **)calloc((size_t)n + 1, sizeof(char *));
    db->fileNr = (int32_t *)calloc((size_t)n + 1, sizeof(int32_t));
    db->fileMd = (double *)calloc((size_t)n + 1, sizeof(double));
    db->nFiles = n;
    rewind(fp);
    if (!fgets(line, sizeof line, fp)) { fclose(fp); return -1; }
    for (int32_t i = 0; i < n && fgets(line, sizeof line, fp); i++) {
        char *save = NULL;
        char *col = strtok_r(line, "\t", &save);       /* index: position is what counts */
        col = strtok_r(NULL, "\t", &save);
        db->fileName[i] = strdup(col ? col : "");
        col = strtok_r(NULL, "\t", &save);
        db->fileNr[i] = col ? (int32_t)atol(col) : 0;
        col = strtok_r(NULL, "\t", &save);
        db->fileMd[i] = col ? (double)atol(col) : 0.0;
    }
    fclose(fp);
    return 0;
}

void igdc_close(igdc_db *db)
{
    if (!db) return;
    if (db->grp) { igd_hip_group_destroy(db->grp); db->grp = NULL; }
    for (int k = 1; k < db->ndev; k++) if (db->devs[k]) igd_hip_close(db->devs[k]);   /* devs[0] == dev */
    if (db->dev) igd_hip_close(db->dev);
    if (db->cName)
        for (int32_t c = 0; c < db->nCtg; c++) free(db->cName[c]);
    if (db->fileName)
        for (int32_t i = 0; i < db->nFiles; i++) free(db->fileName[i]);
    free(db->cName); free(db->fileName); free(db->fileNr); free(db->fileMd);
    free(db->nTile); free(db->nCntFlat); free(db->nCnt); free(db->tBase);
    free(db->dict);
    free(db);
}

/* --------------------------------------------------------------------------------------- */
static int attach_records(igdc_db *db, const void *records, int fd, int device)
{
    igd_hip_desc d;
    memset(&d, 0, sizeof d);
    d.nbp = db->nbp; d.gType = db->gType; d.nCtg = db->nCtg; d.nFiles = db->nFiles;
    d.nTile = db->nTile; d.nCnt = db->nCntFlat; d.records = records; d.nRecords = db->nRecords;
    d.fd = fd; d.fd_offset = db->dataOff;
    if (db->grp) { igd_hip_group_destroy(db->grp); db->grp = NULL; }
    for (int k = 1; k < db->ndev; k++) if (db->devs[k]) igd_hip_close(db->devs[k]);
    db->ndev = 0;
    if (db->dev) { igd_hip_close(db->dev); db->dev = NULL; }
    const int rc = igd_hip_open(&d, device, &db->dev);
    if (rc == IGD_HIP_OK) { db->devs[0] = db->dev; db->ndev = 1; }
    return rc;
}

/* the engine preads the tile region itself, staged through pinned buffers (no mmap faults) */
int igdc_attach_path(igdc_db *db, const char *igd_path, int device)
{
    const int64_t recBytes = db->gType == 0 ? 12 : 16;
    const int64_t need = db->dataOff + recBytes * db->nRecords;
    int fd = open(igd_path, O_RDONLY);
    if (fd < 0) return IGD_HIP_ERR_ARG;
    struct stat st;
    if (fstat(fd, &st) != 0 || (int64_t)st.st_size < need) { close(fd); return IGD_HIP_ERR_ARG; }
    (void)posix_fadvise(fd, 0, 0, POSIX_FADV_SEQUENTIAL);
    int rc = attach_records(db, NULL, fd, device);
    close(fd);
    return rc;
}

int igdc_attach_fp(igdc_db *db, FILE *fp, int device)
{
    const size_t recBytes = db->gType == 0 ? 12 : 16;
    size_t bytes = recBytes * (size_t)db->nRecords;
    void *buf = malloc(bytes ? bytes : 1);
    if (!buf) return IGD_HIP_ERR_NOMEM;
    long keep = ftell(fp);
    int ok = fseeko(fp, (off_t)db->dataOff, SEEK_SET) == 0 && read_exact(fp, buf, bytes) == 0;
    if (keep >= 0) fseek(fp, keep, SEEK_SET);
    int rc = ok ? attach_records(db, buf, -1, device) : IGD_HIP_ERR_ARG;
    free(buf);
    return rc;
}

/* ---- one interval on the host (igd_core.h: igdc_walk_one) -------------------------------- */
int64_t igdc_walk_one(const igdc_db *db, int fd, int32_t ichr, int32_t qs, int32_t qe, int32_t v, int use_v, int rule,
                      int64_t *hits, igdc_emit_fn emit, void *ctx)
{
    if (!db || fd < 0 || ichr < 0 || ichr >= db->nCtg) return 0;           /* :456-457 */
    const int32_t nbp = db->nbp, mT = db->nTile[ichr] - 1;
    const int32_t n1 = qs / nbp;                                           /* C division, as :459 */
    int32_t n2 = (int32_t)((uint32_t)qe - 1u) / nbp;                       /* (qe-1)/nbp with the reference's wrap */
    if (n1 < 0 || n1 > mT) return 0;                                       /* n1 < 0: out of bounds in the reference */
    if (n2 > mT) n2 = mT;
    if (rule == IGD_HIP_RULE_NEST && db->nCnt[ichr][n1] == 0) return 0;    /* :468 */
    const size_t recBytes = db->gType == 0 ? 12 : 16;
    if (db->gType == 0) use_v = 0;
    int64_t total = 0;
    int32_t *buf = NULL;
    size_t cap = 0;
    for (int32_t j = n1; j <= (n2 > n1 ? n2 : n1); j++) {
        const int32_t cnt = db->nCnt[ichr][j];
        if (cnt <= 0) continue;
        const int64_t off = igdc_tile_off(db, ichr, j);
        const size_t bytes = (size_t)cnt * recBytes;
        if (bytes > cap) { free(buf); buf = (int32_t *)malloc(bytes); cap = bytes; if (!buf) return -1; }
        size_t done = 0;
        while (done < bytes) {
            ssize_t got = pread(fd, (char *)buf + done, bytes - done, (off_t)(off + (int64_t)done));
            if (got <= 0) { free(buf); return -1; }
            done += (size_t)got;
        }
        /* later tiles skip the records that start before the tile: they were met in an earlier tile (:510-511) */
        const int64_t lob = j == n1 ? INT64_MIN : (int64_t)(int32_t)((uint32_t)nbp * (uint32_t)j);
        const size_t w = recBytes / 4;
        for (int32_t i = cnt - 1; i >= 0; i--) {                           /* the reverse scans of :489-493, :522-526 */
            const int32_t *r = buf + (size_t)i * w;                        /* idx, start, end[, value] (src/igd_base.h:41-52) */
            if (r[1] < qe && (int64_t)r[1] >= lob && r[2] > qs && (!use_v || r[3] >= v)) {
                if (r[0] < 0 || r[0] >= db->nFiles) continue;              /* the reference indexes hits[] unchecked (:491) */
                if (hits) hits[r[0]]++;
                if (emit) emit(ctx, r[0], r[1], r[2], i, j);
                total++;
            }
        }
    }
    free(buf);
    return total;
}

/* ---- multi-GPU in one process ----------------------------------------------------------- */
void igd_hip_set_error_(const char *msg);      /* libigd_hip.so: sets the calling thread's igd_hip_last_error() text */
int igdc_devices_from_env(int *devices, int max)
{
    const char *e = getenv("IGD_DEVICES");
    int n = 0;
    while (e && *e && n < max) {
        char *end = NULL;
        long d = strtol(e, &end, 10);
        if (end == e) break;
        devices[n++] = (int)d;
        e = end;
        while (*e == ',' || *e == ' ') e++;
    }
    return n;
}

typedef struct { igdc_db *db; const char *path; int device; igd_hip_db *out; int rc; char err[256]; } attach_job;
static void *attach_run(void *arg)
{
    attach_job *J = (attach_job *)arg;
    igdc_db tmp = *J->db;                       /* header tables are shared read-only; only .dev differs */
    tmp.dev = NULL;
    J->rc = igdc_attach_path(&tmp, J->path, J->device);
    J->out = tmp.dev;
    if (J->rc != IGD_HIP_OK) snprintf(J->err, sizeof J->err, "%s", igd_hip_last_error());   /* thread-local text */
    return NULL;
}

int igdc_attach_path_multi(igdc_db *db, const char *igd_path, const int *devices, int n)
{
    if (n < 1 || n > IGDC_MAX_DEVICES) return IGD_HIP_ERR_ARG;
    if (db->grp) { igd_hip_group_destroy(db->grp); db->grp = NULL; }
    for (int k = 0; k < db->ndev; k++) if (db->devs[k]) igd_hip_close(db->devs[k]);
    db->ndev = 0; db->dev = NULL;
    attach_job job[IGDC_MAX_DEVICES];
    pthread_t th[IGDC_MAX_DEVICES];
    int started[IGDC_MAX_DEVICES];
    for (int k = 0; k < n; k++) {
        job[k].db = db; job[k].path = igd_path; job[k].device = devices[k]; job[k].out = NULL; job[k].rc = IGD_HIP_ERR_DEVICE; job[k].err[0] = 0;
        started[k] = k > 0 && pthread_create(&th[k], NULL, attach_run, &job[k]) == 0;
        if (k > 0 && !started[k]) attach_run(&job[k]);
    }
    attach_run(&job[0]);
    int rc = IGD_HIP_OK;
    for (int k = 0; k < n; k++) {
        if (started[k]) pthread_join(th[k], NULL);
        if (job[k].rc != IGD_HIP_OK && rc == IGD_HIP_OK) { rc = job[k].rc; igd_hip_set_error_(job[k].err); }
    }
    if (rc != IGD_HIP_OK) {
        for (int k = 0; k < n; k++) if (job[k].out) igd_hip_close(job[k].out);
        return rc;
    }
    for (int k = 0; k < n; k++) db->devs[k] = job[k].out;
    db->ndev = n;
    db->dev = db->devs[0];
    /* the devices as one group: the communicators of the path's one exchange are built once, here */
    rc = igd_hip_group_create(db->devs, n, &db->grp);
    if (rc != IGD_HIP_OK) {
        for (int k = 0; k < n; k++) { igd_hip_close(db->devs[k]); db->devs[k] = NULL; }
        db->ndev = 0; db->dev = NULL;
        return rc;
    }
    const char *tm = getenv("IGD_TIMING");
    if (tm && *tm && *tm != '0')
        fprintf(stderr, "[igd timing] %d devices, hits[] summed by: %s%s%s\n", n, igd_hip_group_reduce_kind(db->grp),
                igd_hip_group_reduce_note(db->grp)[0] ? " -- " : "", igd_hip_group_reduce_note(db->grp));
    return IGD_HIP_OK;
}

int igdc_search_multi(igdc_db *db, const int32_t *ichr, const int32_t *qs, const int32_t *qe, int64_t nq,
                      int32_t v, int rule, int flags, int64_t *hits, int64_t *total)
{
    if (db->ndev < 1) return IGD_HIP_ERR_ARG;
    if (!db->grp || nq < db->ndev) return igd_hip_search_ex(db->devs[0], ichr, qs, qe, nq, v, rule, flags, hits, total);
    /* contiguous slabs, one per device, and the one exchange of the path: the engine group (RCCL all-reduce of hits[]) */
    return igd_hip_group_search(db->grp, ichr, qs, qe, nq, v, rule, flags, hits, total);
}

/* --------------------------------------------------------------------------------------- */
char *igdc_parse_bed(char *line, int32_t *st, int32_t *en, int require_chr)
{
    char *field[3];
    int nf = 0;
    char *p = line;
    field[nf++] = p;
    for (; *p; ++p) {
        if (*p == '\t') {
            *p = '\0';
            if (nf < 3) field[nf] = p + 1;
            nf++;                         /* fields past the third are cut off, not kept */
        }
    }
    int32_t s = -1, e = -1;
    if (nf >= 2) s = (int32_t)atol(field[1]);
    if (nf >= 3) e = (int32_t)atol(field[2]);
    *st = s; *en = e;
    if (nf < 3) return NULL;
    if (!require_chr) return field[0];
    const char *c = field[0];
    if (c[0] == 'c' && c[1] == 'h' && c[2] == 'r' && strlen(c) < 40 && e > 0) return field[0];
    return NULL;
}

struct igdc_lines {
    gzFile f;
    char *buf;
    size_t cap, beg, end;
    int eof;
};

igdc_lines *igdc_lines_open(const char *path)
{
    gzFile f = gzopen(path, "r");
    if (!f) return NULL;
    gzbuffer(f, 1 << 20);
    igdc_lines *r = (igdc_lines *)calloc(1, sizeof *r);
    r->f = f;
    r->cap = 1 << 20;
    r->buf = (char *)malloc(r->cap + 1);
    return r;
}

void igdc_lines_close(igdc_lines *r)
{
    if (!r) return;
    gzclose(r->f);
    free(r->buf);
    free(r);
}

/* Next line without its '\n'; one trailing '\r' is dropped when the line is longer than one
 * character (src/kseq.h:127).  A final line without '\n' is returned too. */
char *igdc_lines_next(igdc_lines *r, int64_t *len)
{
    for (;;) {
        char *nl = (r->end > r->beg) ? (char *)memchr(r->buf + r->beg, '\n', r->end - r->beg) : NULL;
        if (nl || (r->eof && r->end > r->beg)) {
            char *s = r->buf + r->beg;
            size_t L = nl ? (size_t)(nl - s) : r->end - r->beg;
            r->beg += L + (nl ? 1 : 0);
            if (!nl) r->beg = r->end;
            if (L > 1 && s[L - 1] == '\r') L--;
            s[L] = '\0';
            if (len) *len = (int64_t)L;
            return s;
        }
        if (r->eof) return NULL;
        /* refill: keep the partial line at the front */
        size_t have = r->end - r->beg;
        if (r->beg > 0) {
            memmove(r->buf, r->buf + r->beg, have);
            r->beg = 0; r->end = have;
        }
        if (r->end == r->cap) {
            r->cap *= 2;
            r->buf = (char *)realloc(r->buf, r->cap + 1);
        }
        int got = gzread(r->f, r->buf + r->end, (unsigned)(r->cap - r->end));
        if (got <= 0) r->eof = 1; else r->end += (size_t)got;
    }
}

int igdc_queries_push(igdc_queries *q, int32_t ichr, int32_t qs, int32_t qe)
{
    if (q->n == q->cap) {
        int64_t cap = q->cap ? q->cap * 2 : 4096;
        int32_t *a = (int32_t *)realloc(q->ichr, sizeof(int32_t) * (size_t)cap);
        int32_t *b = (int32_t *)realloc(q->qs, sizeof(int32_t) * (size_t)cap);
        int32_t *c = (int32_t *)realloc(q->qe, sizeof(int32_t) * (size_t)cap);
        if (a) q->ichr = a;
        if (b) q->qs = b;
        if (c) q->qe = c;
        if (!a || !b || !c) return -1;
        q->cap = cap;
    }
    if (q->n > 0 && (ichr < q->ichr[q->n - 1] || (ichr == q->ichr[q->n - 1] && qs < q->qs[q->n - 1]))) q->unsorted = 1;
    q->ichr[q->n] = ichr; q->qs[q->n] = qs; q->qe[q->n] = qe;
    q->n++;
    {
        const int64_t len = (int64_t)qe - (int64_t)qs;
        if (len > q->max_len) q->max_len = len > INT32_MAX ? INT32_MAX : (int32_t)len;
    }
    return 0;
}

int igdc_queries_flags(const igdc_queries *q, int32_t nbp)
{
    if (!q) return 0;
    if (q->unsorted) return IGD_HIP_FLAG_BUCKET;         /* the parser has seen it: the device need not check again */
    return IGD_HIP_FLAG_SORTED | (q->max_len < nbp ? IGD_HIP_FLAG_SHORT : 0);
}

/* A position-sorted BED whose chromosomes come in another order than the database numbers its contigs (`sort -k1,1
 * -k2,2n` puts chr10 before chr2; the database numbers contigs by first appearance in its input files): every contig is
 * ONE run of lines ordered by start, only the runs are out of order.  Counts are sums over queries, so the runs may be
 * put into contig order -- one pass to check, one to copy -- and the batch takes the merge join instead of the bucket
 * path (whose grouping kernels, built for scattered queries, are at their worst on long sorted runs: 334 vs 68 us per
 * 10^6 queries).  Returns 1 when the queries were reordered (q->unsorted cleared), 0 when they are left alone.
 * Not for `-f` / Seqpare, whose output follows the query order. */
int igdc_queries_group_contigs(igdc_queries *q, int32_t nCtg)
{
    if (!q || !q->unsorted || q->n < 2 || nCtg <= 0) return 0;
    int64_t *cnt = (int64_t *)calloc((size_t)nCtg + 1, sizeof(int64_t));
    if (!cnt) return 0;
    int ok = 1;
    for (int64_t i = 0; i < q->n && ok; i++) {
        const int32_t c = q->ichr[i];
        if (c < 0 || c >= nCtg) { ok = 0; break; }
        if (i > 0 && c == q->ichr[i - 1]) { if (q->qs[i] < q->qs[i - 1]) ok = 0; }
        else if (cnt[c] != 0) ok = 0;                     /* the contig's second run */
        cnt[c]++;
    }
    int32_t *a = NULL, *b = NULL, *d = NULL;
    if (ok) {
        a = (int32_t *)malloc(sizeof(int32_t) * (size_t)q->n);
        b = (int32_t *)malloc(sizeof(int32_t) * (size_t)q->n);
        d = (int32_t *)malloc(sizeof(int32_t) * (size_t)q->n);
        if (!a || !b || !d) ok = 0;
    }
    if (ok) {
        int64_t at = 0;
        for (int32_t c = 0; c < nCtg; c++) { const int64_t k = cnt[c]; cnt[c] = at; at += k; }   /* first slot of each contig */
        for (int64_t i = 0; i < q->n;) {                  /* run by run */
            const int32_t c = q->ichr[i];
            int64_t j = i;
            while (j < q->n && q->ichr[j] == c) j++;
            memcpy(a + cnt[c], q->ichr + i, sizeof(int32_t) * (size_t)(j - i));
            memcpy(b + cnt[c], q->qs + i, sizeof(int32_t) * (size_t)(j - i));
            memcpy(d + cnt[c], q->qe + i, sizeof(int32_t) * (size_t)(j - i));
            i = j;
        }
        free(q->ichr); free(q->qs); free(q->qe);
        q->ichr = a; q->qs = b; q->qe = d; q->cap = q->n; q->unsorted = 0;
    } else { free(a); free(b); free(d); }
    free(cnt);
    return ok;
}

void igdc_queries_free(igdc_queries *q)
{
    free(q->ichr); free(q->qs); free(q->qe);
    memset(q, 0, sizeof *q);
}

/* ---------------------------------------------------------------------------------------
 * Query ingest.  Plain-text BED files are mapped and parsed by several
 * threads, each on a newline-aligned slice, without modifying the text; the slices' results
 * are concatenated in file order, so the accepted (contig,start,end) list is exactly what the
 * sequential reader (and the reference's ks_getuntil + parse_bed loop) produces.  gzip input
 * keeps the sequential path (inflate is serial). */
#include <pthread.h>
#include <ctype.h>

/* atol() of the field [p,end): optional leading white space, optional sign, digits; saturates
 * like strtol; then narrowed to int32 the way `int32_t st = atol(..)` narrows. */
static int32_t field_atol32(const char *p, const char *end)
{
    while (p < end && isspace((unsigned char)*p)) p++;
    int neg = 0;
    if (p < end && (*p == '+' || *p == '-')) { neg = (*p == '-'); p++; }
    unsigned long long v = 0;
    int sat = 0;
    for (; p < end && *p >= '0' && *p <= '9'; p++) {
        if (v > (0x7fffffffffffffffULL - (unsigned)(*p - '0')) / 10) sat = 1;
        if (!sat) v = v * 10 + (unsigned)(*p - '0');
    }
    long long r;
    if (sat) r = neg ? (long long)(-0x7fffffffffffffffLL - 1) : 0x7fffffffffffffffLL;
    else r = neg ? -(long long)v : (long long)v;
    return (int32_t)r;
}

static int32_t get_id_n(const igdc_db *db, const char *s, size_t len)
{
    uint32_t h = 2166136261u;
    for (size_t i = 0; i < len; i++) h = (h ^ (unsigned char)s[i]) * 16777619u;
    uint32_t p = h & (uint32_t)(db->dictCap - 1);
    for (;;) {
        int32_t c = db->dict[p];
        if (c < 0) return -1;
        if (strlen(db->cName[c]) == len && memcmp(db->cName[c], s, len) == 0) return c;
        p = (p + 1) & (uint32_t)(db->dictCap - 1);
    }
}

/* one line [s,e) (no '\n'); same accept rule as igdc_parse_bed + dictionary lookup */
static void ingest_line(const igdc_db *db, const char *s, const char *e, int require_chr, igdc_queries *q)
{
    if (e - s > 1 && e[-1] == '\r') e--;                    /* src/kseq.h:127 */
    const char *nul = (const char *)memchr(s, '\0', (size_t)(e - s));
    if (nul) e = nul;                                        /* C-string semantics of the reference */
    const char *f1 = (const char *)memchr(s, '\t', (size_t)(e - s));
    if (!f1) return;
    const char *f2 = (const char *)memchr(f1 + 1, '\t', (size_t)(e - f1 - 1));
    if (!f2) return;
    const char *f3 = (const char *)memchr(f2 + 1, '\t', (size_t)(e - f2 - 1));
    if (!f3) f3 = e;
    const int32_t st = field_atol32(f1 + 1, f2), en = field_atol32(f2 + 1, f3);
    const size_t nl = (size_t)(f1 - s);
    if (require_chr) {
        if (!(nl >= 3 && s[0] == 'c' && s[1] == 'h' && s[2] == 'r' && nl < 40 && en > 0)) {
            /* a name shorter than 3 chars: the reference reads ctg[1], ctg[2] past a NUL only when
             * the earlier ones matched; "c", "ch" fail the prefix test like here */
            return;
        }
    }
    const int32_t id = get_id_n(db, s, nl);
    if (id >= 0) igdc_queries_push(q, id, st, en);
}

typedef struct {
    const igdc_db *db;
    const char *beg, *end;
    int require_chr;
    igdc_queries q;
} ingest_job;

static void *ingest_run(void *arg)
{
    ingest_job *j = (ingest_job *)arg;
    const char *p = j->beg;
    while (p < j->end) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(j->end - p));
        const char *le = nl ? nl : j->end;
        ingest_line(j->db, p, le, j->require_chr, &j->q);
        p = le + 1;
    }
    return NULL;
}

static int read_queries_text_parallel(const igdc_db *db, const char *map, size_t size, int require_chr,
                                      igdc_queries *out)
{
    int nthr = (int)sysconf(_SC_NPROCESSORS_ONLN);
    const char *env = getenv("IGD_PARSE_THREADS");
    if (env && *env) nthr = atoi(env);
    if (nthr < 1) nthr = 1;
    if (nthr > 16) nthr = 16;
    if (size < ((size_t)1 << 20)) nthr = 1;
    ingest_job job[16];
    pthread_t th[16];
    const char *end = map + size;
    const char *cut = map;
    int n = 0;
    for (int t = 0; t < nthr && cut < end; t++) {
        const char *b = cut;
        const char *e = end;
        if (t + 1 < nthr) {
            const char *guess = map + (size / (size_t)nthr) * (size_t)(t + 1);
            if (guess < b) guess = b;
            const char *nl = guess < end ? (const char *)memchr(guess, '\n', (size_t)(end - guess)) : NULL;
            e = nl ? nl + 1 : end;
        }
        memset(&job[n], 0, sizeof job[n]);
        job[n].db = db; job[n].beg = b; job[n].end = e; job[n].require_chr = require_chr;
        cut = e;
        n++;
    }
    for (int t = 1; t < n; t++)
        if (pthread_create(&th[t], NULL, ingest_run, &job[t]) != 0) { th[t] = 0; ingest_run(&job[t]); }
    if (n > 0) ingest_run(&job[0]);
    for (int t = 1; t < n; t++)
        if (th[t]) pthread_join(th[t], NULL);
    int64_t total = 0;
    for (int t = 0; t < n; t++) total += job[t].q.n;
    memset(out, 0, sizeof *out);
    if (n == 1) { *out = job[0].q; return 0; }
    out->cap = total > 0 ? total : 1;
    out->ichr = (int32_t *)malloc(sizeof(int32_t) * (size_t)out->cap);
    out->qs = (int32_t *)malloc(sizeof(int32_t) * (size_t)out->cap);
    out->qe = (int32_t *)malloc(sizeof(int32_t) * (size_t)out->cap);
    if (!out->ichr || !out->qs || !out->qe) return -1;
    for (int t = 0; t < n; t++) {
        igdc_queries *q = &job[t].q;
        if (q->n) {
            /* order across the seam, then inside the slice */
            if (out->n > 0 && (q->ichr[0] < out->ichr[out->n - 1] ||
                               (q->ichr[0] == out->ichr[out->n - 1] && q->qs[0] < out->qs[out->n - 1])))
                out->unsorted = 1;
            memcpy(out->ichr + out->n, q->ichr, sizeof(int32_t) * (size_t)q->n);
            memcpy(out->qs + out->n, q->qs, sizeof(int32_t) * (size_t)q->n);
            memcpy(out->qe + out->n, q->qe, sizeof(int32_t) * (size_t)q->n);
            out->n += q->n;
        }
        out->unsorted |= q->unsorted;
        if (q->max_len > out->max_len) out->max_len = q->max_len;
        igdc_queries_free(q);
    }
    return 0;
}

int igdc_read_queries(const igdc_db *db, const char *qfile, int require_chr, igdc_queries *out)
{
    memset(out, 0, sizeof *out);
    /* plain text -> parallel path; gzip (magic 1f 8b) or anything unmappable -> sequential */
    int fd = open(qfile, O_RDONLY);
    if (fd >= 0) {
        struct stat st;
        unsigned char magic[2] = {0, 0};
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0 && pread(fd, magic, 2, 0) == 2 &&
            !(magic[0] == 0x1f && magic[1] == 0x8b) && !getenv("IGD_PARSE_SEQUENTIAL")) {
            void *map = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (map != MAP_FAILED) {
                (void)madvise(map, (size_t)st.st_size, MADV_SEQUENTIAL);
                int rc = read_queries_text_parallel(db, (const char *)map, (size_t)st.st_size, require_chr, out);
                munmap(map, (size_t)st.st_size);
                close(fd);
                return rc;
            }
        }
        close(fd);
    }
    igdc_lines *r = igdc_lines_open(qfile);
    if (!r) return -1;
    char *line;
    while ((line = igdc_lines_next(r, NULL)) != NULL) {
        int32_t st, en;
        char *chrm = igdc_parse_bed(line, &st, &en, require_chr);
        if (!chrm) continue;
        int32_t id = igdc_get_id(db, chrm);
        if (id < 0) continue;
        if (igdc_queries_push(out, id, st, en) != 0) { igdc_lines_close(r); return -1; }
    }
    igdc_lines_close(r);
    return 0;
}
