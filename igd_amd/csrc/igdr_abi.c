/* igdr_abi.c -- R flavour (include/igdr_abi.h) over igd_core + the HIP engine. */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <sysexits.h>
#include <unistd.h>

#include "igdr_abi.h"
#include "igd_core.h"
#include "igd_create_host.h"

struct iGD_t {
    igdc_db *core;
    char *path;
};

static int device_from_env(void)
{
    const char *e = getenv("IGD_DEVICE");
    return e && *e ? atoi(e) : 0;
}

/* A library must not end the R session that loaded it: an engine failure is reported on stderr, the
 * call returns like the reference's silent failures (hits untouched, NULL handle) and
 * igd_engine_status() keeps the code; the .Call entry points raise it as an R error (Rf_error). */
static int g_fail_rc = 0;
int igd_engine_status(void) { return g_fail_rc; }
void igd_engine_clear(void) { g_fail_rc = 0; }
static void engine_failed(const char *where, int rc)
{
    fprintf(stderr, "IGDr: %s: GPU engine unavailable (code %d): %s\n"
                    "IGDr: this build has no CPU search path.\n", where, rc, igd_hip_last_error());
    g_fail_rc = rc ? rc : IGD_HIP_ERR_DEVICE;
}

iGD_t *open_iGD(char *igdFile)
{
    igdc_db *core = igdc_open(igdFile);
    if (!core) {
        printf("Can't open file %s", igdFile);
        return NULL;
    }
    char *tsv = igdc_index_path(igdFile);
    if (igdc_load_index(core, tsv) != 0) printf("file not found:%s\n", tsv);
    free(tsv);
    /* header and index only, like the reference's open_iGD (IGDr/src/igd_base.c); the tile region goes to the GPU when the
     * first batch that is not small arrives (igdc_search_auto) */
    iGD_t *h = (iGD_t *)calloc(1, sizeof *h);
    h->core = core;
    h->path = strdup(igdFile);
    return h;
}

void close_iGD(iGD_t *iGD)
{
    if (!iGD) return;
    igdc_close(iGD->core);
    free(iGD->path);
    free(iGD);
}

int32_t get_id(iGD_t *iGD, const char *chrm)
{
    return iGD ? igdc_get_id(iGD->core, chrm) : -1;
}

void get_overlaps(iGD_t *iGD, char *chrm, int32_t qs, int32_t qe, int64_t *hits)
{
    if (!iGD) return;
    int32_t ichr = igdc_get_id(iGD->core, chrm);
    if (ichr < 0) return;
    int rc = igdc_search_auto(iGD->core, iGD->path, device_from_env(), &ichr, &qs, &qe, 1, IGD_HIP_NO_VALUE_FILTER,
                              IGD_HIP_RULE_NEST, 0, hits, NULL);
    if (rc != IGD_HIP_OK) engine_failed("get_overlaps", rc);
}

void igdr_search_n32(iGD_t *iGD, int32_t n, const char *const *chrm, const int32_t *qs,
                     const int32_t *qe, int32_t *hits)
{
    if (!iGD || n <= 0) return;
    igdc_queries q;
    memset(&q, 0, sizeof q);
    for (int32_t i = 0; i < n; i++) {
        int32_t id = igdc_get_id(iGD->core, chrm[i]);
        if (id >= 0) igdc_queries_push(&q, id, qs[i], qe[i]);
    }
    const int32_t nf = iGD->core->nFiles;
    int64_t *h64 = (int64_t *)calloc((size_t)nf + 1, sizeof(int64_t));
    if (q.n > 0) {
        int rc = igdc_search_auto(iGD->core, iGD->path, device_from_env(), q.ichr, q.qs, q.qe, q.n, IGD_HIP_NO_VALUE_FILTER,
                                  IGD_HIP_RULE_NEST, 0, h64, NULL);
        if (rc != IGD_HIP_OK) { engine_failed("search", rc); memset(h64, 0, sizeof(int64_t) * (size_t)nf); }
    }
    for (int32_t f = 0; f < nf; f++) hits[f] = (int32_t)((int64_t)hits[f] + h64[f]);
    free(h64);
    igdc_queries_free(&q);
}

void get_overlaps32(iGD_t *iGD, char *chrm, int32_t qs, int32_t qe, int32_t *hits)
{
    const char *name = chrm;
    igdr_search_n32(iGD, 1, &name, &qs, &qe, hits);
}

/* One interval on a database that is opened for just this call (IGDr/R/IGDr.R search_1 via .C): the host reads the
 * interval's own tiles (igdc_walk_one) -- the reference reads them too (IGDr/src/igd_search.c:25-103) -- instead of
 * uploading the whole database to the GPU for a single query. */
void search_1(char **igdFile, char **qchr, int32_t *qs, int32_t *qe, int64_t *hits)
{
    igdc_db *core = igdc_open(*igdFile);
    if (!core) {
        printf("Can't open file %s", *igdFile);
        return;
    }
    char *tsv = igdc_index_path(*igdFile);
    if (igdc_load_index(core, tsv) != 0) printf("file not found:%s\n", tsv);
    free(tsv);
    const int32_t id = igdc_get_id(core, *qchr);
    const int fd = id >= 0 ? open(*igdFile, O_RDONLY) : -1;
    if (fd >= 0) {
        (void)igdc_walk_one(core, fd, id, *qs, *qe, 0, 0, IGD_HIP_RULE_NEST, hits, NULL, NULL);
        close(fd);
    }
    igdc_close(core);
}

void getOverlaps(char **igdFile, char **qFile, int64_t *hits)
{
    iGD_t *h = open_iGD(*igdFile);
    if (!h) return;
    igdc_queries q;
    if (igdc_read_queries(h->core, *qFile, 0, &q) == 0) {
        (void)igdc_queries_group_contigs(&q, h->core->nCtg);  /* a sorted BED with another chromosome order than the database's */
        if (q.n > 0) {
            int rc = igdc_search_auto(h->core, h->path, device_from_env(), q.ichr, q.qs, q.qe, q.n, IGD_HIP_NO_VALUE_FILTER,
                                      IGD_HIP_RULE_NEST, igdc_queries_flags(&q, h->core->nbp), hits, NULL);
            if (rc != IGD_HIP_OK) engine_failed("getOverlaps", rc);
        }
        igdc_queries_free(&q);
    }
    close_iGD(h);
}

/* create_iGD / create_iGD_f, IGDr/src/igd_create.c:19-170, :173-319 (.C: every argument a pointer) */
static void create_common(char **iPath, char **oPath, char **igdName, int *binsize, int mode)
{
    const size_t li = strlen(*iPath), lo = strlen(*oPath);
    char *in = (char *)malloc(li + 4), *out = (char *)malloc(lo + 4);
    strcpy(in, *iPath);
    strcpy(out, *oPath);
    if (lo == 0 || out[lo - 1] != '/') strcat(out, "/");
    if (mode == IGDC_CREATE_GLOB && li > 0) {
        if (in[li - 1] == '/') strcat(in, "*");
        else if (in[li - 1] != '*') strcat(in, "/*");
    }
    const size_t L = strlen(out) + strlen(*igdName) + 8;
    char *probe = (char *)malloc(L);
    snprintf(probe, L, "%s%s.igd", out, *igdName);
    struct stat st;
    if (stat(probe, &st) == 0) printf("The igd database file %s exists!\n", probe);
    else {
        igdc_create_opts o;
        o.ipath = in; o.opath = out; o.name = *igdName;
        o.nbp = (binsize && *binsize > 0) ? *binsize : 16384;
        o.mode = mode; o.msg = IGDC_MSG_R; o.linebuf = 256;
        const char *dv = getenv("IGD_DEVICE");
        o.device = dv ? atoi(dv) : 0;
        const int rc = igdc_create(&o);
        if (rc < 0) engine_failed("create_iGD", rc);
    }
    free(probe); free(in); free(out);
}
void create_iGD(char **iPath, char **oPath, char **igdName, int *binsize) { create_common(iPath, oPath, igdName, binsize, IGDC_CREATE_GLOB); }
void create_iGD_f(char **iPath, char **oPath, char **igdName, int *binsize) { create_common(iPath, oPath, igdName, binsize, IGDC_CREATE_LIST); }

#ifdef IGDR_HAVE_R
/* ---- .Call entry points (need R's headers; not compiled in the build container) ------- */
#include <R.h>
#include <Rdefines.h>

static iGD_t *handle_of(SEXP igdr)
{
    iGD_t *h = (iGD_t *)R_ExternalPtrAddr(igdr);
    if (h == NULL) error("iGD_free: iGDr external pointer is NULL");
    return h;
}

SEXP iGD_free(SEXP igdr)
{
    iGD_t *h = (iGD_t *)R_ExternalPtrAddr(igdr);
    if (h == NULL) return R_NilValue;           /* finalizer after an explicit free */
    close_iGD(h);
    R_SetExternalPtrAddr(igdr, NULL);
    return R_NilValue;
}
static void igdr_finalizer(SEXP igdr) { (void)iGD_free(igdr); }

SEXP iGD_new(SEXP igd_file)
{
    iGD_t *h = open_iGD((char *)CHAR(STRING_ELT(igd_file, 0)));
    if (h == NULL && g_fail_rc) { g_fail_rc = 0; error("IGDr: GPU engine unavailable: %s", igd_hip_last_error()); }
    SEXP igdr, klass, obj;
    PROTECT(igdr = R_MakeExternalPtr(h, R_NilValue, R_NilValue));
    R_RegisterCFinalizer(igdr, igdr_finalizer);
    klass = PROTECT(MAKE_CLASS("IGDr"));
    PROTECT(obj = NEW_OBJECT(klass));
    SET_SLOT(obj, Rf_install("ref"), igdr);
    UNPROTECT(3);
    return obj;
}

SEXP search_1r(SEXP igdr, SEXP qchrm, SEXP qs, SEXP qe)
{
    iGD_t *h = handle_of(igdr);
    SEXP hits;
    PROTECT(hits = allocVector(INTSXP, h->core->nFiles));
    memset(INTEGER(hits), 0, (size_t)h->core->nFiles * sizeof(int));
    get_overlaps32(h, (char *)CHAR(STRING_ELT(qchrm, 0)), INTEGER(qs)[0], INTEGER(qe)[0], INTEGER(hits));
    UNPROTECT(1);
    if (g_fail_rc) { g_fail_rc = 0; error("IGDr: GPU engine failure: %s", igd_hip_last_error()); }
    return hits;
}

SEXP search_nr(SEXP igdr, SEXP n, SEXP qchrm, SEXP qs, SEXP qe)
{
    iGD_t *h = handle_of(igdr);
    const int32_t m = INTEGER(n)[0];
    SEXP hits;
    PROTECT(hits = allocVector(INTSXP, h->core->nFiles));
    memset(INTEGER(hits), 0, (size_t)h->core->nFiles * sizeof(int));
    const char **names = (const char **)malloc(sizeof(char *) * (size_t)(m > 0 ? m : 1));
    for (int32_t i = 0; i < m; i++) names[i] = CHAR(STRING_ELT(qchrm, i));
    igdr_search_n32(h, m, names, INTEGER(qs), INTEGER(qe), INTEGER(hits));
    free(names);
    UNPROTECT(1);
    if (g_fail_rc) { g_fail_rc = 0; error("IGDr: GPU engine failure: %s", igd_hip_last_error()); }
    return hits;
}

static SEXP scalar_int(int v)
{
    SEXP s;
    PROTECT(s = allocVector(INTSXP, 1));
    INTEGER(s)[0] = v;
    UNPROTECT(1);
    return s;
}
SEXP get_cid(SEXP igdr, SEXP chrom) { return scalar_int(get_id(handle_of(igdr), CHAR(STRING_ELT(chrom, 0)))); }
SEXP get_nbp(SEXP igdr) { return scalar_int(handle_of(igdr)->core->nbp); }
SEXP get_nfiles(SEXP igdr) { return scalar_int(handle_of(igdr)->core->nFiles); }
SEXP get_nCtgs(SEXP igdr) { return scalar_int(handle_of(igdr)->core->nCtg); }

SEXP get_binLen(SEXP igdr, SEXP ichr, SEXP bin)
{
    igdc_db *c = handle_of(igdr)->core;
    int i = INTEGER(ichr)[0] - 1, j = INTEGER(bin)[0] - 1;
    if (i >= c->nCtg || i < 0 || j < 0 || j >= c->nTile[i]) return R_NilValue;
    return scalar_int(c->nCnt[i][j]);
}

SEXP get_binData(SEXP igdr, SEXP ichr, SEXP bin)
{
    iGD_t *h = handle_of(igdr);
    igdc_db *c = h->core;
    int i = INTEGER(ichr)[0] - 1, j = INTEGER(bin)[0] - 1;
    if (i < 0 || i >= c->nCtg || j < 0 || j >= c->nTile[i]) return R_NilValue;
    int n = c->nCnt[i][j];
    if (n < 1) return R_NilValue;
    const size_t rb = c->gType == 0 ? 12 : 16;
    int32_t *raw = (int32_t *)malloc(rb * (size_t)n);
    FILE *fp = fopen(h->path, "rb");
    if (!fp || fseeko(fp, (off_t)igdc_tile_off(c, i, j), SEEK_SET) != 0 || fread(raw, rb, (size_t)n, fp) != (size_t)n) {
        if (fp) fclose(fp);
        free(raw);
        return R_NilValue;
    }
    fclose(fp);
    SEXP st = PROTECT(allocVector(INTSXP, n)), en = PROTECT(allocVector(INTSXP, n)), ix = PROTECT(allocVector(INTSXP, n));
    const int w = (int)(rb / 4);
    for (int k = 0; k < n; k++) {
        INTEGER(ix)[k] = raw[k * w]; INTEGER(st)[k] = raw[k * w + 1]; INTEGER(en)[k] = raw[k * w + 2];
    }
    free(raw);
    SEXP out = PROTECT(allocVector(VECSXP, 3));
    SET_VECTOR_ELT(out, 0, ix); SET_VECTOR_ELT(out, 1, st); SET_VECTOR_ELT(out, 2, en);
    UNPROTECT(4);
    return out;
}
#endif /* IGDR_HAVE_R */
