/* igd_py_abi.c -- handle-based flavour (include/igd_py_abi.h) for the Cython wrapper of
 * /root/reference/src_py.  Thin shim over igd_core + the HIP engine; no CPU search. */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sysexits.h>

#include "igd_py_abi.h"
#include "igd_core.h"
#include "igd_create_host.h"

struct iGD_t {
    igdc_db *core;          /* NULL until open_iGD */
    char *path;             /* the engine is attached from it by the first batch that needs the GPU */
};

static int device_from_env(void)
{
    const char *e = getenv("IGD_DEVICE");
    return e && *e ? atoi(e) : 0;
}

/* A library must not end the interpreter that loaded it: an engine failure is reported on stderr,
 * the call returns like the reference's silent failures (hits untouched) and igd_engine_status() keeps
 * the code -- the Python shim (igd_amd/igd_py.py) turns it into an exception. */
static int g_fail_rc = 0;
int igd_engine_status(void) { return g_fail_rc; }
void igd_engine_clear(void) { g_fail_rc = 0; }
static void engine_failed(const char *where, int rc)
{
    fprintf(stderr, "igd_py: %s: GPU engine unavailable (code %d): %s\n"
                    "igd_py: this build has no CPU search path.\n", where, rc, igd_hip_last_error());
    g_fail_rc = rc ? rc : IGD_HIP_ERR_DEVICE;
}

iGD_t *iGD_init(void)
{
    return (iGD_t *)calloc(1, sizeof(iGD_t));
}

int32_t get_nFiles(iGD_t *iGD)
{
    return iGD && iGD->core ? iGD->core->nFiles : 0;
}

void open_iGD(iGD_t *iGD, char *igdFile)
{
    if (!iGD) return;
    if (iGD->core) { igdc_close(iGD->core); iGD->core = NULL; }
    free(iGD->path); iGD->path = NULL;
    igdc_db *core = igdc_open(igdFile);
    if (!core) {
        printf("Can't open file %s", igdFile);
        return;
    }
    char *tsv = igdc_index_path(igdFile);
    if (igdc_load_index(core, tsv) != 0) printf("file not found:%s\n", tsv);
    free(tsv);
    /* header and index only, like the reference's open_iGD (src_py/igd_base.c); the tile region goes to the GPU when the
     * first batch that is not small arrives (igdc_search_auto) */
    iGD->core = core;
    iGD->path = strdup(igdFile);
}

void close_iGD(iGD_t *iGD)
{
    if (!iGD) return;
    if (iGD->core) igdc_close(iGD->core);
    free(iGD->path);
    free(iGD);
}

void create_iGD(iGD_t *iGD, char *iPath, char *oPath, char *igdName, int tile_size)
{
    /* the reference appends "/", "*" or "/" "*" to the CALLER's buffers (src_py/igd_create.c:22-31) --
     * which its own Cython wrapper hands over as immutable Python bytes objects; copies are extended here */
    const size_t lo = strlen(oPath), li = strlen(iPath);
    char *oBuf = (char *)malloc(lo + 4), *iBuf = (char *)malloc(li + 4);
    if (!oBuf || !iBuf) { free(oBuf); free(iBuf); return; }
    memcpy(oBuf, oPath, lo + 1);
    memcpy(iBuf, iPath, li + 1);
    oPath = oBuf; iPath = iBuf;
    if (lo && oPath[lo - 1] != '/') strcat(oPath, "/");
    if (li && iPath[li - 1] == '/') strcat(iPath, "*");
    else if (li && iPath[li - 1] != '*') strcat(iPath, "/*");
    size_t L = strlen(oPath) + strlen(igdName) + 8;
    char *probe = (char *)malloc(L);
    snprintf(probe, L, "%s%s.igd", oPath, igdName);
    struct stat st;
    if (stat(probe, &st) == 0) {
        printf("The igd database file %s exists!\n", probe);
        free(probe); free(oBuf); free(iBuf);
        return;
    }
    igdc_create_opts o;
    o.ipath = iPath; o.opath = oPath; o.name = igdName;
    o.nbp = tile_size > 0 ? tile_size : 16384;
    o.mode = IGDC_CREATE_GLOB; o.msg = IGDC_MSG_PY; o.linebuf = 256;       /* src_py/igd_create.c:70 */
    const char *dv = getenv("IGD_DEVICE");
    o.device = dv ? atoi(dv) : 0;
    const int rc = igdc_create(&o);
    if (rc < 0) engine_failed("create_iGD", rc);
    if (rc == 0 && iGD) open_iGD(iGD, probe);                              /* src_py/igd_create.c:140-141 */
    free(probe); free(oBuf); free(iBuf);
}

void get_overlaps(iGD_t *iGD, char *chrm, int32_t qs, int32_t qe, int64_t *hits)
{
    if (!iGD || !iGD->core) return;
    int32_t ichr = igdc_get_id(iGD->core, chrm);
    if (ichr < 0) return;
    int rc = igdc_search_auto(iGD->core, iGD->path, device_from_env(), &ichr, &qs, &qe, 1, IGD_HIP_NO_VALUE_FILTER,
                              IGD_HIP_RULE_NEST, 0, hits, NULL);
    if (rc != IGD_HIP_OK) engine_failed("get_overlaps", rc);
}

int64_t getOverlaps(iGD_t *iGD, char *qFile, int64_t *hits)
{
    if (!iGD || !iGD->core) return 0;
    igdc_queries q;
    if (igdc_read_queries(iGD->core, qFile, 0, &q) != 0) return 0;
    (void)igdc_queries_group_contigs(&q, iGD->core->nCtg);    /* a sorted BED with another chromosome order than the database's */
    if (q.n > 0) {
        int rc = igdc_search_auto(iGD->core, iGD->path, device_from_env(), q.ichr, q.qs, q.qe, q.n, IGD_HIP_NO_VALUE_FILTER,
                                  IGD_HIP_RULE_NEST, igdc_queries_flags(&q, iGD->core->nbp), hits, NULL);
        if (rc != IGD_HIP_OK) { engine_failed("getOverlaps", rc); igdc_queries_free(&q); return 0; }
    }
    igdc_queries_free(&q);
    int64_t nols = 0;
    for (int32_t i = 0; i < iGD->core->nFiles; i++) nols += hits[i];
    return nols;
}
