/* igd_hip_lazy.c -- the host flavours' door to the HIP engine: libigd_hip.so is mapped when the FIRST engine function is
 * called, not when the process starts.
 *
 * Why: libigd_hip.so needs libamdhip64 and a dozen ROCm libraries; mapping and initialising them costs ~13 ms of every
 * process start (measured: `igd` without arguments 13.0 ms, the reference 1.1 ms) -- more than the reference needs for a
 * whole `igd search -q` of 10^3 queries (6 ms).  The entry points that never touch the GPU -- `-r`, get_overlaps*, small
 * query files (igd_hostpath.c), the header loaders -- should not pay it.  So libigd.so / libigd_py.so / libigdr.so do NOT
 * link -ligd_hip: they carry these trampolines, one per engine function the host code calls (include/igd_hip.h), which
 * dlopen "libigd_hip.so" from the directory this library was loaded from and forward.
 *
 * Not a fallback of any kind: if the engine library cannot be loaded every trampoline fails loudly -- the int-returning
 * ones with IGD_HIP_ERR_DEVICE, igd_hip_last_error() with dlopen's reason -- and the callers report it like any other
 * engine failure ("no CPU search path").
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "igd_hip.h"

static void *g_lib = NULL;
static char g_why[600] = "";
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static void load_engine(void)
{
    Dl_info di;
    char path[4096];
    path[0] = 0;
    if (dladdr((void *)load_engine, &di) && di.dli_fname) {
        const char *slash = strrchr(di.dli_fname, '/');
        if (slash && (size_t)(slash - di.dli_fname) + 32 < sizeof path) {
            memcpy(path, di.dli_fname, (size_t)(slash - di.dli_fname) + 1);
            strcpy(path + (slash - di.dli_fname) + 1, "libigd_hip.so");
        }
    }
    if (path[0]) g_lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!g_lib) {
        const char *e1 = dlerror();
        snprintf(g_why, sizeof g_why, "the HIP engine library could not be loaded (%s)", e1 ? e1 : path);
        g_lib = dlopen("libigd_hip.so", RTLD_NOW | RTLD_LOCAL);      /* the loader's search path (LD_LIBRARY_PATH, rpath) */
    }
}

static void *engine_sym(const char *name)
{
    pthread_once(&g_once, load_engine);
    return g_lib ? dlsym(g_lib, name) : NULL;
}

/* the engine is mapped already?  (the command line tool leaves through _exit() only when it is: igd_main.c) */
int igd_hip_lazy_loaded(void) { return g_lib != NULL; }

#define RESOLVE(type, name)                               \
    static type fn = NULL;                                \
    if (!fn) fn = (type)engine_sym(name)

const char *igd_hip_last_error(void)
{
    typedef const char *(*fn_t)(void);
    RESOLVE(fn_t, "igd_hip_last_error");
    return fn ? fn() : g_why;
}

void igd_hip_set_error_(const char *msg)
{
    typedef void (*fn_t)(const char *);
    RESOLVE(fn_t, "igd_hip_set_error_");
    if (fn) fn(msg);
}

int igd_hip_device_count(void)
{
    typedef int (*fn_t)(void);
    RESOLVE(fn_t, "igd_hip_device_count");
    return fn ? fn() : 0;
}

/* Answered here, without mapping the engine: a small `search -q -f` asks for the step of its loop before it knows that the
 * host path takes the file (the question alone cost the 13 ms this file exists to avoid).  Same rule as the
 * engine's max_batch(): 2^24 queries, lowered by the TEST-ONLY variable IGD_HIP_MAX_BATCH. */
int64_t igd_hip_max_batch(void)
{
    return igd_hip_max_batch_rule(getenv("IGD_HIP_MAX_BATCH"));    /* (include/igd_hip.h: the one definition) */
}

int igd_hip_open(const igd_hip_desc *desc, int device, igd_hip_db **out)
{
    typedef int (*fn_t)(const igd_hip_desc *, int, igd_hip_db **);
    RESOLVE(fn_t, "igd_hip_open");
    return fn ? fn(desc, device, out) : IGD_HIP_ERR_DEVICE;
}

void igd_hip_close(igd_hip_db *db)
{
    typedef void (*fn_t)(igd_hip_db *);
    if (!db) return;                                   /* closing nothing must not map the engine */
    RESOLVE(fn_t, "igd_hip_close");
    if (fn) fn(db);
}

int32_t igd_hip_nfiles(const igd_hip_db *db)
{
    typedef int32_t (*fn_t)(const igd_hip_db *);
    if (!db) return 0;
    RESOLVE(fn_t, "igd_hip_nfiles");
    return fn ? fn(db) : 0;
}

int igd_hip_search(igd_hip_db *db, const int32_t *ichr, const int32_t *qs, const int32_t *qe, int64_t nq, int32_t v, int rule,
                   int64_t *hits, int64_t *total)
{
    typedef int (*fn_t)(igd_hip_db *, const int32_t *, const int32_t *, const int32_t *, int64_t, int32_t, int, int64_t *, int64_t *);
    RESOLVE(fn_t, "igd_hip_search");
    return fn ? fn(db, ichr, qs, qe, nq, v, rule, hits, total) : IGD_HIP_ERR_DEVICE;
}

int igd_hip_search_ex(igd_hip_db *db, const int32_t *ichr, const int32_t *qs, const int32_t *qe, int64_t nq, int32_t v, int rule,
                      int flags, int64_t *hits, int64_t *total)
{
    typedef int (*fn_t)(igd_hip_db *, const int32_t *, const int32_t *, const int32_t *, int64_t, int32_t, int, int, int64_t *, int64_t *);
    RESOLVE(fn_t, "igd_hip_search_ex");
    return fn ? fn(db, ichr, qs, qe, nq, v, rule, flags, hits, total) : IGD_HIP_ERR_DEVICE;
}

int igd_hip_enumerate_stream(igd_hip_db *db, const int32_t *ichr, const int32_t *qs, const int32_t *qe, int64_t nq, int64_t *qoff,
                             igd_hip_enum_sink sink, void *ctx, int64_t *total)
{
    typedef int (*fn_t)(igd_hip_db *, const int32_t *, const int32_t *, const int32_t *, int64_t, int64_t *, igd_hip_enum_sink, void *, int64_t *);
    RESOLVE(fn_t, "igd_hip_enumerate_stream");
    return fn ? fn(db, ichr, qs, qe, nq, qoff, sink, ctx, total) : IGD_HIP_ERR_DEVICE;
}

int igd_hip_enumerate_stream8(igd_hip_db *db, const int32_t *ichr, const int32_t *qs, const int32_t *qe, int64_t nq, int64_t *qoff,
                              igd_hip_enum_sink8 sink, void *ctx, int64_t *total)
{
    typedef int (*fn_t)(igd_hip_db *, const int32_t *, const int32_t *, const int32_t *, int64_t, int64_t *, igd_hip_enum_sink8, void *, int64_t *);
    RESOLVE(fn_t, "igd_hip_enumerate_stream8");
    return fn ? fn(db, ichr, qs, qe, nq, qoff, sink, ctx, total) : IGD_HIP_ERR_DEVICE;
}

int igd_hip_hit8_idx_bits(igd_hip_db *db)
{
    typedef int (*fn_t)(igd_hip_db *);
    if (!db) return -1;
    RESOLVE(fn_t, "igd_hip_hit8_idx_bits");
    return fn ? fn(db) : -1;
}

int igd_hip_hitmap(igd_hip_db *db, int use_v, int32_t v, uint32_t *hitmap, int64_t *total)
{
    typedef int (*fn_t)(igd_hip_db *, int, int32_t, uint32_t *, int64_t *);
    RESOLVE(fn_t, "igd_hip_hitmap");
    return fn ? fn(db, use_v, v, hitmap, total) : IGD_HIP_ERR_DEVICE;
}

int igd_hip_seqpare_add(igd_hip_db *db, const int32_t *ichr, const int32_t *qs, const int32_t *qe, int64_t nq, const int32_t *qgroup,
                        int32_t nGroups, double *sums)
{
    typedef int (*fn_t)(igd_hip_db *, const int32_t *, const int32_t *, const int32_t *, int64_t, const int32_t *, int32_t, double *);
    RESOLVE(fn_t, "igd_hip_seqpare_add");
    return fn ? fn(db, ichr, qs, qe, nq, qgroup, nGroups, sums) : IGD_HIP_ERR_DEVICE;
}

int igd_hip_create(const igd_hip_create_desc *d, int device, igd_hip_created *out)
{
    typedef int (*fn_t)(const igd_hip_create_desc *, int, igd_hip_created *);
    RESOLVE(fn_t, "igd_hip_create");
    return fn ? fn(d, device, out) : IGD_HIP_ERR_DEVICE;
}

void igd_hip_created_free(igd_hip_created *c)
{
    typedef void (*fn_t)(igd_hip_created *);
    RESOLVE(fn_t, "igd_hip_created_free");
    if (fn) fn(c);
}

int igd_hip_group_create(igd_hip_db *const *dbs, int n, igd_hip_group **out)
{
    typedef int (*fn_t)(igd_hip_db *const *, int, igd_hip_group **);
    RESOLVE(fn_t, "igd_hip_group_create");
    return fn ? fn(dbs, n, out) : IGD_HIP_ERR_DEVICE;
}

void igd_hip_group_destroy(igd_hip_group *g)
{
    typedef void (*fn_t)(igd_hip_group *);
    if (!g) return;
    RESOLVE(fn_t, "igd_hip_group_destroy");
    if (fn) fn(g);
}

const char *igd_hip_group_reduce_kind(const igd_hip_group *g)
{
    typedef const char *(*fn_t)(const igd_hip_group *);
    RESOLVE(fn_t, "igd_hip_group_reduce_kind");
    return fn ? fn(g) : "";
}

const char *igd_hip_group_reduce_note(const igd_hip_group *g)
{
    typedef const char *(*fn_t)(const igd_hip_group *);
    RESOLVE(fn_t, "igd_hip_group_reduce_note");
    return fn ? fn(g) : "";
}

int igd_hip_group_search(igd_hip_group *g, const int32_t *ichr, const int32_t *qs, const int32_t *qe, int64_t nq, int32_t v, int rule,
                         int flags, int64_t *hits, int64_t *total)
{
    typedef int (*fn_t)(igd_hip_group *, const int32_t *, const int32_t *, const int32_t *, int64_t, int32_t, int, int, int64_t *, int64_t *);
    RESOLVE(fn_t, "igd_hip_group_search");
    return fn ? fn(g, ichr, qs, qe, nq, v, rule, flags, hits, total) : IGD_HIP_ERR_DEVICE;
}
