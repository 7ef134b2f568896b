// engine/host_group.hpp -- part of igd_hip.hip (included there once; not a stand-alone header).
// device groups of one process: native RCCL all-reduce of hits[]
// ------------------------------------------------------------------------------------------
// Several devices driven by ONE process (the C host's form of SURVEY.md 8e; the process-per-GPU form is bench.py over
// torch.distributed).  The database is resident on every device of the group, a query set is cut into contiguous slabs,
// one per device, and the path's ONE exchange -- the sum of the nFiles-long hits[] vectors, which the reference keeps as one
// accumulator over all queries and prints once (src/igd_search.c:925,1032-1039) -- is an RCCL all-reduce over xGMI,
// enqueued on every engine's own stream right behind its kernels: ncclAllReduce(d_hits, nFiles, ncclInt64, ncclSum).
// librccl is mapped only when a group is created (it is large, and a one-device `igd search` has no use for it).  If it
// cannot be loaded or the communicators cannot be built, the vectors are added on the host instead (SURVEY.md section 5's
// fallback; igd_hip_group_reduce_kind() says which of the two a group uses, IGD_MULTI_REDUCE=host forces the second).
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <thread>
struct RcclApi {
    void *lib;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*GroupStart)(void);
    ncclResult_t (*GroupEnd)(void);
    const char *(*GetErrorString)(ncclResult_t);
};
static RcclApi *rccl_api(void)
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, []() {
        memset(&api, 0, sizeof api);
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.lib) break;
        }
        if (!api.lib) return;
        api.CommInitAll = (decltype(api.CommInitAll))dlsym(api.lib, "ncclCommInitAll");
        api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
        api.AllReduce = (decltype(api.AllReduce))dlsym(api.lib, "ncclAllReduce");
        api.GroupStart = (decltype(api.GroupStart))dlsym(api.lib, "ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))dlsym(api.lib, "ncclGroupEnd");
        api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
        if (!api.CommInitAll || !api.CommDestroy || !api.AllReduce || !api.GroupStart || !api.GroupEnd) { dlclose(api.lib); api.lib = nullptr; }
    });
    return api.lib ? &api : nullptr;
}

#define IGD_GROUP_MAX 16
struct igd_hip_group {
    int n;
    igd_hip_db *db[IGD_GROUP_MAX];
    ncclComm_t comm[IGD_GROUP_MAX];
    bool rccl;
    char why[256];                // why the host add is used instead of RCCL (empty: RCCL is)
};

extern "C" int igd_hip_group_create(igd_hip_db *const *dbs, int n, igd_hip_group **out)
{
    if (!dbs || !out || n < 1 || n > IGD_GROUP_MAX) { snprintf(g_err, sizeof g_err, "igd_hip_group_create: bad argument"); return IGD_HIP_ERR_ARG; }
    for (int r = 0; r < n; r++)
        if (!dbs[r] || dbs[r]->nFiles != dbs[0]->nFiles) { snprintf(g_err, sizeof g_err, "igd_hip_group_create: the databases differ"); return IGD_HIP_ERR_ARG; }
    igd_hip_group *g = new (std::nothrow) igd_hip_group();
    if (!g) return IGD_HIP_ERR_NOMEM;
    g->n = n; g->rccl = false; g->why[0] = 0;
    int devs[IGD_GROUP_MAX];
    bool distinct = true;
    for (int r = 0; r < n; r++) {
        g->db[r] = dbs[r]; g->comm[r] = nullptr; devs[r] = dbs[r]->device;
        for (int k = 0; k < r; k++) if (devs[k] == devs[r]) distinct = false;
    }
    const char *how = getenv("IGD_MULTI_REDUCE");
    if (how && !strcmp(how, "host")) snprintf(g->why, sizeof g->why, "IGD_MULTI_REDUCE=host");
    else if (!distinct) snprintf(g->why, sizeof g->why, "a device is listed twice: RCCL wants one rank per GPU");
    else if (RcclApi *A = rccl_api()) {
        // RCCL announces its version on STDOUT when the first communicator is built -- stdout is the command line tool's
        // result (the reference's table): file descriptor 1 points at stderr while the communicators are made
        fflush(stdout);
        const int keep = dup(1);
        if (keep >= 0) (void)dup2(2, 1);
        const ncclResult_t e = A->CommInitAll(g->comm, n, devs);
        fflush(stdout);
        if (keep >= 0) { (void)dup2(keep, 1); close(keep); }
        if (e == ncclSuccess) g->rccl = true;
        else {
            snprintf(g->why, sizeof g->why, "ncclCommInitAll: %s", A->GetErrorString ? A->GetErrorString(e) : "failed");
            for (int r = 0; r < n; r++) g->comm[r] = nullptr;
        }
    } else {
        const char *de = dlerror();                          // (once: the call clears the error it returns)
        snprintf(g->why, sizeof g->why, "librccl could not be loaded: %s", de ? de : "?");
    }
    // a multi-device job whose one exchange is NOT the RCCL all-reduce says so once, whether or not IGD_TIMING is set
    if (!g->rccl && distinct && !(how && !strcmp(how, "host")))
        fprintf(stderr, "igd: %d devices, but hits[] is summed on the host (%s)\n", n, g->why);
    if (!g->rccl && how && !strcmp(how, "rccl")) {           // the caller insists: no silent host add
        snprintf(g_err, sizeof g_err, "igd_hip_group_create: IGD_MULTI_REDUCE=rccl but %s", g->why);
        delete g;
        return IGD_HIP_ERR_DEVICE;
    }
    *out = g;
    return IGD_HIP_OK;
}

extern "C" void igd_hip_group_destroy(igd_hip_group *g)
{
    if (!g) return;
    if (g->rccl) if (RcclApi *A = rccl_api()) for (int r = 0; r < g->n; r++) if (g->comm[r]) (void)A->CommDestroy(g->comm[r]);
    delete g;
}

extern "C" const char *igd_hip_group_reduce_kind(const igd_hip_group *g) { return !g ? "" : g->rccl ? "rccl" : "host"; }
extern "C" const char *igd_hip_group_reduce_note(const igd_hip_group *g) { return g ? g->why : ""; }

extern "C" int igd_hip_group_search(igd_hip_group *g, const int32_t *ichr, const int32_t *qs, const int32_t *qe, int64_t nq,
                                    int32_t v, int rule, int flags, int64_t *hits, int64_t *total)
{
    if (!g || !hits || nq < 0 || (nq > 0 && (!ichr || !qs || !qe))) { snprintf(g_err, sizeof g_err, "igd_hip_group_search: bad argument"); return IGD_HIP_ERR_ARG; }
    if (total) *total = 0;
    const int n = g->n;
    const int32_t nf = g->db[0]->nFiles;
    if (nq == 0 || nf == 0) return IGD_HIP_OK;
    // contiguous slabs (the rule of igd_amd/dist.py shard_bounds), one host thread per device; an empty slab leaves zeros
    int rcs[IGD_GROUP_MAX];
    char errs[IGD_GROUP_MAX][256];
    std::thread th[IGD_GROUP_MAX];
    const int64_t base = nq / n, rem = nq % n;
    auto work = [&](int r) {
        const int64_t lo = r * base + (r < rem ? r : rem), m = base + (r < rem ? 1 : 0);
        rcs[r] = search_slab_resident(g->db[r], ichr + lo, qs + lo, qe + lo, m, v, rule, flags);
        errs[r][0] = 0;
        if (rcs[r] != IGD_HIP_OK) snprintf(errs[r], sizeof errs[r], "%s", g_err);      // thread-local text
    };
    bool threaded[IGD_GROUP_MAX] = {false};
    for (int r = 1; r < n; r++) {
        try { th[r] = std::thread(work, r); threaded[r] = true; }
        catch (...) { threaded[r] = false; }                 // (no thread to be had: the slab runs on this one -- nothing is thrown across the C boundary)
    }
    work(0);
    for (int r = 1; r < n; r++) if (!threaded[r]) work(r);
    int rc = IGD_HIP_OK;
    for (int r = 0; r < n; r++) {
        if (r > 0 && threaded[r]) th[r].join();
        if (rcs[r] != IGD_HIP_OK && rc == IGD_HIP_OK) { rc = rcs[r]; snprintf(g_err, sizeof g_err, "%s", errs[r]); }
    }
    if (rc != IGD_HIP_OK) return rc;
    std::vector<int64_t> h((size_t)nf);
    int64_t tot = 0;
    if (g->rccl) {
        RcclApi *A = rccl_api();
        // the one exchange of the path: every device ends up with the sum, device 0's copy is returned
        ncclResult_t e = A->GroupStart();
        for (int r = 0; r < n && e == ncclSuccess; r++) {
            e = A->AllReduce(g->db[r]->d_hits, g->db[r]->d_hits, (size_t)nf, ncclInt64, ncclSum, g->comm[r], g->db[r]->stream);
            if (e == ncclSuccess) e = A->AllReduce(g->db[r]->d_total, g->db[r]->d_total, 1, ncclInt64, ncclSum, g->comm[r], g->db[r]->stream);
        }
        const ncclResult_t e2 = A->GroupEnd();
        if (e == ncclSuccess) e = e2;
        if (e != ncclSuccess) { snprintf(g_err, sizeof g_err, "igd_hip_group_search: ncclAllReduce: %s", A->GetErrorString ? A->GetErrorString(e) : "failed"); return IGD_HIP_ERR_DEVICE; }
        for (int r = 0; r < n; r++) { HIPCHK(hipSetDevice(g->db[r]->device)); HIPCHK(hipStreamSynchronize(g->db[r]->stream)); }
        HIPCHK(hipSetDevice(g->db[0]->device));
        HIPCHK(hipMemcpy(h.data(), g->db[0]->d_hits, (size_t)nf * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(&tot, g->db[0]->d_total, 8, hipMemcpyDeviceToHost));
        for (int32_t f = 0; f < nf; f++) hits[f] += h[f];
    } else {
        for (int r = 0; r < n; r++) {
            int64_t t1 = 0;
            HIPCHK(hipSetDevice(g->db[r]->device));
            // (the engine streams do not wait for the null stream's copies: an EMPTY slab has only enqueued the clearing of
            // its d_hits -- without this the previous search's counts could be read and added again)
            HIPCHK(hipStreamSynchronize(g->db[r]->stream));
            HIPCHK(hipMemcpy(h.data(), g->db[r]->d_hits, (size_t)nf * 8, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(&t1, g->db[r]->d_total, 8, hipMemcpyDeviceToHost));
            for (int32_t f = 0; f < nf; f++) hits[f] += h[f];
            tot += t1;
        }
    }
    if (total) *total = tot;
    return IGD_HIP_OK;
}
