// engine/host_misc.hpp -- part of igd_hip.hip (included there once; not a stand-alone header).
// igd_hip_hitmap, igd_hip_batch_stats
extern "C" int igd_hip_hitmap(igd_hip_db *db, int use_v, int32_t v, uint32_t *hitmap, int64_t *total)
{
    if (!db || !hitmap) {
        snprintf(g_err, sizeof g_err, "igd_hip_hitmap: bad argument");
        return IGD_HIP_ERR_ARG;
    }
    if (total) *total = 0;
    if (db->gType != 1) {
        snprintf(g_err, sizeof g_err, "igd_hip_hitmap: needs a gType-1 database (the reference's getMap reads 16-byte records)");
        return IGD_HIP_ERR_ARG;
    }
    const size_t cells = (size_t)db->nFiles * (size_t)db->nFiles;
    if (cells == 0 || db->nT == 0) return IGD_HIP_OK;
    if (cells * 4 > ((size_t)64 << 30)) {
        snprintf(g_err, sizeof g_err, "igd_hip_hitmap: %d x %d matrix does not fit", db->nFiles, db->nFiles);
        return IGD_HIP_ERR_NOMEM;
    }
    HIPCHK(hipSetDevice(db->device));
    uint32_t *d_map = nullptr;
    u64 *d_tot = nullptr;
    int rc;
    if ((rc = dalloc(&d_map, cells, nullptr)) != IGD_HIP_OK) return rc;
    if ((rc = dalloc(&d_tot, 1, nullptr)) != IGD_HIP_OK) { (void)hipFree(d_map); return rc; }
    hipStream_t st = db->stream;
    hipError_t e = hipMemsetAsync(d_map, 0, cells * 4, st);
    if (e == hipSuccess) e = hipMemsetAsync(d_tot, 0, 8, st);
    if (e == hipSuccess) {
        int grid = db->nT < 256 * 16 ? db->nT : 256 * 16;
        if (use_v) igd_hitmap_tiles<true><<<grid, IGD_MAP_WG, 0, st>>>(db->v, v, d_map, d_tot);
        else igd_hitmap_tiles<false><<<grid, IGD_MAP_WG, 0, st>>>(db->v, v, d_map, d_tot);
        e = hipGetLastError();
    }
    std::vector<uint32_t> h;
    u64 tot = 0;
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) { h.resize(cells); e = hipMemcpy(h.data(), d_map, cells * 4, hipMemcpyDeviceToHost); }
    if (e == hipSuccess) e = hipMemcpy(&tot, d_tot, 8, hipMemcpyDeviceToHost);
    (void)hipFree(d_map); (void)hipFree(d_tot);
    if (e != hipSuccess) { set_err("hitmap", e, __FILE__, __LINE__); return IGD_HIP_ERR_DEVICE; }
    for (size_t c = 0; c < cells; c++) hitmap[c] += h[c];            // hitmap[][]++ semantics: added to
    if (total) *total = (int64_t)tot;
    return IGD_HIP_OK;
}

extern "C" int igd_hip_batch_stats(igd_hip_db *db, const int32_t *d_ichr, const int32_t *d_qs,
                                   const int32_t *d_qe, int64_t nq, int32_t v, int rule, igd_hip_stats *out)
{
    if (!db || !out || nq < 0 || nq > IGD_MAX_BATCH) return IGD_HIP_ERR_ARG;
    memset(out, 0, sizeof *out);
    if (nq == 0) return IGD_HIP_OK;
    HIPCHK(hipSetDevice(db->device));
    u64 *d_acc = nullptr;
    int64_t *d_h = nullptr, *d_t = nullptr;
    int rc;
    if ((rc = dalloc(&d_acc, 4, nullptr)) != IGD_HIP_OK) return rc;
    if ((rc = dalloc(&d_h, (size_t)db->nFiles + 1, nullptr)) != IGD_HIP_OK) { (void)hipFree(d_acc); return rc; }
    if ((rc = dalloc(&d_t, 1, nullptr)) != IGD_HIP_OK) { (void)hipFree(d_acc); (void)hipFree(d_h); return rc; }
    hipStream_t st = db->stream;
    (void)hipMemsetAsync(d_acc, 0, 32, st);
    (void)hipMemsetAsync(d_h, 0, ((size_t)db->nFiles + 1) * 8, st);
    (void)hipMemsetAsync(d_t, 0, 8, st);
    k_batch_stats<<<(int)((nq + 255) / 256), 256, 0, st>>>(db->v, d_ichr, d_qs, d_qe, (int)nq, rule, d_acc);
    bool saved = db->evOn;
    db->evOn = false;
    rc = igd_hip_search_dev(db, d_ichr, d_qs, d_qe, nq, v, rule, 0, d_h, d_t, st);
    db->evOn = saved;
    u64 acc[4] = {0, 0, 0, 0};
    int64_t tot = 0;
    hipError_t e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipMemcpy(acc, d_acc, 32, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(&tot, d_t, 8, hipMemcpyDeviceToHost);
    (void)hipFree(d_acc); (void)hipFree(d_h); (void)hipFree(d_t);
    if (rc != IGD_HIP_OK) return rc;
    if (e != hipSuccess) { set_err("batch_stats", e, __FILE__, __LINE__); return IGD_HIP_ERR_DEVICE; }
    out->queries = (int64_t)acc[0]; out->pairs = (int64_t)acc[1];
    out->S = (int64_t)acc[2]; out->B = (int64_t)acc[3]; out->H = tot;
    return IGD_HIP_OK;
}
