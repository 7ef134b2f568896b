// engine/measure.hpp -- part of igd_hip.hip (included there once; not a stand-alone header).
// instrumentation: compulsory traffic, streaming rates of the box, launch profile
// ------------------------------------------------------------------------------------------
// instrumentation: compulsory traffic of the scan kernel for one batch (include/igd_hip.h)
__global__ void k_unit_traffic(DbView db, const int32_t *__restrict__ firstQ, const int32_t *__restrict__ pairN,
                               const int32_t *__restrict__ spill, int epoch, int path /* 0 bucket, 1 merge join exact, 2 merge join compact */,
                               int rankOK, u64 *__restrict__ acc /* units, records, pairs (bucket path), queries of rank-method tiles */)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    u64 nu = 0, nr = 0, np = 0, nd = 0;
    if (u < db.nUnits) {
        const Unit un = db.units[u];
        if (un.n > 0) {
            const int lj = UNIT_J(un);
            const int lb = lj < IGD_SHORT_TILES - 1 ? lj : IGD_SHORT_TILES - 1;
            const bool firstUnit = UNIT_FLAGS(un) & 1;
            if (path == 1) {        // exactly the test of issue_unit: the candidate range of the unit's tile is not empty
                if (firstQ[un.tile + 1] > firstQ[un.tile - lb]) { nu = 1; nr = (u64)un.n; }
            } else if (path == 2) { // exactly the test of s_issue: first-tile queries, or a marked tile with earlier queries
                const int f0 = firstQ[un.tile], c0 = firstQ[un.tile + 1] - f0;
                const int cl = spill[un.tile] == epoch ? f0 - firstQ[un.tile - lb] : 0;
                if (c0 | cl) { nu = 1; nr = (u64)un.n; }
                if (firstUnit) {
                    if (rankOK && c0 >= IGD_DENSE_MIN) nd = (u64)c0;
                }
            } else if (pairN[un.tile] != 0) {             // negative: the tile went to heavy_bucket_body
                nu = 1; nr = (u64)un.n;
                if (firstUnit) np = (u64)(pairN[un.tile] < 0 ? -pairN[un.tile] : pairN[un.tile]);
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        nu += __shfl_down(nu, o); nr += __shfl_down(nr, o); np += __shfl_down(np, o); nd += __shfl_down(nd, o);
    }
    if ((threadIdx.x & 63) == 0) {
        if (nu) atomicAdd(&acc[0], nu);
        if (nr) atomicAdd(&acc[1], nr);
        if (np) atomicAdd(&acc[2], np);
        if (nd) atomicAdd(&acc[3], nd);
    }
}

// ... of igd_scan_direct: a visited unit's records, the records of the next tile that ride with a tile's first unit, and the
// tile's queries once per unit of the tile (8 bytes each: q_qs, q_qe)
__global__ void k_unit_traffic_direct(DbView db, const int32_t *__restrict__ firstQ, int rule, u64 *__restrict__ acc /* units, records, -, queries read */)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    u64 nu = 0, nr = 0, nd = 0;
    if (u < db.nUnits) {
        const Unit un = db.units[u];
        const int c0 = firstQ[un.tile + 1] - firstQ[un.tile];
        const bool first = UNIT_FLAGS(un) & 1;
        if (c0 > 0 && c0 <= IGD_HEAVY_FIRST && (un.n > 0 || first)) {
            nd = (u64)c0;                                 // (a placeholder still reads -- checks -- its queries)
            const bool dead = rule == IGD_HIP_RULE_NEST && un.n == 0;
            if (un.n > 0 || !dead) { nu = 1; nr = (u64)un.n + (first && !dead ? (u64)(db.tileD[un.tile].y & 127) : 0); }
        }
    }
    for (int o = 32; o > 0; o >>= 1) { nu += __shfl_down(nu, o); nr += __shfl_down(nr, o); nd += __shfl_down(nd, o); }
    if ((threadIdx.x & 63) == 0) {
        if (nu) atomicAdd(&acc[0], nu);
        if (nr) atomicAdd(&acc[1], nr);
        if (nd) atomicAdd(&acc[3], nd);
    }
}

extern "C" int igd_hip_batch_traffic(igd_hip_db *db, const int32_t *d_ichr, const int32_t *d_qs, const int32_t *d_qe,
                                     int64_t nq, int32_t v, int rule, int flags, igd_hip_traffic *out)
{
    if (!db || !out || nq < 0 || nq > IGD_MAX_BATCH) return IGD_HIP_ERR_ARG;
    if (db->inner) {                                     // (the kernels of such a database run on its re-tiled copy)
        db->inner->vnest = rule == IGD_HIP_RULE_NEST ? 1 : 0;
        return igd_hip_batch_traffic(db->inner, d_ichr, d_qs, d_qe, nq, v, IGD_HIP_RULE_FLAT, flags, out);
    }
    memset(out, 0, sizeof *out);
    if (nq == 0 || db->nT == 0 || db->nFiles == 0) return IGD_HIP_OK;
    HIPCHK(hipSetDevice(db->device));
    u64 *d_acc = nullptr;
    int64_t *d_h = nullptr;
    int rc;
    // an order promise of the caller's own batches is settled first: the sync that closes this measurement clears the
    // device's sticky report, and no broken batch may go unreported
    if (db->promised && (rc = igd_hip_sync(db, db->stream)) != IGD_HIP_OK) return rc;
    if ((rc = dalloc(&d_acc, 4, nullptr)) != IGD_HIP_OK) return rc;
    if ((rc = dalloc(&d_h, (size_t)db->nFiles + 1, nullptr)) != IGD_HIP_OK) { (void)hipFree(d_acc); return rc; }
    hipStream_t st = db->stream;
    (void)hipMemsetAsync(d_acc, 0, 32, st);
    (void)hipMemsetAsync(d_h, 0, ((size_t)db->nFiles + 1) * 8, st);
    const bool saved = db->evOn;
    db->evOn = false;
    rc = igd_hip_search_dev(db, d_ichr, d_qs, d_qe, nq, v, rule, flags & ~IGD_HIP_FLAG_ZERO_FIRST, d_h, nullptr, st);
    db->evOn = saved;
    int32_t ctl[4] = {0, 0, 0, 0};
    hipError_t e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipMemcpy(ctl, db->d_ctl, sizeof ctl, hipMemcpyDeviceToHost);
    if (rc == IGD_HIP_OK && e == hipSuccess) {
        const int mode = (flags & IGD_HIP_FLAG_SORTED) ? 1 : (flags & IGD_HIP_FLAG_BUCKET) ? 2 : 0;
        const bool sortedPath = mode == 1 || (mode == 0 && ctl[CTL_UNSORTED] != db->epoch);
        const bool useV = (v != IGD_HIP_NO_VALUE_FILTER && db->gType == 1);
        const bool packed = db->packed && !(flags & IGD_HIP_FLAG_EXACT) && (!useV || db->packedV);
        const int path = !sortedPath ? 0 : packed ? 2 : 1;
        const bool direct = db->lastDirect != 0 && sortedPath;
        if (direct) k_unit_traffic_direct<<<(db->nUnits + 255) / 256, 256, 0, st>>>(db->v, db->d_firstQ, rule, d_acc);
        else
        k_unit_traffic<<<(db->nUnits + 255) / 256, 256, 0, st>>>(db->v, db->d_firstQ, db->d_pairN, db->d_spill, db->epoch, path,
                                                               ctl[CTL_NOTSTART] != db->epoch ? 1 : 0, d_acc);
        u64 acc[4] = {0, 0, 0, 0};
        e = hipStreamSynchronize(st);
        if (e == hipSuccess) e = hipMemcpy(acc, d_acc, 32, hipMemcpyDeviceToHost);
        const int recB = packed ? (useV ? 8 : 6) : (useV ? 16 : 12);
        out->units = (int64_t)acc[0];
        out->records = (int64_t)acc[1];
        out->record_bytes = (int64_t)acc[1] * recB;
        out->unit_bytes = (int64_t)sizeof(Unit) * db->nUnits + (path == 2 ? 8ll * (db->nT + 1) : sortedPath ? 4ll * (db->nT + 1) : 8ll * db->nT);
        if (direct) out->unit_bytes = (int64_t)(sizeof(Unit) + 8 + 16) * db->nUnits;      // per unit: its descriptor, two entries of firstQ[], its tile's int4 of tileD[]
        // merge join, compact image: one 4-byte word per query (qw0), the compacted later-tile words (later[]: every entry is
        // read at least once), the starts (q_qs) of the tiles the rank method handles; exact arrays: qw, qs, qe;
        // bucket path: 8 B per pair
        int64_t nLaterWords = 0;
        if (path == 2 && !direct) {
            const int64_t nb = (nq + ((int64_t)1 << db->lbShift) - 1) >> db->lbShift;
            std::vector<int32_t> hdr((size_t)nb * 2);
            e = hipMemcpy(hdr.data(), db->d_laterHdr, (size_t)nb * 8, hipMemcpyDeviceToHost);
            for (int64_t b = 0; b < nb; b++) nLaterWords += hdr[(size_t)b * 2];
        }
        out->query_bytes = direct ? 8ll * (int64_t)acc[3] : path == 2 ? 4ll * nq + 4ll * nLaterWords + 4ll * (int64_t)acc[3]
                         : path == 1 ? 12ll * nq : 8ll * (int64_t)acc[2];
        out->slab_bytes = db->ldsHits ? (int64_t)db->grid * db->nFiles * (path == 2 ? 4 : 8) : 8ll * db->nFiles;   // (merge join: 32-bit rows)
        out->total = out->record_bytes + out->unit_bytes + out->query_bytes + out->slab_bytes;
    }
    (void)hipFree(d_acc); (void)hipFree(d_h);
    (void)igd_hip_sync(db, st);                            // consumes a promise made for this measurement batch
    if (rc != IGD_HIP_OK) return rc;
    if (e != hipSuccess) { set_err("batch_traffic", e, __FILE__, __LINE__); return IGD_HIP_ERR_DEVICE; }
    return IGD_HIP_OK;
}

// ------------------------------------------------------------------------------------------
// instrumentation: what the memory system of THIS box gives simple streaming kernels (bench.py quotes
// it next to the roofline): a float4 copy (the guide's 6.29 TB/s measurement), a float4 read-only sum,
// pinned D2H / H2D copies.
__global__ __launch_bounds__(256) void k_copy16(const float4 *__restrict__ in, float4 *__restrict__ outp, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i + 3 * stride < n; i += 4 * stride) {          // four independent 16-byte loads in flight per lane
        const float4 a = in[i], b = in[i + stride], c = in[i + 2 * stride], d = in[i + 3 * stride];
        outp[i] = a; outp[i + stride] = b; outp[i + 2 * stride] = c; outp[i + 3 * stride] = d;
    }
    for (; i < n; i += stride) outp[i] = in[i];
}
__global__ __launch_bounds__(256) void k_read16(const float4 *__restrict__ in, size_t n, float *__restrict__ sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const float4 a = in[i], b = in[i + stride], c = in[i + 2 * stride], d = in[i + 3 * stride];
        acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + c.x + c.y + c.z + c.w + d.x + d.y + d.z + d.w;
    }
    for (; i < n; i += stride) { const float4 a = in[i]; acc += a.x + a.y + a.z + a.w; }
    if (acc == 123.456f) *sink = acc;                       // never true for the zero-filled buffer; keeps the loads
}

// one 16-byte vector per thread, non-temporal: the plainest streaming copy / read there is
typedef float igd_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_copy16_flat(const float4 *__restrict__ in, float4 *__restrict__ outp, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(__builtin_nontemporal_load((const igd_f4 *)in + i), (igd_f4 *)outp + i);
}
__global__ __launch_bounds__(256) void k_read16_flat(const float4 *__restrict__ in, size_t n, float *__restrict__ sink)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const igd_f4 a = __builtin_nontemporal_load((const igd_f4 *)in + i);
        if (a.x + a.y + a.z + a.w == 123.456f) *sink = a.x;
    }
}

extern "C" int igd_hip_measure_rates(int device, double rates[4])
{
    if (!rates) return IGD_HIP_ERR_ARG;
    for (int k = 0; k < 4; k++) rates[k] = 0.0;
    HIPCHK(hipSetDevice(device));
    const size_t bytes = (size_t)1 << 30, n16 = bytes / 16, hb = (size_t)256 << 20;
    float4 *a = nullptr, *b = nullptr;
    float *sink = nullptr;
    void *h = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t st = nullptr;
    hipError_t e = hipMalloc((void **)&a, bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&b, bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&sink, 16);
    if (e == hipSuccess) e = hipHostMalloc(&h, hb, hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMemsetAsync(a, 0, bytes, st);
    if (e == hipSuccess) e = hipMemsetAsync(b, 0, bytes, st);
    if (e == hipSuccess) memset(h, 0, hb);
    const int grid = 256 * 16, reps = 12;
    for (int which = 0; which < 4 && e == hipSuccess; which++) {
        float best = 1e30f;
        for (int r = 0; r < reps + 2 && e == hipSuccess; r++) {
            e = hipEventRecord(e0, st);
            // two shapes of each kernel, alternating: grid-stride with four loads in flight / one float4 per thread; best wins
            if (which == 0) { if (r & 1) k_copy16_flat<<<(unsigned)((n16 + 255) / 256), 256, 0, st>>>(a, b, n16); else k_copy16<<<grid, 256, 0, st>>>(a, b, n16); }
            else if (which == 1) { if (r & 1) k_read16_flat<<<(unsigned)((n16 + 255) / 256), 256, 0, st>>>(a, n16, sink); else k_read16<<<grid, 256, 0, st>>>(a, n16, sink); }
            else if (which == 2) { if (e == hipSuccess) e = hipMemcpyAsync(h, a, hb, hipMemcpyDeviceToHost, st); }
            else { if (e == hipSuccess) e = hipMemcpyAsync(a, h, hb, hipMemcpyHostToDevice, st); }
            if (e == hipSuccess) e = hipEventRecord(e1, st);
            if (e == hipSuccess) e = hipEventSynchronize(e1);
            float ms = 0.f;
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
            if (r >= 2 && ms < best) best = ms;
        }
        const double moved = which == 0 ? 2.0 * (double)bytes : which == 1 ? (double)bytes : (double)hb;
        if (e == hipSuccess && best > 0.f) rates[which] = moved / ((double)best * 1e-3) / 1e9;
    }
    if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    if (sink) (void)hipFree(sink);
    if (h) (void)hipHostFree(h);
    if (e != hipSuccess) { set_err("measure_rates", e, __FILE__, __LINE__); return IGD_HIP_ERR_DEVICE; }
    return IGD_HIP_OK;
}

extern "C" int igd_hip_profile_begin(igd_hip_db *db, int max_launches)
{
    if (!db || max_launches <= 0) return IGD_HIP_ERR_ARG;
    if (db->inner) return igd_hip_profile_begin(db->inner, max_launches);
    HIPCHK(hipSetDevice(db->device));
    while ((int)db->ev.size() < 4 * max_launches) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        db->ev.push_back(e);
    }
    db->evMax = max_launches;
    db->evUsed = 0;
    db->evSeen = 0;
    db->evOn = true;
    return IGD_HIP_OK;
}

extern "C" int igd_hip_profile_sampling(igd_hip_db *db, int every)
{
    if (!db || every < 1) return IGD_HIP_ERR_ARG;
    if (db->inner) return igd_hip_profile_sampling(db->inner, every);
    db->evEvery = every;
    return IGD_HIP_OK;
}

extern "C" int igd_hip_profile_end(igd_hip_db *db, int *n_launches, double *avg_scan_ms, double *avg_pipeline_ms)
{
    if (!db) return IGD_HIP_ERR_ARG;
    if (db->inner) return igd_hip_profile_end(db->inner, n_launches, avg_scan_ms, avg_pipeline_ms);
    HIPCHK(hipSetDevice(db->device));
    db->evOn = false;
    int n = db->evUsed;
    double scan = 0, pipe = 0;
    int npipe = 0;
    for (int i = 0; i < n; i++) {
        HIPCHK(hipEventSynchronize(db->ev[4 * i + 2]));
        float a = 0, b = 0;
        HIPCHK(hipEventElapsedTime(&a, db->ev[4 * i + 1], db->ev[4 * i + 2]));
        scan += a;
        if (i < IGD_PIPE_EVENTS) {
            HIPCHK(hipEventSynchronize(db->ev[4 * i + 3]));
            HIPCHK(hipEventElapsedTime(&b, db->ev[4 * i + 0], db->ev[4 * i + 3]));
            pipe += b; npipe++;
        }
    }
    if (n_launches) *n_launches = n;
    if (avg_scan_ms) *avg_scan_ms = n ? scan / n : 0.0;
    if (avg_pipeline_ms) *avg_pipeline_ms = npipe ? pipe / npipe : 0.0;
    db->evUsed = 0;
    return IGD_HIP_OK;
}
