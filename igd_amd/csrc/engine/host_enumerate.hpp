// engine/host_enumerate.hpp -- part of igd_hip.hip (included there once; not a stand-alone header).
// `-f` on the host side: chunked, double-buffered
// `-f` on the host side.  The path is bound by the 16 bytes per overlap that cross PCIe (and, in the
// command line tool, by turning them into text), so the result is produced in CHUNKS of contiguous
// query ranges and streamed: while chunk k's device->host copy runs on the copy stream into one of
// two pinned buffers, chunk k+1 is being filled on the compute stream and the caller's sink is
// formatting chunk k-1.  Every buffer is part of a persistent workspace (no allocation per call).
//   COUNT pass over the whole batch -> qcount -> scan -> qoff (device + host)
//   chunks: the longest query range whose overlaps fit one buffer
//   per chunk: FILL [qa,qb) -> d_enumOut[k&1] -> async D2H -> pinned h_enumPin[k&1] (or the final array) -> sink
static int ensure_enum_workspace(igd_hip_db *db, int64_t nq, int64_t chunkHits, bool needPinned)
{
    int rc;
    if (nq > db->enumQCap) {
        HIPCHK(hipDeviceSynchronize());
        void *ps[] = {db->d_qcount, db->d_qoff, db->d_enumBsum};
        for (void *q : ps) if (q) (void)hipFree(q);
        db->d_qcount = db->d_qoff = db->d_enumBsum = nullptr;
        db->enumQCap = 0;
        if ((rc = dalloc(&db->d_qcount, (size_t)nq, nullptr)) != IGD_HIP_OK) return rc;
        if ((rc = dalloc(&db->d_qoff, (size_t)nq + 1, nullptr)) != IGD_HIP_OK) return rc;
        if ((rc = dalloc(&db->d_enumBsum, (size_t)(nq / IGD_SCAN_TILE + 2), nullptr)) != IGD_HIP_OK) return rc;
        db->enumQCap = nq;
    }
    if (!db->copyStream) {
        HIPCHK(hipStreamCreateWithFlags(&db->copyStream, hipStreamNonBlocking));
        for (int k = 0; k < 2; k++) {
            HIPCHK(hipEventCreateWithFlags(&db->evFill[k], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&db->evCopy[k], hipEventDisableTiming));
        }
    }
    if (chunkHits > db->enumChunkCap) {
        HIPCHK(hipDeviceSynchronize());
        for (int k = 0; k < 2; k++) {
            if (db->d_enumOut[k]) (void)hipFree(db->d_enumOut[k]);
            if (db->h_enumPin[k]) (void)hipHostFree(db->h_enumPin[k]);
            db->d_enumOut[k] = nullptr; db->h_enumPin[k] = nullptr;
        }
        db->enumChunkCap = 0; db->enumPinned = false;
        for (int k = 0; k < 2; k++)
            if ((rc = dalloc(&db->d_enumOut[k], (size_t)chunkHits, nullptr)) != IGD_HIP_OK) return rc;
        db->enumChunkCap = chunkHits;
    }
    if (needPinned && !db->enumPinned) {
        for (int k = 0; k < 2; k++)
            if (hipHostMalloc((void **)&db->h_enumPin[k], (size_t)db->enumChunkCap * sizeof(igd_hip_hit), hipHostMallocDefault) != hipSuccess) {
                snprintf(g_err, sizeof g_err, "igd_hip_enumerate: pinned host allocation failed");
                return IGD_HIP_ERR_NOMEM;
            }
        db->enumPinned = true;
    }
    return IGD_HIP_OK;
}

#define IGD_ENUM_CHUNK_HITS ((int64_t)2 << 20)     // 32 MiB of igd_hip_hit per chunk buffer (pinning memory costs ~0.3 ms per MiB)

// whole != nullptr: the chunks are copied straight to their place in `whole` (pinned, qoff[nq] records);
// otherwise every chunk is handed to `sink` from one of the two pinned chunk buffers.
// sink8 != nullptr: the packed stream -- 8 bytes per overlap in the same chunk buffers (twice the overlaps per chunk)
static int enumerate_core(igd_hip_db *db, const int32_t *ichr, const int32_t *qs, const int32_t *qe, int64_t nq,
                          int64_t *qoff, bool wantWhole, igd_hip_hit **wholeOut, igd_hip_enum_sink sink, void *ctx, int64_t *total,
                          igd_hip_enum_sink8 sink8 = nullptr)
{
    const bool p8 = sink8 != nullptr;
    const int idxBits = db->hit8Bits;
    const bool anySink = sink != nullptr || p8;
    const char *tenv = getenv("IGD_TIMING");
    const bool tim = tenv && *tenv && *tenv != '0';
    double t0 = wall_s();
#define ENUM_PHASE(name) do { if (tim) { double t_ = wall_s(); fprintf(stderr, "[igd timing]   enumerate: %-24s %8.2f ms\n", name, 1e3 * (t_ - t0)); t0 = t_; } } while (0)
    HIPCHK(hipSetDevice(db->device));
    hipStream_t st = db->stream;
    int rc = ensure_qstage(db, nq);
    if (rc != IGD_HIP_OK) return rc;
    int64_t chunkHits = IGD_ENUM_CHUNK_HITS;
    if (const char *ce = getenv("IGD_ENUM_CHUNK_HITS")) { if (atoll(ce) > 0) chunkHits = atoll(ce); }   // tests: many small chunks
    rc = ensure_enum_workspace(db, nq, db->enumChunkCap > chunkHits ? db->enumChunkCap : chunkHits, !wantWhole);
    if (rc != IGD_HIP_OK) return rc;
    ENUM_PHASE("workspace");
    HIPCHK(hipMemcpyAsync(db->d_qc, ichr, (size_t)nq * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(db->d_qs, qs, (size_t)nq * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(db->d_qe, qe, (size_t)nq * 4, hipMemcpyHostToDevice, st));
    const int egrid = db->grid * 4;                        // 256-thread workgroups: 8 per CU
    igd_enum_queries<false><<<egrid, 256, 0, st>>>(db->v, db->d_qc, db->d_qs, db->d_qe, 0, (int)nq, db->d_qcount, nullptr, 0, nullptr);
    {
        const int sb = (int)((nq + IGD_SCAN_TILE - 1) / IGD_SCAN_TILE);
        k_scan64_sums<<<sb, IGD_SCAN_BLOCK, 0, st>>>(db->d_qcount, (int)nq, db->d_enumBsum);
        k_scan64_apply<<<sb, IGD_SCAN_BLOCK, 0, st>>>(db->d_qcount, (int)nq, db->d_enumBsum, db->d_qoff);
    }
    HIPCHK(hipMemcpyAsync(qoff, db->d_qoff, ((size_t)nq + 1) * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    ENUM_PHASE("H2D + count + scan + qoff");
    const int64_t tot = qoff[nq];
    if (total) *total = tot;
    if (tot == 0) {
        // no overlap at all: the sink still sees the batch's queries once (the command line tool prints a line per query)
        if (anySink && (p8 ? sink8(ctx, 0, nq, qoff, nullptr, idxBits) : sink(ctx, 0, nq, qoff, nullptr)) != 0) {
            snprintf(g_err, sizeof g_err, "igd_hip_enumerate_stream: stopped by the sink");
            return IGD_HIP_ERR_ARG;
        }
        return IGD_HIP_OK;
    }
    int64_t maxq = 0;
    for (int64_t i = 0; i < nq; i++) if (qoff[i + 1] - qoff[i] > maxq) maxq = qoff[i + 1] - qoff[i];
    if (maxq > db->enumChunkCap * (p8 ? 2 : 1)) {          // one query larger than a chunk buffer: grow them
        rc = ensure_enum_workspace(db, nq, p8 ? (maxq + 1) / 2 : maxq, !wantWhole);
        if (rc != IGD_HIP_OK) return rc;
    }
    igd_hip_hit *whole = nullptr;
    if (wantWhole) {
        size_t got = 0;
        size_t *hdr = (size_t *)pinned_take((size_t)tot * sizeof(igd_hip_hit), &got);
        if (!hdr) { snprintf(g_err, sizeof g_err, "igd_hip_enumerate: pinned host allocation failed"); return IGD_HIP_ERR_NOMEM; }
        hdr[0] = got;
        whole = (igd_hip_hit *)((char *)hdr + 64);
        ENUM_PHASE("pinned result buffer");
    }
    static const bool zeroCopy = getenv("IGD_ENUM_ZEROCOPY") != nullptr;   // A/B: the fill kernel stores straight into pinned host memory
    const int64_t cap = db->enumChunkCap * (p8 ? 2 : 1);   // overlaps per chunk buffer
    const size_t hitBytes = p8 ? sizeof(igd_hip_hit8) : sizeof(igd_hip_hit);
    int64_t qa = 0, prevA = 0, prevB = 0;
    int k = 0;
    hipError_t e = hipSuccess;
    int sinkRc = 0;
    while (qa < nq && e == hipSuccess && sinkRc == 0) {
        int64_t qb = qa + 1;                               // longest range [qa,qb) whose overlaps fit the buffer
        {
            int64_t lo = qa + 1, hi = nq;                  // qoff is non-decreasing: bisect
            while (lo < hi) {
                const int64_t mid = lo + (hi - lo + 1) / 2;
                if (qoff[mid] - qoff[qa] <= cap) lo = mid; else hi = mid - 1;
            }
            qb = lo;
        }
        const int64_t nh = qoff[qb] - qoff[qa];
        const int b = k & 1;
        if (nh > 0) {
            igd_hip_hit *hostDst = whole ? whole + qoff[qa] : db->h_enumPin[b];
            if (k >= 2) e = hipStreamWaitEvent(st, db->evCopy[b], 0);          // the buffer's previous copy is done
            if (e != hipSuccess) break;
            if (p8)
                igd_enum_queries<true, false, true><<<egrid, 256, 0, st>>>(db->v, db->d_qc, db->d_qs, db->d_qe, (int)qa, (int)qb, nullptr,
                                                                           db->d_qoff, qoff[qa], zeroCopy ? hostDst : db->d_enumOut[b], idxBits);
            else
            igd_enum_queries<true><<<egrid, 256, 0, st>>>(db->v, db->d_qc, db->d_qs, db->d_qe, (int)qa, (int)qb, nullptr,
                                                          db->d_qoff, qoff[qa], zeroCopy ? hostDst : db->d_enumOut[b]);
            e = hipEventRecord(db->evFill[b], st);
            if (e == hipSuccess) e = hipStreamWaitEvent(db->copyStream, db->evFill[b], 0);
            if (e == hipSuccess && !zeroCopy)
                e = hipMemcpyAsync(hostDst, db->d_enumOut[b], (size_t)nh * hitBytes, hipMemcpyDeviceToHost, db->copyStream);
            if (e == hipSuccess) e = hipEventRecord(db->evCopy[b], db->copyStream);
            if (e != hipSuccess) break;
        }
        if (anySink && k >= 1 && prevB > prevA) {         // hand out the previous chunk while this one is produced
            if (qoff[prevB] > qoff[prevA]) e = hipEventSynchronize(db->evCopy[(k - 1) & 1]);
            if (e == hipSuccess) sinkRc = p8 ? sink8(ctx, prevA, prevB, qoff, (const igd_hip_hit8 *)db->h_enumPin[(k - 1) & 1], idxBits)
                                             : sink(ctx, prevA, prevB, qoff, db->h_enumPin[(k - 1) & 1]);
        }
        prevA = qa; prevB = qb;
        qa = qb;
        if (nh > 0 || anySink) k++;
    }
    {   // both streams are drained whatever happened: after a failed call an earlier chunk's copy may still be writing into
        // `whole` or a pinned chunk buffer, which are released / reused right below
        const hipError_t e1 = hipStreamSynchronize(db->copyStream), e2 = hipStreamSynchronize(st);
        if (e == hipSuccess) e = e1;
        if (e == hipSuccess) e = e2;
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess && anySink && sinkRc == 0 && prevB > prevA)
        sinkRc = p8 ? sink8(ctx, prevA, prevB, qoff, (const igd_hip_hit8 *)db->h_enumPin[(k - 1) & 1], idxBits)
                    : sink(ctx, prevA, prevB, qoff, db->h_enumPin[(k - 1) & 1]);
    ENUM_PHASE("fill + D2H (+ sink)");
#undef ENUM_PHASE
    if (e != hipSuccess) {
        if (whole) igd_hip_free(whole);
        set_err("enumerate fill", e, __FILE__, __LINE__);
        return IGD_HIP_ERR_DEVICE;
    }
    if (sinkRc != 0) {
        snprintf(g_err, sizeof g_err, "igd_hip_enumerate_stream: the sink stopped the enumeration (%d)", sinkRc);
        return IGD_HIP_ERR_ARG;
    }
    if (wholeOut) *wholeOut = whole;
    return IGD_HIP_OK;
}

extern "C" int igd_hip_enumerate(igd_hip_db *db, const int32_t *ichr, const int32_t *qs, const int32_t *qe,
                                 int64_t nq, int64_t *qoff, igd_hip_hit **out, int64_t *total)
{
    if (!db || !qoff || !out || nq < 0 || nq > max_batch() || (nq > 0 && (!ichr || !qs || !qe))) {
        snprintf(g_err, sizeof g_err, "igd_hip_enumerate: bad argument (batch limit %lld)", (long long)max_batch());
        return IGD_HIP_ERR_ARG;
    }
    *out = nullptr;
    if (total) *total = 0;
    for (int64_t i = 0; i <= nq; i++) qoff[i] = 0;
    if (nq == 0 || db->nT == 0) return IGD_HIP_OK;
    return enumerate_core(db, ichr, qs, qe, nq, qoff, true, out, nullptr, nullptr, total);
}

extern "C" int igd_hip_enumerate_stream(igd_hip_db *db, const int32_t *ichr, const int32_t *qs, const int32_t *qe,
                                        int64_t nq, int64_t *qoff, igd_hip_enum_sink sink, void *ctx, int64_t *total)
{
    if (!db || !qoff || !sink || nq < 0 || nq > max_batch() || (nq > 0 && (!ichr || !qs || !qe))) {
        snprintf(g_err, sizeof g_err, "igd_hip_enumerate_stream: bad argument (batch limit %lld)", (long long)max_batch());
        return IGD_HIP_ERR_ARG;
    }
    if (total) *total = 0;
    for (int64_t i = 0; i <= nq; i++) qoff[i] = 0;
    if (nq == 0) return IGD_HIP_OK;
    if (db->nT == 0) return sink(ctx, 0, nq, qoff, nullptr) == 0 ? IGD_HIP_OK : IGD_HIP_ERR_ARG;
    return enumerate_core(db, ichr, qs, qe, nq, qoff, false, nullptr, sink, ctx, total);
}

// Whether this database's records fit igd_hip_hit8, and how its second word is split (looked at once: one pass over the exact
// start / end arrays, 0.1 ms for 5 x 10^7 records).
extern "C" int igd_hip_hit8_idx_bits(igd_hip_db *db)
{
    if (!db) return -1;
    if (db->hit8State == 0) {
        if (hipSetDevice(db->device) != hipSuccess) return -1;
        int bits = 0;
        while (bits < 31 && (1ll << bits) < (long long)db->nFiles) bits++;
        unsigned int mx = 0, *d_mx = nullptr;
        bool ok = db->nFiles >= 1 && (1ll << bits) >= (long long)db->nFiles && hipMalloc((void **)&d_mx, 4) == hipSuccess;
        if (ok && db->nRec > 0) {
            ok = hipMemsetAsync(d_mx, 0, 4, db->stream) == hipSuccess;
            if (ok) {
                k_max_len<<<db->grid * 4, 256, 0, db->stream>>>(db->v.start, db->v.end, (int64_t)db->nRec, d_mx);
                ok = hipMemcpyAsync(&mx, d_mx, 4, hipMemcpyDeviceToHost, db->stream) == hipSuccess && hipStreamSynchronize(db->stream) == hipSuccess;
            }
        }
        if (d_mx) (void)hipFree(d_mx);
        if (!ok) return -1;                              // (not remembered: a transient failure)
        // the length field holds 32 - bits bits; end < start shows as a huge unsigned length
        db->hit8Bits = bits;
        db->hit8State = (bits >= 32 || (bits > 0 && ((unsigned long long)mx >> (32 - bits)) != 0ull)) ? 2 : 1;
    }
    return db->hit8State == 1 ? db->hit8Bits : -1;
}

extern "C" int igd_hip_enumerate_stream8(igd_hip_db *db, const int32_t *ichr, const int32_t *qs, const int32_t *qe,
                                         int64_t nq, int64_t *qoff, igd_hip_enum_sink8 sink, void *ctx, int64_t *total)
{
    if (!db || !qoff || !sink || nq < 0 || nq > max_batch() || (nq > 0 && (!ichr || !qs || !qe))) {
        snprintf(g_err, sizeof g_err, "igd_hip_enumerate_stream8: bad argument (batch limit %lld)", (long long)max_batch());
        return IGD_HIP_ERR_ARG;
    }
    const int bits = igd_hip_hit8_idx_bits(db);
    if (bits < 0) {
        snprintf(g_err, sizeof g_err, "igd_hip_enumerate_stream8: a record of this database does not fit 8 bytes (igd_hip_hit8_idx_bits() < 0): "
                                      "use igd_hip_enumerate_stream");
        return IGD_HIP_ERR_ARG;
    }
    if (total) *total = 0;
    for (int64_t i = 0; i <= nq; i++) qoff[i] = 0;
    if (nq == 0) return IGD_HIP_OK;
    if (db->nT == 0) return sink(ctx, 0, nq, qoff, nullptr, bits) == 0 ? IGD_HIP_OK : IGD_HIP_ERR_ARG;
    return enumerate_core(db, ichr, qs, qe, nq, qoff, false, nullptr, nullptr, ctx, total, sink);
}
