// engine/host_open.hpp -- part of igd_hip.hip (included there once; not a stand-alone header).
// handles: allocation, close, pinned buffers, re-tiled copy, igd_hip_open
// ==========================================================================================
// host side
static thread_local igd_hip_db *t_arenaOwner = nullptr;   // set while igd_hip_open builds the image

template <typename T>
static int dalloc(T **p, size_t n, int64_t *acct)
{
    *p = nullptr;
    if (n == 0) n = 1;
    if (t_arenaOwner && t_arenaOwner->arena) {
        igd_hip_db *o = t_arenaOwner;
        const size_t at = (o->arenaUsed + 255) & ~(size_t)255, bytes = n * sizeof(T);
        if (at + bytes <= o->arenaSize) {
            *p = (T *)(o->arena + at);
            o->arenaUsed = at + bytes;
            if (acct) *acct += (int64_t)bytes;
            return IGD_HIP_OK;
        }
    }
    hipError_t e = hipMalloc((void **)p, n * sizeof(T));
    if (e != hipSuccess) {
        set_err("hipMalloc", e, __FILE__, __LINE__);
        return IGD_HIP_ERR_NOMEM;
    }
    if (acct) *acct += (int64_t)(n * sizeof(T));
    return IGD_HIP_OK;
}

extern "C" void igd_hip_close(igd_hip_db *db)
{
    if (!db) return;
    if (t_arenaOwner == db) t_arenaOwner = nullptr;
    if (db->inner) { igd_hip_close(db->inner); db->inner = nullptr; }
    (void)hipSetDevice(db->device);
    if (db->d_rEmpty) (void)hipFree(db->d_rEmpty);
    void *ptrs[] = {db->d_start, db->d_end, db->d_idx, db->d_value, db->d_tileOff, db->d_tileCnt,
                    db->d_tileBd, db->d_ctgBase, db->d_ctgNTile, db->d_tileUnit0, db->d_heavy, db->d_far, db->d_tileD, db->d_tileBits,
                    db->d_pairCnt, db->d_pairPos, db->d_blockSums, db->d_pairs, db->d_long, db->d_fix, db->d_ctl,
                    db->d_units, db->d_firstQ, db->d_pairN, db->d_pse, db->d_px, db->d_pxv,
                    db->d_slab, db->d_qc, db->d_qs, db->d_qe, db->d_hits, db->d_total, db->d_qw, db->d_later, db->d_spill, db->d_laterHdr, db->d_lpos, db->d_cov,
                    db->d_spTable, db->d_spT, db->d_runIchr, db->d_spSub};
    for (void *p : ptrs)
        if (p && !(db->arena && (char *)p >= db->arena && (char *)p < db->arena + db->arenaSize)) (void)hipFree(p);
    if (db->arena) (void)hipFree(db->arena);
    {
        void *es[] = {db->d_qcount, db->d_qoff, db->d_enumBsum, db->d_enumOut[0], db->d_enumOut[1]};
        for (void *p : es) if (p) (void)hipFree(p);
        for (int k = 0; k < 2; k++) {
            if (db->h_enumPin[k]) (void)hipHostFree(db->h_enumPin[k]);
            if (db->evFill[k]) (void)hipEventDestroy(db->evFill[k]);
            if (db->evCopy[k]) (void)hipEventDestroy(db->evCopy[k]);
        }
        if (db->copyStream) (void)hipStreamDestroy(db->copyStream);
    }
    for (hipEvent_t e : db->ev) (void)hipEventDestroy(e);
    if (db->stream) (void)hipStreamDestroy(db->stream);
    delete db;
}

extern "C" int igd_hip_device(const igd_hip_db *db) { return db ? db->device : -1; }
extern "C" int32_t igd_hip_nfiles(const igd_hip_db *db) { return db ? db->nFiles : 0; }
extern "C" int64_t igd_hip_resident_bytes(const igd_hip_db *db) { return db ? db->resident + (db->inner ? db->inner->resident : 0) : 0; }
// Enumeration results are returned in PINNED host memory (the D2H copy of ~16 bytes per overlap is
// the slowest step of `-f`; pageable memory runs it at a fifth of the PCIe rate).  Pinning is
// expensive, so one released buffer is kept for the next call.
static void *g_pinCache = nullptr;
static size_t g_pinCacheBytes = 0;
static std::mutex g_pinLock;                              // engines of several devices run on threads of one process (igdc_search_multi)
static void *pinned_take(size_t bytes, size_t *got)
{
    {
        std::lock_guard<std::mutex> lk(g_pinLock);
        if (g_pinCache && g_pinCacheBytes >= bytes) {
            void *p = g_pinCache;
            *got = g_pinCacheBytes;
            g_pinCache = nullptr; g_pinCacheBytes = 0;
            return p;
        }
    }
    void *p = nullptr;
    size_t want = bytes + bytes / 8 + 4096;
    if (hipHostMalloc(&p, want + 64, hipHostMallocDefault) != hipSuccess) return nullptr;
    *got = want;
    return p;
}
extern "C" void igd_hip_free(void *p)
{
    if (!p) return;
    size_t *hdr = (size_t *)((char *)p - 64);            // size header in front of the payload
    void *drop = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_pinLock);
        if (g_pinCache && g_pinCacheBytes >= hdr[0]) drop = hdr;
        else { drop = g_pinCache; g_pinCache = hdr; g_pinCacheBytes = hdr[0]; }
    }
    if (drop) (void)hipHostFree(drop);
}
extern "C" const char *igd_hip_scan_kernel_name(void) { return "igd_scan_sorted"; }

// which scan kernel the last batch of `db` ran on (waits for it: the device decides for an IGD_HIP_FLAG default batch)
extern "C" const char *igd_hip_last_scan_kernel(igd_hip_db *db)
{
    if (db && db->inner) return igd_hip_last_scan_kernel(db->inner);
    if (!db || db->epoch == 0) return "";
    if (db->lastMode == 2) return "igd_scan_tiles";
    if (db->lastDirect) return "igd_scan_direct";
    int32_t uns = 0;
    if (hipSetDevice(db->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(&uns, db->d_ctl + CTL_UNSORTED, 4, hipMemcpyDeviceToHost) != hipSuccess) return "";
    if (uns == db->epoch) return "igd_scan_tiles";       // found unordered: the bucket path's kernel
    return db->lastPacked ? "igd_scan_sorted" : "igd_scan_tiles";
}

static double wall_s(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
#define OPEN_PHASE(name) do { if (tim) { double t_ = wall_s(); fprintf(stderr, "[igd timing]   open: %-22s %8.1f ms\n", name, 1e3 * (t_ - t0)); t0 = t_; } } while (0)

// see igd_hip_open: the records of `db` once each, bucketed again in tiles of 2^14 bp, as a database of its own
static int build_retiled(igd_hip_db *db, const igd_hip_desc *d, int realShift, const std::vector<Unit> &units, int device)
{
    const int64_t n = db->nRec;
    std::vector<int32_t> st((size_t)n), en((size_t)n), ix((size_t)n), va;
    HIPCHK(hipMemcpy(st.data(), db->d_start, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(en.data(), db->d_end, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ix.data(), db->d_idx, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (db->gType == 1) { va.resize((size_t)n); HIPCHK(hipMemcpy(va.data(), db->d_value, (size_t)n * 4, hipMemcpyDeviceToHost)); }
    // every record ONCE: of its copies (one per tile it reaches into) the one in the tile it starts in
    std::vector<int32_t> uc, us, ue, uv, uf;
    uc.reserve((size_t)n); us.reserve((size_t)n); ue.reserve((size_t)n); uf.reserve((size_t)n);
    if (db->gType == 1) uv.reserve((size_t)n);
    const int32_t W = d->nbp;
    int64_t t = 0, r = 0;
    for (int32_t c = 0; c < d->nCtg; c++)
        for (int32_t j = 0; j < d->nTile[c]; j++, t++) {
            const int64_t T0 = (int64_t)j * W;
            for (int32_t k = 0; k < d->nCnt[t]; k++, r++) {
                // (a file `create` did not write -- a record outside its tile, empty or starting before the contig -- keeps its own
                // tiles: what the reference makes of such a record depends on where it was put)
                if (!((int64_t)st[(size_t)r] < T0 + W && (int64_t)en[(size_t)r] > T0) || st[(size_t)r] < 0 || st[(size_t)r] >= en[(size_t)r]) {
                    snprintf(g_err, sizeof g_err, "a record that `igd create` would not have written (contig %d, tile %d)", c, j);
                    return IGD_HIP_ERR_ARG;
                }
                if ((int64_t)st[(size_t)r] < T0) continue;              // begins in an earlier tile: counted there
                uc.push_back(c); us.push_back(st[(size_t)r]); ue.push_back(en[(size_t)r]); uf.push_back(ix[(size_t)r]);
                if (db->gType == 1) uv.push_back(va[(size_t)r]);
            }
        }
    (void)units;
    igd_hip_create_desc cd;
    memset(&cd, 0, sizeof cd);
    cd.nbp = 1 << 14; cd.gType = db->gType; cd.nCtg = d->nCtg; cd.n = (int64_t)us.size();
    cd.ctg = uc.data(); cd.start = us.data(); cd.end = ue.data(); cd.value = db->gType == 1 ? uv.data() : nullptr; cd.file = uf.data();
    cd.ctgName = nullptr; cd.out_fd = -1;
    igd_hip_created made;
    memset(&made, 0, sizeof made);
    int rc = igd_hip_create(&cd, device, &made);
    if (rc != IGD_HIP_OK) return rc;
    igd_hip_desc vd;
    memset(&vd, 0, sizeof vd);
    vd.nbp = 1 << 14; vd.gType = db->gType; vd.nCtg = d->nCtg; vd.nFiles = d->nFiles;
    // (a contig without records has no tile in what `create` returns; the loaders want one)
    std::vector<int32_t> vnT((size_t)d->nCtg), vCnt;
    {
        int64_t at = 0;
        for (int32_t c = 0; c < d->nCtg; c++) {
            const int32_t k = made.nTile[c];
            vnT[(size_t)c] = k > 0 ? k : 1;
            if (k > 0) { vCnt.insert(vCnt.end(), made.nCnt + at, made.nCnt + at + k); at += k; }
            else vCnt.push_back(0);
        }
    }
    vd.nTile = vnT.data(); vd.nCnt = vCnt.data(); vd.records = made.records; vd.nRecords = made.nRecords; vd.fd = -1;
    igd_hip_db *inner = nullptr;
    rc = igd_hip_open(&vd, device, &inner);
    igd_hip_created_free(&made);
    if (rc != IGD_HIP_OK) return rc;
    HIPCHK(hipSetDevice(db->device));
    // the file's tiles: which of them are empty (rule NEST, :468)
    std::vector<uint32_t> bits((size_t)((db->nT + 31) / 32) + 1, 0u);
    for (int64_t g = 0; g < db->nT; g++) if (d->nCnt[g] == 0) bits[(size_t)(g >> 5)] |= 1u << (g & 31);
    if ((rc = dalloc(&db->d_rEmpty, bits.size(), nullptr)) != IGD_HIP_OK) { igd_hip_close(inner); return rc; }
    HIPCHK(hipMemcpy(db->d_rEmpty, bits.data(), bits.size() * 4, hipMemcpyHostToDevice));
    inner->v.vshift = realShift;
    inner->v.rNTile = db->d_ctgNTile; inner->v.rBase = db->d_ctgBase; inner->v.rEmpty = db->d_rEmpty;
    db->inner = inner;
    return IGD_HIP_OK;
}

extern "C" int igd_hip_open(const igd_hip_desc *d, int device, igd_hip_db **out)
{
    const char *tenv = getenv("IGD_TIMING");
    const bool tim = tenv && *tenv && *tenv != '0';
    double t0 = wall_s();
    if (!d || !out || d->nbp <= 0 || d->nCtg < 0 || d->nFiles < 0 || d->nRecords < 0 ||
        (d->gType != 0 && d->gType != 1) || (d->nCtg > 0 && (!d->nTile || !d->nCnt)) ||
        (d->nRecords > 0 && !d->records && d->fd < 0)) {
        snprintf(g_err, sizeof g_err, "igd_hip_open: bad descriptor");
        return IGD_HIP_ERR_ARG;
    }
    if (igd_hip_build_wrong_counts()) {
        const char *ok = getenv("IGD_HIP_ALLOW_EXP_BUILD");
        if (!ok || ok[0] != '1') {
            snprintf(g_err, sizeof g_err, "igd_hip_open: this libigd_hip.so is a measurement build (IGD_EXP=0x%x) that gives WRONG counts; "
                     "set IGD_HIP_ALLOW_EXP_BUILD=1 to use it anyway", igd_hip_build_flags());
            return IGD_HIP_ERR_ARG;
        }
        fprintf(stderr, "igd_hip: WARNING: measurement build IGD_EXP=0x%x -- counts are WRONG on purpose\n", igd_hip_build_flags());
    }
    // Bringing up the HIP runtime is the largest single item of a command line search (50 .. 120 ms on the pool's hosts, more on
    // some): it runs on a thread of its own while this one builds the host-side tables (24 ms for the roadmap-scale header).
    // (Error texts are thread-local: a failure is looked at again from this thread below.)
    std::thread warm;
    try {
        warm = std::thread([device]() {
            int n = 0;
            if (hipGetDeviceCount(&n) == hipSuccess && device >= 0 && device < n && hipSetDevice(device) == hipSuccess) (void)hipFree(nullptr);
        });
    } catch (...) { }
    struct JoinWarm { std::thread &t; ~JoinWarm() { if (t.joinable()) t.join(); } } joinWarm{warm};
    igd_hip_db *db = new igd_hip_db();   // value-initialised: every field zero
    db->device = device;
    {   // the environment is read here, once: the per-batch entry points look at nothing but the handle
        const char *fr = getenv("IGD_HIP_RANK");
        db->forceRank = fr && *fr ? atoi(fr) : -1;
        const char *fb = getenv("IGD_HIP_BIG");
        db->bigImage = fb && *fb == '1';                 // (|| the record count, once it is known)
        db->qbVec1 = getenv("IGD_HIP_QB_VEC1") != nullptr;
        const char *fd = getenv("IGD_HIP_DIRECT");
        db->forceDirect = fd && *fd ? atoi(fd) : -1;
        db->timing = tim;
    }
    db->nbp = d->nbp; db->gType = d->gType; db->nCtg = d->nCtg; db->nFiles = d->nFiles;
    db->nRec = d->nRecords;

    // host-side tables
    int64_t nT = 0;
    for (int c = 0; c < d->nCtg; c++) nT += d->nTile[c];
    bool jfits = true;
    for (int c = 0; c < d->nCtg; c++) jfits = jfits && d->nTile[c] < (1 << 27);
    // the merge join packs (global tile number << 4 | span) into one int32 per query (k_query_bounds):
    // the TOTAL number of tiles has to stay below 2^27, not just every contig's
    if (nT >= (1 << 27) || !jfits) {
        snprintf(g_err, sizeof g_err, "igd_hip_open: too many tiles (%lld; the engine's limit is 2^27-1)", (long long)nT);
        delete db;
        return IGD_HIP_ERR_ARG;
    }
    db->nT = (int32_t)nT;
    std::vector<int64_t> tileOff((size_t)nT + 1);
    std::vector<int32_t> tileCnt((size_t)nT + 1), tileBd((size_t)nT + 1), ctgBase((size_t)d->nCtg + 1),
        ctgNTile((size_t)d->nCtg + 1), tileUnit0((size_t)nT + 1);
    std::vector<Unit> units;
    int64_t off = 0;
    int32_t maxIdxCheck = 0;
    (void)maxIdxCheck;
    {
        int64_t t = 0;
        for (int c = 0; c < d->nCtg; c++) {
            ctgBase[c] = (int32_t)t;
            ctgNTile[c] = d->nTile[c];
            for (int j = 0; j < d->nTile[c]; j++, t++) {
                int32_t cnt = d->nCnt[t];
                if (cnt < 0) cnt = 0;
                if (cnt > db->maxTileCnt) db->maxTileCnt = cnt;
                tileOff[t] = off;
                tileCnt[t] = cnt;
                tileUnit0[t] = (int32_t)units.size();
                // tile start coordinate; computed with wrap like `bd` at src/igd_search.c:496,529
                tileBd[t] = (j == 0) ? INT_MIN : (int32_t)((uint32_t)d->nbp * (uint32_t)j);
                for (int32_t r0 = 0; r0 < cnt || r0 == 0; r0 += IGD_CHUNK) {
                    Unit u;
                    u.off = off + r0;
                    u.tile = (int32_t)t;
                    u.n = cnt - r0 < IGD_CHUNK ? cnt - r0 : IGD_CHUNK;
                    for (int r = 0; r < 6; r++) u.W[r] = 0;
                    u.pre = 0;
                    int fl = r0 == 0 ? 1 : 0;
                    for (int k = 1; k < IGD_SHORT_TILES && k <= j; k++)
                        if (d->nCnt[t - k] <= 0) fl |= 1 << k;
                    u.jf = (j << 4) | fl;
                    units.push_back(u);
                }
                off += cnt;
            }
        }
        tileOff[nT] = off;
        tileUnit0[nT] = (int32_t)units.size();
    }
    if (off != d->nRecords) {
        snprintf(g_err, sizeof g_err, "igd_hip_open: nRecords %lld != sum(nCnt) %lld",
                 (long long)d->nRecords, (long long)off);
        delete db;
        return IGD_HIP_ERR_ARG;
    }
    db->nUnits = (int32_t)units.size();
    OPEN_PHASE("host tables");
    if (warm.joinable()) warm.join();
    {
        const int ndev = igd_hip_device_count();
        if (ndev <= 0) {
            if (!g_err[0]) snprintf(g_err, sizeof g_err, "igd_hip_open: no HIP device");
            delete db;
            return IGD_HIP_ERR_DEVICE;
        }
        if (device < 0 || device >= ndev) {
            snprintf(g_err, sizeof g_err, "igd_hip_open: device %d out of range (%d visible)", device, ndev);
            delete db;
            return IGD_HIP_ERR_ARG;
        }
        const hipError_t e_ = hipSetDevice(device);
        if (e_ != hipSuccess) { set_err("hipSetDevice", e_, __FILE__, __LINE__); delete db; return IGD_HIP_ERR_DEVICE; }
    }
    OPEN_PHASE("HIP runtime init (beside the tables)");

    int rc;
    int64_t *acct = &db->resident;
#define TRY(x) do { rc = (x); if (rc != IGD_HIP_OK) { igd_hip_close(db); return rc; } } while (0)
#define TRYHIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_err(#x, e_, __FILE__, __LINE__); igd_hip_close(db); return IGD_HIP_ERR_DEVICE; } } while (0)
    TRYHIP(hipStreamCreateWithFlags(&db->stream, hipStreamNonBlocking));
    OPEN_PHASE("stream");
    size_t n = (size_t)d->nRecords;
    {   // launch geometry first: the slab is part of the arena
        int cus = 0;                                     // one attribute, not hipGetDeviceProperties (~30 ms)
        TRYHIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
        if (cus <= 0) cus = 256;
        // more files than LDS counters (15 360): the batch is scanned once per WINDOW of files (up to IGD_MAX_WINDOWS passes:
        // 163 840 files; beyond that per-record global atomics)
        db->winN = d->nFiles; db->nWin = 1;
        if ((size_t)d->nFiles * 8 > IGD_LDS_HITS_MAX_BYTES) {
            // (windows of at most 10 240 files: two workgroups of the merge join's full build per CU then still have LDS arrays
            // of 512 query starts -- with 15 360 counters they had none, and a dense batch took 631 instead of 150 us per pass)
            const int cap = IGD_WINDOW_FILES;
            db->nWin = (d->nFiles + cap - 1) / cap;
            db->winN = ((d->nFiles + db->nWin - 1) / db->nWin + 31) & ~31;      // windows of equal size (the last one may be shorter)
            if (db->winN > cap) db->winN = cap;
            db->nWin = (d->nFiles + db->winN - 1) / db->winN;
            // (no window builds for images addressed with 64-bit unit bases)
            if (db->nWin > IGD_MAX_WINDOWS || getenv("IGD_HIP_NO_WINDOWS") || db->bigImage || (int64_t)n + IGD_CHUNK >= (1ll << 30)) { db->winN = d->nFiles; db->nWin = 1; }
        }
        db->ldsBytes = (int)((size_t)db->winN * 8);
        db->ldsHits = db->ldsBytes <= IGD_LDS_HITS_MAX_BYTES;
        {   // igd_scan_sorted: counters + per wave (sorted starts, histogram, the tile's query starts).  The last array takes
            // what two workgroups per CU leave of the 160 KiB: tiles with more queries bisect the caller's array instead
            const int hitB = db->ldsHits ? (int)((((size_t)db->winN * 4) + 15) & ~(size_t)15) : 0;   // (32-bit counters: CNT32)
            constexpr int waveLds = IGD_D_WLDS > IGD_WLDS_BYTES ? IGD_D_WLDS : IGD_WLDS_BYTES;        // (igd_scan_direct's areas are the larger ones)
            int spare = (160 * 1024 / 2 - 512 - hitB) / (IGD_WG_RANK / IGD_WAVE) - waveLds;   // the full build: 2 workgroups per CU
            db->sbCap = 0;                               // a power of two (s_compute pads the array to one)
            for (int c = 64; c <= 2048 && c <= spare / 2; c <<= 1) db->sbCap = c;
            if (db->sbCap == 0) {                        // counters that leave two workgroups per CU nothing: one workgroup, with the arrays
                spare = (160 * 1024 - 512 - hitB) / (IGD_WG_RANK / IGD_WAVE) - waveLds;
                for (int c = 64; c <= 2048 && c <= spare / 2; c <<= 1) db->sbCap = c;
            }
            db->ldsSorted = hitB + (IGD_WG_RANK / IGD_WAVE) * (IGD_WLDS_BYTES + 2 * db->sbCap);
            db->ldsDirect = hitB + (IGD_WG_DIR / IGD_WAVE) * (IGD_D_WLDS + 2 * db->sbCap);       // igd_scan_direct: the same, its waves' areas a little larger
        }
        int perCU = (IGD_WPE * 256) / IGD_WG;             // IGD_WPE waves per SIMD = 4 * IGD_WPE per CU
        if (getenv("IGD_HIP_WG_PER_CU")) perCU = atoi(getenv("IGD_HIP_WG_PER_CU"));
        if (db->ldsHits && db->ldsBytes > 0) {
            int fit = (160 * 1024) / (db->ldsSorted + 256);
            if (fit < 1) fit = 1;
            if (fit < perCU) perCU = fit;
        }
        if (perCU < 1) perCU = 1;
        db->grid = cus * perCU;
        const size_t slabB = db->ldsHits ? (size_t)db->grid * (size_t)(db->winN > 0 ? db->winN : 1) * 8 : 0;
        const bool willPack = d->nbp <= 32768 && d->nFiles <= 65536 && n > 0;
        size_t total = 16 * n + (willPack ? 10 * (n + IGD_CHUNK) : 0) + 128 * ((size_t)nT + 2) + (sizeof(Unit) + 4) * (units.size() + 1) + 4 * (IGD_HEAVY_MAX + IGD_HEAVYS_MAX) +
                       slabB + 8 * ((size_t)d->nFiles + 8) + 64 * 1024;
        db->arena = nullptr;
        if (hipMalloc((void **)&db->arena, total) == hipSuccess) { db->arenaSize = total; db->arenaUsed = 0; }
        else db->arena = nullptr;                        // fall back to individual allocations
        t_arenaOwner = db;
    }
    OPEN_PHASE("device props, arena");
    TRY(dalloc(&db->d_start, n, acct));
    TRY(dalloc(&db->d_end, n, acct));
    TRY(dalloc(&db->d_idx, n, acct));
    if (d->gType == 1) TRY(dalloc(&db->d_value, n, acct));
    TRY(dalloc(&db->d_tileOff, (size_t)nT + 1, acct));
    TRY(dalloc(&db->d_tileCnt, (size_t)nT + 1, acct));
    TRY(dalloc(&db->d_tileBd, (size_t)nT + 1, acct));
    TRY(dalloc(&db->d_ctgBase, (size_t)d->nCtg + 1, acct));
    TRY(dalloc(&db->d_ctgNTile, (size_t)d->nCtg + 1, acct));
    TRY(dalloc(&db->d_tileUnit0, (size_t)nT + 1, acct));
    TRY(dalloc(&db->d_heavy, IGD_HEAVY_MAX + IGD_HEAVYS_MAX, acct));   // bucket path's list, merge join's list
    TRY(dalloc(&db->d_units, units.size(), acct));
    TRY(dalloc(&db->d_far, units.size() + 1, acct));
    TRY(dalloc(&db->d_tileD, (size_t)nT + 1, acct));
    TRY(dalloc(&db->d_tileBits, ((size_t)nT + 31) / 32 + 1, acct));
    TRY(dalloc(&db->d_firstQ, (size_t)nT + 2, acct));
    TRY(dalloc(&db->d_lpos, (size_t)nT + 2 + IGD_SHORT_TILES, acct));
    TRY(dalloc(&db->d_pairN, (size_t)nT + 1, acct));
    TRY(dalloc(&db->d_spill, (size_t)nT + 2, acct));
    TRY(dalloc(&db->d_cov, 4 * IGD_COV_LEN(nT), acct));
    TRY(dalloc(&db->d_pairCnt, (size_t)nT + 1, acct));
    TRY(dalloc(&db->d_pairPos, (size_t)nT + 1, acct));
    TRY(dalloc(&db->d_blockSums, (size_t)(nT / IGD_SCAN_TILE + 2), acct));
    TRY(dalloc(&db->d_ctl, IGD_CTL_WORDS, acct));
    TRY(dalloc(&db->d_hits, (size_t)d->nFiles + 1, acct));
    TRY(dalloc(&db->d_total, 4, acct));
    TRYHIP(hipMemcpy(db->d_tileOff, tileOff.data(), ((size_t)nT + 1) * 8, hipMemcpyHostToDevice));
    OPEN_PHASE("first H2D copy");
    TRYHIP(hipMemcpy(db->d_tileCnt, tileCnt.data(), ((size_t)nT + 1) * 4, hipMemcpyHostToDevice));
    {
        std::vector<uint32_t> bits(((size_t)nT + 31) / 32 + 1, 0u);
        for (int64_t t = 0; t < nT; t++) if (tileCnt[(size_t)t] > 0) bits[(size_t)(t >> 5)] |= 1u << (t & 31);
        TRYHIP(hipMemcpy(db->d_tileBits, bits.data(), bits.size() * 4, hipMemcpyHostToDevice));
    }
    TRYHIP(hipMemcpy(db->d_tileBd, tileBd.data(), ((size_t)nT + 1) * 4, hipMemcpyHostToDevice));
    TRYHIP(hipMemcpy(db->d_ctgBase, ctgBase.data(), ((size_t)d->nCtg + 1) * 4, hipMemcpyHostToDevice));
    TRYHIP(hipMemcpy(db->d_ctgNTile, ctgNTile.data(), ((size_t)d->nCtg + 1) * 4, hipMemcpyHostToDevice));
    TRYHIP(hipMemcpy(db->d_tileUnit0, tileUnit0.data(), ((size_t)nT + 1) * 4, hipMemcpyHostToDevice));
    if (!units.empty())
        TRYHIP(hipMemcpy(db->d_units, units.data(), units.size() * sizeof(Unit), hipMemcpyHostToDevice));
    OPEN_PHASE("other table copies");
    // records: the AoS region goes through two pinned staging buffers -- the CPU fills one
    // (memcpy from the caller's memory, or pread from the .igd when desc->fd is used) while the
    // previous one is copied to the GPU and transposed there (SoA) on the engine's stream.
    if (n > 0) {
        const size_t recBytes = d->gType == 1 ? 16 : 12;
        const size_t slice = (size_t)1 << 20;            // records per stage (16 MiB of gdata_t)
        int nthr = (int)std::thread::hardware_concurrency();
        nthr = nthr < 1 ? 1 : (nthr > 8 ? 8 : nthr);
        const size_t sl = n < slice ? n : slice;
        void *d_aos[2] = {nullptr, nullptr}, *h_pin[2] = {nullptr, nullptr};
        hipEvent_t ev[2] = {nullptr, nullptr};
        hipError_t e = hipSuccess;
        for (int k = 0; k < 2 && e == hipSuccess; k++) {
            e = hipMalloc(&d_aos[k], sl * recBytes);
            if (e == hipSuccess) e = hipHostMalloc(&h_pin[k], sl * recBytes, hipHostMallocDefault);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming);
        }
        bool ioerr = false;
        int k = 0;
        OPEN_PHASE("staging buffers");
        for (size_t r0 = 0; r0 < n && e == hipSuccess && !ioerr; r0 += slice, k ^= 1) {
            const size_t m = n - r0 < slice ? n - r0 : slice;
            if (r0 >= 2 * slice) e = hipEventSynchronize(ev[k]);    // this stage's previous copy is done
            if (e != hipSuccess) break;
            {   // fill the stage with a few threads: one core copies page cache at only ~4 GB/s
                const size_t bytes = m * recBytes;
                const size_t part = ((bytes / nthr) + 4095) & ~(size_t)4095;
                std::vector<std::thread> th;
                std::vector<int> bad((size_t)nthr, 0);
                for (int t = 0; t < nthr; t++) {
                    const size_t b0 = (size_t)t * part;
                    if (b0 >= bytes) break;
                    const size_t b1 = b0 + part < bytes ? b0 + part : bytes;
                    char *dst = (char *)h_pin[k];
                    auto job = [=, &bad]() {
                        if (d->records) { memcpy(dst + b0, (const char *)d->records + r0 * recBytes + b0, b1 - b0); return; }
                        size_t done = b0;
                        while (done < b1) {
                            ssize_t got = pread(d->fd, dst + done, b1 - done, (off_t)(d->fd_offset + (int64_t)(r0 * recBytes + done)));
                            if (got <= 0) { bad[(size_t)t] = 1; return; }
                            done += (size_t)got;
                        }
                    };
                    if (t + 1 < nthr && b1 < bytes) th.emplace_back(job); else job();
                }
                for (auto &x : th) x.join();
                for (int b : bad) ioerr = ioerr || b;
                if (ioerr) break;
            }
            e = hipMemcpyAsync(d_aos[k], h_pin[k], m * recBytes, hipMemcpyHostToDevice, db->stream);
            if (e != hipSuccess) break;
            int blocks = (int)((m + 255) / 256);
            if (blocks > 256 * 32) blocks = 256 * 32;
            if (d->gType == 1)
                k_aos_to_soa16<<<blocks, 256, 0, db->stream>>>((const int4 *)d_aos[k], (int64_t)m,
                    db->d_start + r0, db->d_end + r0, db->d_idx + r0, db->d_value + r0);
            else
                k_aos_to_soa12<<<blocks, 256, 0, db->stream>>>((const int32_t *)d_aos[k], (int64_t)m,
                    db->d_start + r0, db->d_end + r0, db->d_idx + r0);
            e = hipEventRecord(ev[k], db->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(db->stream);
        OPEN_PHASE("read + upload + SoA");
        for (int q = 0; q < 2; q++) {
            if (d_aos[q]) (void)hipFree(d_aos[q]);
            if (h_pin[q]) (void)hipHostFree(h_pin[q]);
            if (ev[q]) (void)hipEventDestroy(ev[q]);
        }
        if (e != hipSuccess || ioerr) {
            if (ioerr) snprintf(g_err, sizeof g_err, "igd_hip_open: short read of the tile region");
            else set_err("upload/transpose", e, __FILE__, __LINE__);
            igd_hip_close(db);
            return ioerr ? IGD_HIP_ERR_ARG : IGD_HIP_ERR_DEVICE;
        }
        TRYHIP(hipMemset(db->d_ctl, 0, IGD_CTL_WORDS * 4));
        k_idx_range<<<256 * 8, 256, 0, db->stream>>>(db->d_idx, (int64_t)n, d->nFiles, db->d_ctl);
        int32_t bad = 0;
        TRYHIP(hipStreamSynchronize(db->stream));
        TRYHIP(hipMemcpy(&bad, db->d_ctl, 4, hipMemcpyDeviceToHost));
        if (bad) {
            snprintf(g_err, sizeof g_err, "igd_hip_open: a record's dataset index is outside [0,%d) "
                     "(the _index.tsv does not match the .igd)", d->nFiles);
            igd_hip_close(db);
            return IGD_HIP_ERR_ARG;
        }
    }
    TRYHIP(hipMemset(db->d_pairCnt, 0, ((size_t)nT + 1) * 4));
    TRYHIP(hipMemset(db->d_spill, 0, ((size_t)nT + 2) * 4));
    TRYHIP(hipMemset(db->d_cov, 0, 4 * IGD_COV_LEN(nT) * 4));
    TRYHIP(hipMemset(db->d_ctl, 0, IGD_CTL_WORDS * 4));

    // launch geometry of the scan kernel: computed above (db->grid, db->ldsBytes, db->ldsHits)
    if (db->ldsHits) {
        TRY(dalloc(&db->d_slab, (size_t)db->grid * (size_t)(db->winN > 0 ? db->winN : 1), acct));
        if (db->ldsBytes > 64 * 1024) {
            const void *fns[] = {(const void *)igd_scan_tiles<true, false, true, false>, (const void *)igd_scan_tiles<true, true, true, false>,
                                 (const void *)igd_scan_tiles<false, false, true, false>, (const void *)igd_scan_tiles<false, true, true, false>,
                                 (const void *)igd_scan_tiles<true, false, true, true>, (const void *)igd_scan_tiles<true, true, true, true>,
                                 (const void *)igd_scan_tiles<false, false, true, true>, (const void *)igd_scan_tiles<false, true, true, true>};
            for (const void *fn : fns)
                TRYHIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, db->ldsBytes));
            if (db->nWin > 1) {
                const void *wfn[] = {(const void *)igd_scan_tiles<true, false, true, false, true>, (const void *)igd_scan_tiles<true, true, true, false, true>,
                                     (const void *)igd_scan_tiles<false, false, true, false, true>, (const void *)igd_scan_tiles<false, true, true, false, true>,
                                     (const void *)igd_scan_tiles<true, false, true, true, true>, (const void *)igd_scan_tiles<true, true, true, true, true>,
                                     (const void *)igd_scan_tiles<false, false, true, true, true>, (const void *)igd_scan_tiles<false, true, true, true, true>};
                for (const void *fn : wfn)
                    TRYHIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, db->ldsBytes));
            }
        }
    }
    if (db->ldsSorted > 64 * 1024) {
        // every instantiation launch_scan can pick: <USE_V, LDS_HITS, CNT32 = LDS_HITS, BIG, RANK> -- without LDS counters the
        // waves' rank-method areas alone are 75 KiB
#define IGD_SORTED_FNS(V, LH) (const void *)igd_scan_sorted<V, LH, LH, true, true>, (const void *)igd_scan_sorted<V, LH, LH, false, false>, \
                              (const void *)igd_scan_sorted<V, LH, LH, false, true>
        const void *sfn[] = {IGD_SORTED_FNS(false, true), IGD_SORTED_FNS(true, true), IGD_SORTED_FNS(false, false), IGD_SORTED_FNS(true, false)};
#undef IGD_SORTED_FNS
        for (const void *fn : sfn) TRYHIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, db->ldsSorted));
        if (db->nFiles <= 8) {       // FEW = 1 (one file) / 2 (up to eight): launch_scan's builds for databases of very few files
            const void *ffn[] = {(const void *)igd_scan_sorted<false, true, true, false, false, 1>, (const void *)igd_scan_sorted<false, true, true, false, true, 1>,
                                 (const void *)igd_scan_sorted<true, true, true, false, false, 1>, (const void *)igd_scan_sorted<true, true, true, false, true, 1>,
                                 (const void *)igd_scan_sorted<false, true, true, false, false, 2>, (const void *)igd_scan_sorted<false, true, true, false, true, 2>,
                                 (const void *)igd_scan_sorted<true, true, true, false, false, 2>, (const void *)igd_scan_sorted<true, true, true, false, true, 2>};
            for (const void *fn : ffn) TRYHIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, db->ldsSorted));
        }
        if (db->nWin > 1) {
            const void *wfn[] = {(const void *)igd_scan_sorted<false, true, true, false, false, 3>, (const void *)igd_scan_sorted<false, true, true, false, true, 3>,
                                 (const void *)igd_scan_sorted<true, true, true, false, false, 3>, (const void *)igd_scan_sorted<true, true, true, false, true, 3>};
            for (const void *fn : wfn) TRYHIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, db->ldsSorted));
        }
    }
    if (db->ldsDirect > 64 * 1024) {
        const void *dfn[] = {(const void *)igd_scan_direct<false>, (const void *)igd_scan_direct<true>};
        for (const void *fn : dfn) TRYHIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, db->ldsDirect));
    }
    {   // the batch's last launch: its workgroups of 16 waves carry 16 rank-method areas (the skew valves) and the 64-bit counters
        // of the long queries' work (up to 48 KiB): beyond the 64 KiB a kernel gets without asking
        const void *tfn[] = {(const void *)k_reduce_slabs<false>, (const void *)k_reduce_slabs<true>};
        for (const void *fn : tfn) TRYHIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    }
#undef TRY
#undef TRYHIP
    DbView &v = db->v;
    v.nbp = db->nbp; v.nCtg = db->nCtg; v.nT = db->nT; v.nFiles = db->nFiles;
    v.vshift = -1; v.rNTile = nullptr; v.rBase = nullptr; v.rEmpty = nullptr;
    v.shift = -1;
    for (int b = 0; b < 31; b++)
        if (db->nbp == (1 << b)) v.shift = b;
    v.units = db->d_units; v.nUnits = db->nUnits;

    v.start = db->d_start; v.end = db->d_end; v.idx = db->d_idx; v.value = db->d_value;
    v.tileOff = db->d_tileOff; v.tileCnt = db->d_tileCnt; v.tileBd = db->d_tileBd; v.tileBits = db->d_tileBits;
    v.ctgBase = db->d_ctgBase; v.ctgNTile = db->d_ctgNTile; v.tileUnit0 = db->d_tileUnit0; v.cov = db->d_cov;
    OPEN_PHASE("idx check, slab");
    // compact image (see k_pack_units): needs tile-relative offsets and idx to fit 16 bits
    db->packed = db->nbp <= 32768 && db->nFiles <= 65536 && db->nRec > 0 && !getenv("IGD_HIP_NO_PACK");
    if (db->packed) {
        const size_t n = (size_t)db->nRec;
        int rc2;
        // + one chunk of padding: the scan kernel's loads run up to a chunk past a unit's end
        if ((rc2 = dalloc(&db->d_pse, n + IGD_CHUNK, &db->resident)) != IGD_HIP_OK ||
            (rc2 = dalloc(&db->d_px, n + IGD_CHUNK, &db->resident)) != IGD_HIP_OK ||
            (db->gType == 1 && (rc2 = dalloc(&db->d_pxv, n + IGD_CHUNK, &db->resident)) != IGD_HIP_OK)) {
            igd_hip_close(db);
            return rc2;
        }
        hipError_t e = hipMemset(db->d_ctl, 0, IGD_CTL_WORDS * 4);
        // the chunk of padding behind the dataset numbers is READ (igd_scan_sorted lets the lanes past the last unit's end
        // name whatever datasets follow): it has to hold valid numbers
        if (e == hipSuccess) e = hipMemsetAsync(db->d_px + n, 0, IGD_CHUNK * sizeof(uint16_t), db->stream);
        if (e == hipSuccess && db->d_pxv) e = hipMemsetAsync(db->d_pxv + n, 0, IGD_CHUNK * sizeof(uint32_t), db->stream);
        if (e == hipSuccess) {
            k_pack_units<<<256 * 8, 256, 0, db->stream>>>(v, db->d_units, db->d_pse, db->d_px, db->d_pxv, db->d_ctl);
            e = hipStreamSynchronize(db->stream);
        }
        int32_t fl = 0;
        if (e == hipSuccess) e = hipMemcpy(&fl, db->d_ctl, 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemset(db->d_ctl, 0, IGD_CTL_WORDS * 4);
        if (e != hipSuccess) {
            set_err("pack", e, __FILE__, __LINE__);
            igd_hip_close(db);
            return IGD_HIP_ERR_DEVICE;
        }
        db->packedV = db->gType == 1 && !(fl & 1);
        if (fl & 2) db->packed = false;          // a record outside its tile: exact arrays only
        v.pse = db->d_pse; v.px = db->d_px; v.pxv = db->d_pxv;
        // the DIRECT step's per-tile table (scan_direct.hpp): power-of-two tiles of at most 2^14 bp (two tile widths fit 16 bits),
        // record numbers that fit 32 bits, contig tables that fit its LDS arrays
        v.tileD = nullptr;
        bool everyCtgHasTiles = true;                    // (a contig without tiles would lend its queries to a neighbour's range)
        for (int c = 0; c < d->nCtg; c++) everyCtgHasTiles = everyCtgHasTiles && d->nTile[c] > 0;
        if (db->packed && v.shift >= 0 && v.shift <= 14 && db->nCtg <= QB_CTG && !db->bigImage && (int64_t)n + IGD_CHUNK < (1ll << 30) && db->nT > 0 && everyCtgHasTiles) {
            k_tile_desc<<<(db->nT + 255) / 256, 256, 0, db->stream>>>(v, db->d_tileD);
            if (hipStreamSynchronize(db->stream) == hipSuccess && hipGetLastError() == hipSuccess) v.tileD = db->d_tileD;
        }
    }
    OPEN_PHASE("compact image");
    t_arenaOwner = nullptr;
    // A file bucketed with another tile width than the image is made for (the reference accepts -b 11..19,
    // src/igd_create.c:454-457): tiles of 2^16 .. 2^19 bp do not fit the compact image's 16-bit offsets, tiles of 2^11 .. 2^13 bp
    // make units of a few dozen records whose fixed cost dominates.  The counting searches of such a database run on a
    // RE-TILED copy -- the same records bucketed again in tiles of 2^14 bp by the engine's own `create` path -- which is a
    // database of its own (db->inner) plus what the file's tiling decides (DbView::vshift).  Enumeration, the hit map and
    // Seqpare depend on the file's tiles and record order and stay on this image.
    {
        int sh = -1;
        for (int b = 0; b < 31; b++) if (d->nbp == (1 << b)) sh = b;
        const bool want = sh >= 0 && sh != 14 && sh != 15 && n > 0 && !getenv("IGD_HIP_NO_RETILE") && d->nFiles > 0;
        if (want) {
            const int rc3 = build_retiled(db, d, sh, units, device);
            if (rc3 != IGD_HIP_OK) {
                // (out of memory for the second copy, say: the database still works over its own tiles, only slower)
                const char *force = getenv("IGD_HIP_RETILE");
                if (force && !strcmp(force, "force")) { igd_hip_close(db); return rc3; }
                if (tim) fprintf(stderr, "[igd timing]   open: no re-tiled copy (%s): searching the file's own tiles\n", g_err);
                if (db->inner) { igd_hip_close(db->inner); db->inner = nullptr; }
            }
            OPEN_PHASE("re-tiled copy (2^14 bp)");
        }
    }
    *out = db;
    return IGD_HIP_OK;
}
