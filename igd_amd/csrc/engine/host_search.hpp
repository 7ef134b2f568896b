// engine/host_search.hpp -- part of igd_hip.hip (included there once; not a stand-alone header).
// workspaces, launches, igd_hip_search_dev / _runs_dev / _search / _search_ex, sync

// workspace for `nq` queries with `pairBytes` per pair slot
// Per-batch workspace.  pairBytes == 0: the caller promised an ordered batch -- only the merge join's
// arrays are needed (the CLI's common case: no 100+ MB of bucket structures to allocate first).
static int ensure_workspace(igd_hip_db *db, int64_t nq, int pairBytes)
{
    int rc;
    if (nq > db->wsQueries) {
        HIPCHK(hipDeviceSynchronize());
        if (db->d_fix) (void)hipFree(db->d_fix);
        if (db->d_qw) (void)hipFree(db->d_qw);
        if (db->d_later) (void)hipFree(db->d_later);
        if (db->d_laterHdr) (void)hipFree(db->d_laterHdr);
        db->d_fix = nullptr; db->d_qw = nullptr; db->d_later = nullptr; db->d_laterHdr = nullptr;
        db->wsQueries = 0;
        if ((rc = dalloc(&db->d_fix, (size_t)nq * 2, nullptr)) != IGD_HIP_OK) return rc;   // a query can be both long and WALK_FIRST
        if ((rc = dalloc(&db->d_qw, (size_t)nq + 64, nullptr)) != IGD_HIP_OK) return rc;
        if ((rc = dalloc(&db->d_later, (size_t)nq + 4096 + 64, nullptr)) != IGD_HIP_OK) return rc;   // whole later blocks (<= 4096 queries)
        if ((rc = dalloc(&db->d_laterHdr, 2 * ((size_t)nq / 256 + 2), nullptr)) != IGD_HIP_OK) return rc;
        db->wsQueries = nq;
    }
    if (pairBytes == 0 || (nq <= db->wsBucket && pairBytes <= db->pairBytes)) return IGD_HIP_OK;
    HIPCHK(hipDeviceSynchronize());
    const int64_t cap = nq > db->wsBucket ? nq : db->wsBucket;
    const int pb = pairBytes > db->pairBytes ? pairBytes : db->pairBytes;
    {
        void *ws[] = { db->d_pairs, db->d_long, db->d_spTable, db->d_spT, db->d_spSub };
        for (void *q : ws) if (q) (void)hipFree(q);
        db->d_pairs = nullptr; db->d_long = nullptr; db->d_spTable = nullptr; db->d_spT = nullptr; db->d_spSub = nullptr;
    }
    db->wsBucket = 0;
    if ((rc = dalloc((char **)&db->d_pairs, (size_t)cap * IGD_SHORT_TILES * (size_t)pb, nullptr)) != IGD_HIP_OK) return rc;
    if ((rc = dalloc(&db->d_long, (size_t)cap, nullptr)) != IGD_HIP_OK) return rc;
    {   // split path geometry: <= SP_MAXC coarse buckets of 2^shift tiles, 2^shift counters x 2 in LDS
        int sh = SP_MINSHIFT;
        while (sh < 13 && ((db->nT + (1 << sh) - 1) >> sh) > SP_MAXC) sh++;
        db->spShift = (((db->nT + (1 << sh) - 1) >> sh) <= SP_MAXC && sh <= 12) ? sh : -1;   // 2 * 4 * 2^12 = 32 KiB of LDS
        db->spCoarse = db->spShift >= 0 ? (db->nT + (1 << sh) - 1) >> sh : 0;
        if (db->spShift >= 0) {
            const size_t nWG = (size_t)((cap + SP_Q - 1) / SP_Q);
            if ((rc = dalloc(&db->d_spTable, nWG * (size_t)db->spCoarse, nullptr)) != IGD_HIP_OK) return rc;
            if ((rc = dalloc(&db->d_spT, nWG * SP_CAP, nullptr)) != IGD_HIP_OK) return rc;
            const size_t subN = (((size_t)db->spCoarse * SPF_S) << db->spShift) + 2 * (size_t)db->spCoarse + 16;
            if ((rc = dalloc(&db->d_spSub, subN, nullptr)) != IGD_HIP_OK) return rc;
            HIPCHK(hipMemset(db->d_spSub, 0, subN * 4));           // (bucketLong[] holds epoch stamps)
        }
    }
    db->wsBucket = cap;
    db->pairBytes = pb;
    return IGD_HIP_OK;
}

// The bucket step (count -> scan -> scatter).  gate != 0: every kernel returns at once unless
// k_query_bounds marked this batch unsorted (ctl[CTL_UNSORTED] == gate).  Leaves the pair
// counts in d_pairN, the range ends in d_pairPos, and d_pairCnt zeroed again.
static int launch_bucket(igd_hip_db *db, const int32_t *d_ichr, const int32_t *d_qs, const int32_t *d_qe,
                         int nq, int rule, int gate, int packed, hipStream_t st, u64 *zeroHits = nullptr,
                         u64 *zeroTotal = nullptr)
{
    const int nT = db->nT;
    const int qb = (nq + 255) / 256;
    const int qbz = zeroHits ? ((nq > db->nFiles ? nq : db->nFiles) + 255) / 256 : qb;
    k_count_pairs<<<qbz, 256, 0, st>>>(db->v, d_ichr, d_qs, d_qe, nq, rule, packed, db->d_pairCnt, db->d_long, db->d_ctl,
                                      gate, db->epoch, zeroHits, zeroTotal);
    const int sb = (nT + IGD_SCAN_TILE - 1) / IGD_SCAN_TILE;
    k_scan_block_sums<<<sb, IGD_SCAN_BLOCK, 0, st>>>(db->d_pairCnt, nT, db->d_blockSums, db->d_ctl, gate);
    k_scan_apply<<<sb, IGD_SCAN_BLOCK, 0, st>>>(db->d_pairCnt, nT, db->d_blockSums, db->d_pairPos, db->d_pairN,
                                                db->d_ctl, gate);
    k_scatter_pairs<<<qb, 256, 0, st>>>(db->v, d_ichr, d_qs, d_qe, nq, rule, packed, db->d_pairPos, db->d_pairs,
                                                db->d_ctl, gate);
    HIPCHK(hipGetLastError());
    return IGD_HIP_OK;
}

// The same grouping without global atomics (k_split_*); (qs,qe) pairs only.
static int launch_split(igd_hip_db *db, const int32_t *d_ichr, const int32_t *d_qs, const int32_t *d_qe,
                        int nq, int rule, int gate, int packed, hipStream_t st, u64 *zeroHits, u64 *zeroTotal)
{
    const int nWG = (nq + SP_Q - 1) / SP_Q;
    static const bool oneWG = getenv("IGD_HIP_SPLIT_ONE") != nullptr;      // A/B: one workgroup per coarse bucket, whatever the batch (until round 4)
    const bool shared = !oneWG && db->d_spSub != nullptr;
    uint32_t *bases = shared ? db->d_spSub + (((size_t)db->spCoarse * SPF_S) << db->spShift) : nullptr;
    int32_t *blong = shared ? (int32_t *)(bases + db->spCoarse) : nullptr;
    // (one bit per tile in LDS when they fit 40 KiB -- up to 327 680 tiles: hg38 in 16 kbp tiles has 188 505)
    static const bool noBits = getenv("IGD_HIP_SPLIT_NOBITS") != nullptr;            // A/B: tileCnt[] gathers (until round 5)
    const int bitsWords = (!noBits && db->v.tileBits && db->nT <= 40 * 1024 * 8) ? (db->nT + 31) / 32 + 1 : 0;   // (+ the array's word of slack)
    const int ctgStaged = (!noBits && db->nCtg <= 1024) ? db->nCtg : 0;                // ... and the two contig tables (8 KiB at most)
    // ... and the workgroup's region of tuples, put together in the same LDS before it is written out (SP_Q queries + 1/8 for
    // second tiles: 54 KiB; with the kernel's 8.3 KiB of counters within the 64 KiB a launch gets without asking)
    static const bool noRegion = getenv("IGD_HIP_SPLIT_NOREGION") != nullptr;       // A/B: tuples stored one by one (until round 5)
    const int stageCap = (noBits || noRegion) ? 0 : SP_Q + SP_Q / 8;
    size_t ldsLocal = (size_t)bitsWords * 4 + (size_t)ctgStaged * 8;
    if (ldsLocal < (size_t)stageCap * 12 + 16) ldsLocal = (size_t)stageCap * 12 + 16;
    static const bool noFast = getenv("IGD_HIP_SPLIT_NOFAST") != nullptr;           // A/B: the general per-query code for every database
    if (!noFast && bitsWords && ctgStaged && db->v.vshift < 0 && db->v.shift >= 0 && db->spShift >= 2)
    k_split_local<true><<<nWG, SP_WG, ldsLocal, st>>>(db->v, d_ichr, d_qs, d_qe, nq, rule, packed, db->spShift, db->spCoarse, db->d_spTable,
                                         db->d_spT, db->d_long, db->d_ctl, gate, db->epoch, zeroHits, zeroTotal, blong, bitsWords, ctgStaged, stageCap);
    else
    k_split_local<false><<<nWG, SP_WG, ldsLocal, st>>>(db->v, d_ichr, d_qs, d_qe, nq, rule, packed, db->spShift, db->spCoarse, db->d_spTable,
                                         db->d_spT, db->d_long, db->d_ctl, gate, db->epoch, zeroHits, zeroTotal, blong, bitsWords, ctgStaged, stageCap);
    // staging area of the usual bucket (split_fine_whole): twice the bucket's share of an evenly spread batch, within 64 KiB of LDS
    static const bool noStage = getenv("IGD_HIP_SPLIT_NOSTAGE") != nullptr;       // A/B: segments walked from memory, twice (until round 5)
    const size_t ldsCnt = (size_t)2 * 4 << db->spShift;
    int64_t cap = 2 * (((int64_t)nq + nq / 8) / db->spCoarse) + 256;
    if (cap < 1024) cap = 1024;
    if (cap > (int64_t)((65536 - 256 - ldsCnt) / 12)) cap = (int64_t)((65536 - 256 - ldsCnt) / 12);
    if (noStage || cap < 256) cap = 0;
    if (!shared)
    k_split_fine<<<db->spCoarse, SPF_WG, ldsCnt + (size_t)cap * 12, st>>>(db->nT, db->spShift, db->spCoarse, nWG, db->d_spTable,
                                                                          db->d_spT,
                                                                          db->d_pairN, db->d_pairPos, (int2 *)db->d_pairs,
                                                                          db->d_ctl, gate, db->d_ctl, db->epoch, packed ? db->d_heavy : nullptr, (int)cap);
    else {
        k_split_fine_a<<<db->spCoarse * SPF_S, SPF_WG, ldsCnt + (size_t)cap * 12, st>>>(db->nT, db->spShift, db->spCoarse, nWG, db->d_spTable, db->d_spT,
            db->d_spSub, bases, blong, db->d_pairN, db->d_pairPos, (int2 *)db->d_pairs, db->d_ctl, gate, db->d_ctl, db->epoch, packed ? db->d_heavy : nullptr, (int)cap);
        // (an empty launch when no bucket is piled up: 4.5 us, with 96 workgroups as with 512)
        k_split_fine_b<<<db->spCoarse * SPF_S < 512 ? db->spCoarse * SPF_S : 512, SPF_WG, (size_t)4 << db->spShift, st>>>(db->nT, db->spShift, db->spCoarse, nWG, db->d_spTable, db->d_spT,
            db->d_spSub, bases, blong, db->d_pairN, db->d_pairPos, (int2 *)db->d_pairs, db->d_ctl, gate, db->d_ctl, db->epoch, packed ? db->d_heavy : nullptr);
    }
    HIPCHK(hipGetLastError());
    return IGD_HIP_OK;
}

// the merge join's arguments for one batch (also handed to the batch's last launch, which hosts its skew valve)
static SortK make_sortk(igd_hip_db *db, const ScanArgs &a)
{
    SortArgs sa;
    sa.firstQ = a.firstQ; sa.spill = db->d_spill; sa.qw0 = db->d_qw; sa.later = db->d_later; sa.q_qs = a.q_qs; sa.ctl = a.ctl;
    sa.laterHdr = (const int2 *)db->d_laterHdr; sa.lbShift = db->lbShift; sa.lpos = db->d_lpos;
    sa.nq = a.nq; sa.v = a.v; sa.epoch = a.epoch; sa.mode = a.mode; sa.out = a.out; sa.rule = a.rule;
    sa.sbCap = db->sbCap; sa.wldsBytes = IGD_WLDS_BYTES + 2 * db->sbCap;
    sa.ctlw = db->d_ctl; sa.heavyS = db->d_heavy + IGD_HEAVY_MAX; sa.farList = db->d_far; sa.tailHistOff = -1; sa.noList = 0;
    SortK K;
    K.db = db->v; K.a = sa; K.hitsOut = (u64 *)a.hitsOut; K.totalOut = a.total;
    return K;
}

// One launch of the merge join's kernel.  A timed launch (igd_hip_profile_begin) of a promised-sorted batch passes its event
// pair to the launch itself: hipExtLaunchKernel stamps them with the dispatch's own start and end -- the figures a profiler
// reads (rocprofv3 --kernel-trace) -- where two hipEventRecord packets around the kernel also time the two packet gaps
// (3-6 % of a 70 us kernel) and put two more packets between the step's kernels.
template <typename F, typename KA>
static void launch_sorted(igd_hip_db *db, F kernel, int grid, int block, size_t lds, hipStream_t st, const KA &K)
{
    if (db->evStart) {
        KA k = K;
        void *args[] = {&k};
        (void)hipExtLaunchKernel((const void *)kernel, dim3(grid), dim3(block), args, lds, st, db->evStart, db->evStop, 0);
        db->evStart = db->evStop = nullptr;              // (one kernel per pair)
    } else kernel<<<grid, block, lds, st>>>(K);
}

// win >= 0: pass `win` of a batch against a database with more files than LDS counters (igd_hip_db::winN)
template <bool USE_V, bool LDS_HITS, bool PACKED>
static void launch_scan(igd_hip_db *db, const ScanArgs &a, hipStream_t st, int win = -1)
{
    const int fileLo = win > 0 ? win * db->winN : 0;
    const int fileN = win >= 0 ? (db->nFiles - fileLo < db->winN ? db->nFiles - fileLo : db->winN) : db->nFiles;
    const size_t lds = LDS_HITS ? db->ldsBytes : 0;
    if (a.mode != 2 && PACKED) {                         // merge join over the compact image: its own kernel
        const bool big = db->bigImage || db->nRec + IGD_CHUNK >= (1ll << 30);
        const size_t ldsS = (size_t)db->ldsSorted;
        SortK K = make_sortk(db, a);
        if (win >= 0) { K.db.nFiles = fileN; K.db.fileLo = fileLo; K.hitsOut += fileLo; K.a.noList = win > 0 ? 1 : 0; }
        // sparse on average (fewer than 28 queries per tile): the lean build, whose pairwise path is not burdened with the rank
        // method's registers; tiles that are dense all the same go to heavy_sorted_body
        const int forceRank = db->forceRank;              // tests: 0 lean, 1 full (IGD_HIP_RANK, read at open)
        // ... and a batch that visits a fraction of the units (fewer queries than tiles) runs the full build too: it steps
        // through the visited units only (10^3 queries: 43.7 -> 13.4 us, 10^5: 44.3 -> 35.7 us; 3 x 10^5: 51.5 vs 53.5 us)
        // (the rank method starts at 32 queries per tile; below an average of ~28 the full build mostly runs its pairwise path,
        // at 6 instead of 8 waves per SIMD -- measured lean / full, same box: 8 per tile 79.8 / 92.7 us, 16: 104 / 122,
        // 21: 119 / 137, 32: 154 / 147)
        bool lean = forceRank >= 0 ? forceRank == 0 : ((int64_t)a.nq < 28ll * db->nT && (int64_t)a.nq >= (int64_t)db->nT);
        // the lean build's 32-bit workgroup counters need no run-time guard when even a workgroup whose every unit is as
        // dense as that build lets one be stays below 2^32 (any database below ~10^9 records); otherwise: the full build
        const int64_t wavesLean = (int64_t)db->grid * (IGD_WG_LEAN / IGD_WAVE);
        if (((int64_t)db->nUnits + wavesLean - 1) / wavesLean * (IGD_WG_LEAN / IGD_WAVE) * (IGD_LEAN_FIRST + IGD_WAVE) * IGD_CHUNK >= (1ll << 32)) lean = false;
        // <USE_V, LDS_HITS, CNT32, BIG, RANK>: workgroups with LDS counters keep them in 32 bits (igd_scan_sorted guards the range itself)
        // (a database of one file / of up to eight: builds whose lanes do not all add to the same few LDS counters)
        const int few = (LDS_HITS && !big) ? (win >= 0 ? 3 : db->nFiles == 1 ? 1 : db->nFiles <= 8 ? 2 : 0) : 0;
        if (big) launch_sorted(db, igd_scan_sorted<USE_V, LDS_HITS, LDS_HITS, true, true>, db->grid, IGD_WG_RANK, ldsS, st, K);
        else if (lean) {
            const size_t l = LDS_HITS ? ldsS : 0;         // (the lean build's only LDS is its counters)
            if (few == 1) launch_sorted(db, igd_scan_sorted<USE_V, LDS_HITS, LDS_HITS, false, false, LDS_HITS ? 1 : 0>, db->grid, IGD_WG_LEAN, l, st, K);
            else if (few == 2) launch_sorted(db, igd_scan_sorted<USE_V, LDS_HITS, LDS_HITS, false, false, LDS_HITS ? 2 : 0>, db->grid, IGD_WG_LEAN, l, st, K);
            else if (few == 3) launch_sorted(db, igd_scan_sorted<USE_V, LDS_HITS, LDS_HITS, false, false, LDS_HITS ? 3 : 0>, db->grid, IGD_WG_LEAN, l, st, K);
            else launch_sorted(db, igd_scan_sorted<USE_V, LDS_HITS, LDS_HITS, false, false>, db->grid, IGD_WG_LEAN, l, st, K);
        } else {
            if (few == 1) launch_sorted(db, igd_scan_sorted<USE_V, LDS_HITS, LDS_HITS, false, true, LDS_HITS ? 1 : 0>, db->grid, IGD_WG_RANK, ldsS, st, K);
            else if (few == 2) launch_sorted(db, igd_scan_sorted<USE_V, LDS_HITS, LDS_HITS, false, true, LDS_HITS ? 2 : 0>, db->grid, IGD_WG_RANK, ldsS, st, K);
            else if (few == 3) launch_sorted(db, igd_scan_sorted<USE_V, LDS_HITS, LDS_HITS, false, true, LDS_HITS ? 3 : 0>, db->grid, IGD_WG_RANK, ldsS, st, K);
            else launch_sorted(db, igd_scan_sorted<USE_V, LDS_HITS, LDS_HITS, false, true>, db->grid, IGD_WG_RANK, ldsS, st, K);
        }
    } else
    if (a.mode != 2) {
        if (win >= 0) { DbView v = db->v; v.nFiles = fileN; v.fileLo = fileLo; igd_scan_tiles<true, USE_V, LDS_HITS, PACKED, LDS_HITS><<<db->grid, IGD_WG, lds, st>>>(v, a); }
        else igd_scan_tiles<true, USE_V, LDS_HITS, PACKED><<<db->grid, IGD_WG, lds, st>>>(db->v, a);
    }
    if (a.mode != 1) {
        if (win >= 0) { DbView v = db->v; v.nFiles = fileN; v.fileLo = fileLo; igd_scan_tiles<false, USE_V, LDS_HITS, PACKED, LDS_HITS><<<db->grid, IGD_WG, lds, st>>>(v, a); }
        else igd_scan_tiles<false, USE_V, LDS_HITS, PACKED><<<db->grid, IGD_WG, lds, st>>>(db->v, a);
    }
}
template <bool LDS_HITS>
static void launch_scan_any(igd_hip_db *db, const ScanArgs &a, bool useV, bool packed, hipStream_t st, int win = -1)
{
    if (packed) {
        if (useV) launch_scan<true, LDS_HITS, true>(db, a, st, win); else launch_scan<false, LDS_HITS, true>(db, a, st, win);
    } else {
        if (useV) launch_scan<true, LDS_HITS, false>(db, a, st, win); else launch_scan<false, LDS_HITS, false>(db, a, st, win);
    }
}

// runs -> one contig number per query (only for the batches k_query_bounds' RUNS build does not take: see search_dev_impl)
__global__ void k_expand_runs(const int32_t *__restrict__ runs, int nCtg, int32_t *__restrict__ ichr, int nq)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    int lo = 0, hi = nCtg;                               // largest c with runs[c] <= i
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (runs[mid] <= i) lo = mid; else hi = mid; }
    ichr[i] = lo;
}

static int search_dev_impl(igd_hip_db *db, const int32_t *d_ichr, const int32_t *d_runs, const int32_t *d_qs,
                           const int32_t *d_qe, int64_t nq, int32_t v, int rule, int flags,
                           int64_t *d_hits, int64_t *d_total, void *stream);

extern "C" int igd_hip_search_dev(igd_hip_db *db, const int32_t *d_ichr, const int32_t *d_qs,
                                  const int32_t *d_qe, int64_t nq, int32_t v, int rule, int flags,
                                  int64_t *d_hits, int64_t *d_total, void *stream)
{
    return search_dev_impl(db, d_ichr, nullptr, d_qs, d_qe, nq, v, rule, flags, d_hits, d_total, stream);
}

extern "C" int igd_hip_search_runs_dev(igd_hip_db *db, const int32_t *d_run_start, const int32_t *d_qs,
                                       const int32_t *d_qe, int64_t nq, int32_t v, int rule, int flags,
                                       int64_t *d_hits, int64_t *d_total, void *stream)
{
    if (!d_run_start || (flags & IGD_HIP_FLAG_BUCKET)) {
        snprintf(g_err, sizeof g_err, "igd_hip_search_runs_dev: bad argument");
        return IGD_HIP_ERR_ARG;
    }
    return search_dev_impl(db, nullptr, d_run_start, d_qs, d_qe, nq, v, rule, flags | IGD_HIP_FLAG_SORTED, d_hits, d_total, stream);
}

static int search_dev_impl(igd_hip_db *db, const int32_t *d_ichr, const int32_t *d_runs, const int32_t *d_qs,
                           const int32_t *d_qe, int64_t nq, int32_t v, int rule, int flags,
                           int64_t *d_hits, int64_t *d_total, void *stream)
{
    if (!db || !d_hits || nq < 0 || nq > IGD_MAX_BATCH || (rule != IGD_HIP_RULE_NEST && rule != IGD_HIP_RULE_FLAT) ||
        ((flags & IGD_HIP_FLAG_SORTED) && (flags & IGD_HIP_FLAG_BUCKET))) {
        snprintf(g_err, sizeof g_err, "igd_hip_search_dev: bad argument");
        return IGD_HIP_ERR_ARG;
    }
    if (db->nFiles == 0) return IGD_HIP_OK;
    if (db->inner) {                                     // a file of another tile width: counted on the re-tiled copy (igd_hip_open)
        db->inner->vnest = rule == IGD_HIP_RULE_NEST ? 1 : 0;
        return search_dev_impl(db->inner, d_ichr, d_runs, d_qs, d_qe, nq, v, IGD_HIP_RULE_FLAT, flags, d_hits, d_total,
                               stream ? stream : (void *)db->stream);
    }
    HIPCHK(hipSetDevice(db->device));
    hipStream_t st = stream ? (hipStream_t)stream : db->stream;
    if (nq == 0 || db->nT == 0) {
        if (flags & IGD_HIP_FLAG_ZERO_FIRST) {
            HIPCHK(hipMemsetAsync(d_hits, 0, (size_t)db->nFiles * 8, st));
            if (d_total) HIPCHK(hipMemsetAsync(d_total, 0, 8, st));
        }
        return IGD_HIP_OK;
    }
    int rc = ensure_workspace(db, nq, (flags & IGD_HIP_FLAG_SORTED) ? 0 : 8);
    if (rc != IGD_HIP_OK) return rc;
    const bool useV = (v != IGD_HIP_NO_VALUE_FILTER && db->gType == 1);   // gType 0 has no value field
    const int mode = (flags & IGD_HIP_FLAG_SORTED) ? 1 : (flags & IGD_HIP_FLAG_BUCKET) ? 2 : 0;
    const int krule = rule | ((db->v.vshift >= 0 && db->vnest) ? 0x100 : 0);   // what the grouping kernels get: bit 8 = rule NEST on the FILE's tiles (a re-tiled copy)
    const bool packed = db->packed && !(flags & IGD_HIP_FLAG_EXACT) && (!useV || db->packedV);
    if (db->epoch >= 0x3fffffff || db->covStale) {       // the epoch stamps start over (or a batch ended before its last launch)
        HIPCHK(hipMemsetAsync(db->d_spill, 0, ((size_t)db->nT + 2) * 4, st));
        HIPCHK(hipMemsetAsync(db->d_cov, 0, 4 * IGD_COV_LEN(db->nT) * 4, st));
        HIPCHK(hipMemsetAsync(db->d_ctl + CTL_COV, 0, 4 * 4, st));
        if (db->d_spSub && db->spShift >= 0)                 // (the piled-up buckets' epoch stamps)
            HIPCHK(hipMemsetAsync(db->d_spSub + (((size_t)db->spCoarse * SPF_S) << db->spShift) + db->spCoarse, 0, (size_t)db->spCoarse * 4, st));
        if (!db->covStale) db->epoch = 0;
        db->covStale = false;
    }
    db->epoch++;
    // From here on kernels of this batch may have written the coverage difference arrays: ANY error exit before the batch's
    // last launch (a failed hipEventRecord / hipGetLastError as much as a failed launch_split) must have them cleared
    // before the next batch -- two batches later (same parity) long queries would otherwise add to stale +1 / -1 entries.
    struct StaleGuard { igd_hip_db *d; bool done; ~StaleGuard() { if (!done) d->covStale = true; } } guard{db, false};
    int slot = -1;
    if (db->evOn && db->evUsed < db->evMax && (db->evSeen++ % (db->evEvery > 0 ? db->evEvery : 1)) == 0) slot = db->evUsed++;
    // the whole pipeline is bracketed for the first IGD_PIPE_EVENTS launches only: every event is one more packet in the
    // stream between two kernels, and the scan kernel's own pair is the one every timed launch needs
    const bool pipeEv = slot >= 0 && slot < IGD_PIPE_EVENTS;
    if (pipeEv) HIPCHK(hipEventRecord(db->ev[4 * slot + 0], st));
    // IGD_HIP_FLAG_ZERO_FIRST: the first kernel of the batch clears hits[] (and total)
    u64 *zh = (flags & IGD_HIP_FLAG_ZERO_FIRST) ? (u64 *)d_hits : nullptr;
    u64 *zt = (flags & IGD_HIP_FLAG_ZERO_FIRST) ? (u64 *)d_total : nullptr;
    // The DIRECT step (scan_direct.hpp): a batch promised sorted AND short that is dense on average -- no per-query pre-pass,
    // the scan kernel reads q_qs / q_qe itself.  (IGD_HIP_DIRECT=1 at open: every promised-sorted batch that can -- tests.)
    const bool direct = mode == 1 && packed && db->ldsHits && db->nWin == 1 && db->v.tileD != nullptr && db->v.vshift < 0 && db->v.shift >= 0 && db->nCtg <= QB_CTG &&
                        (db->forceDirect > 0 || (db->forceDirect < 0 && (flags & IGD_HIP_FLAG_SHORT) && nq >= 28ll * db->nT));
    db->lastDirect = direct ? 1 : 0;
    if (mode != 2) {
        bool vec = ((((uintptr_t)d_ichr) | ((uintptr_t)d_qs) | ((uintptr_t)d_qe)) & 15) == 0;   // our own word arrays are aligned
        if (db->qbVec1) vec = false;                  // A/B (IGD_HIP_QB_VEC1, read at open)
        const bool fast = packed && db->v.shift >= 0 && db->nCtg <= QB_CTG;
        // a small batch: one query per thread (more waves share the gaps between its queries), and enough workgroups for
        // the head and tail of firstQ[] -- 10^3 queries left 190 000 entries to ONE workgroup: 90 us
        if (nq < 65536) vec = false;
        // a batch given as contig runs: k_query_bounds' RUNS build takes it as it is when it is the usual kind (compact image,
        // power-of-two tiles, four queries per thread); any other batch gets its contig numbers written out first
        bool runsK = d_runs != nullptr && vec && fast;
        if (d_runs && !runsK) {
            if (nq > db->runCap) {
                HIPCHK(hipStreamSynchronize(st));
                if (db->d_runIchr) (void)hipFree(db->d_runIchr);
                db->d_runIchr = nullptr; db->runCap = 0;
                if ((rc = dalloc(&db->d_runIchr, (size_t)nq, nullptr)) != IGD_HIP_OK) return rc;
                db->runCap = nq;
            }
            k_expand_runs<<<(int)((nq + 255) / 256), 256, 0, st>>>(d_runs, db->nCtg, db->d_runIchr, (int)nq);
            d_ichr = db->d_runIchr;
            vec = vec && (((uintptr_t)d_ichr) & 15) == 0;
        }
        const int fillBlocks = (int)((db->nT >> 10) < 256 ? (db->nT >> 10) + 1 : 256);
        // large batches: later blocks of 4096 queries (workgroups of 1024 threads), so that the candidate range of a tile --
        // the queries of three tiles -- spans at most two blocks even at hundreds of queries per tile
        const bool wide = vec && nq >= ((int64_t)1 << 22);
#define QB_GRID(PER_) ((int)((nq + (PER_) - 1) / (PER_)) > fillBlocks ? (int)((nq + (PER_) - 1) / (PER_)) : fillBlocks)
#define QB_LAUNCH(VEC_, FAST_, WGT_)                                                                                                  \
    if (direct && FAST_ && runsK && VEC_ == 4)          /* the DIRECT step: the bounds alone (BONLY) */                               \
        k_query_bounds<4, true, WGT_, true, true><<<QB_GRID(WGT_ * 4), WGT_, 0, st>>>(db->v, d_runs, d_qs, d_qe, (int)nq, krule,      \
        1, db->d_firstQ, db->d_lpos, db->d_fix, db->d_ctl, db->epoch, zh, zt, db->d_qw, db->d_later, db->d_spill,                      \
        (int2 *)db->d_laterHdr, 1);                                                                                                  \
    else if (direct && FAST_)                                                                                                        \
        k_query_bounds<VEC_, true, WGT_, false, true><<<QB_GRID(WGT_ * VEC_), WGT_, 0, st>>>(db->v, d_ichr, d_qs, d_qe, (int)nq, krule, \
        1, db->d_firstQ, db->d_lpos, db->d_fix, db->d_ctl, db->epoch, zh, zt, db->d_qw, db->d_later, db->d_spill,                      \
        (int2 *)db->d_laterHdr, 1);                                                                                                  \
    else if (runsK && VEC_ == 4 && FAST_)                                                                                             \
        k_query_bounds<4, true, WGT_, true><<<QB_GRID(WGT_ * 4), WGT_, 0, st>>>(db->v, d_runs, d_qs, d_qe, (int)nq, krule,            \
        packed ? 1 : 0, db->d_firstQ, db->d_lpos, db->d_fix, db->d_ctl, db->epoch, zh, zt, db->d_qw, db->d_later, db->d_spill,         \
        (int2 *)db->d_laterHdr, mode == 1 ? 1 : 0);                                                                                  \
    else                                                                                                                             \
    k_query_bounds<VEC_, FAST_, WGT_><<<QB_GRID(WGT_ * VEC_), WGT_, 0, st>>>(db->v, d_ichr, d_qs, d_qe, (int)nq, krule,               \
        packed ? 1 : 0, db->d_firstQ, db->d_lpos, db->d_fix, db->d_ctl, db->epoch, zh, zt, db->d_qw, db->d_later, db->d_spill,         \
        (int2 *)db->d_laterHdr, mode == 1 ? 1 : 0)
#ifndef IGD_QB_WIDE
#define IGD_QB_WIDE 1024
#endif
        // (the bounds alone need no later blocks: workgroups of 256 -- more of them in flight, no barrier across 16 waves)
        if (wide && !(direct && !getenv("IGD_HIP_BONLY_WIDE"))) { if (fast) QB_LAUNCH(4, true, IGD_QB_WIDE); else QB_LAUNCH(4, false, IGD_QB_WIDE); }
        else if (vec) { if (fast) QB_LAUNCH(4, true, 256); else QB_LAUNCH(4, false, 256); }
        else { if (fast) QB_LAUNCH(1, true, 256); else QB_LAUNCH(1, false, 256); }
#undef QB_LAUNCH
#undef QB_GRID
        db->lbShift = wide ? (IGD_QB_WIDE == 1024 ? 12 : IGD_QB_WIDE == 512 ? 11 : 10) : vec ? 10 : 8;
    }
    if (mode != 1) {
        static const bool oldBucket = getenv("IGD_HIP_ATOMIC_BUCKETS") != nullptr;   // A/B: the counting sort with global atomics
        if (db->spShift >= 0 && !oldBucket)
            rc = launch_split(db, d_ichr, d_qs, d_qe, (int)nq, krule, mode == 2 ? 0 : db->epoch, packed ? 1 : 0, st,
                              mode == 2 ? zh : nullptr, mode == 2 ? zt : nullptr);
        else
            rc = launch_bucket(db, d_ichr, d_qs, d_qe, (int)nq, krule, mode == 2 ? 0 : db->epoch, packed ? 1 : 0, st,
                                      mode == 2 ? zh : nullptr, mode == 2 ? zt : nullptr);
        if (rc != IGD_HIP_OK) return rc;                 // (guard: covStale)
    }
    // (a promised-sorted batch over the compact image in one pass: the scan kernel's own dispatch carries the pair)
    const bool extEv = slot >= 0 && mode == 1 && packed && db->ldsHits && db->nWin == 1;
    if (slot >= 0 && !extEv) HIPCHK(hipEventRecord(db->ev[4 * slot + 1], st));
    ScanArgs a;
    a.firstQ = db->d_firstQ; a.pairN = db->d_pairN; a.pairPos = db->d_pairPos; a.pairs = (const int2 *)db->d_pairs;
    a.walkList = nullptr; a.ctl = db->d_ctl; a.q_ichr = d_ichr; a.q_qs = d_qs; a.q_qe = d_qe; a.q_w = db->d_qw;
    a.total = (u64 *)d_total; a.hitsOut = (u64 *)d_hits;
    a.nq = (int)nq; a.v = v; a.rule = rule; a.epoch = db->epoch; a.mode = mode;
    a.packedWalk = packed ? (useV ? 2 : 1) : 0;
    // the skew valves ride in the batch's last launch: bit 0 bucket path, bit 1 merge join, bit 2 BIG image
    const int valves = direct ? 8 : (mode != 1 && packed && db->spShift >= 0 ? 1 : 0) | (mode != 2 && packed ? 2 : 0) |
                       (db->bigImage || db->nRec + IGD_CHUNK >= (1ll << 30) ? 4 : 0);
    // (the valve's slices of IGD_HEAVY_SLICE queries are beyond any LDS array of query starts: its waves get none)
    const size_t tailWave = direct ? (size_t)IGD_D_WLDS : (size_t)IGD_WLDS_BYTES;                    // a wave's rank-method area in the last launch
    size_t tailLds = (valves & (2 | 8)) ? (size_t)(db->ldsHits ? IGD_TAIL_WG / IGD_WAVE : 4) * tailWave : 0;   // (k_exact_walk: workgroups of 4 waves)
    int tailHistOff = -1;                                // u64 counters for the exact walks and the coverage, when the files fit
    if ((size_t)db->nFiles * 8 <= (size_t)48 * 1024) { tailHistOff = (int)tailLds; tailLds += (size_t)db->nFiles * 8; }
    if (db->ldsHits) {
        a.out = db->d_slab;
        const SortK K = make_sortk(db, a);
        // (more files than LDS counters: one pass of scan + reduction per window of files; the batch's tail rides in the last)
        for (int win = 0; win < db->nWin; win++) {
            const bool last = win == db->nWin - 1;
            const int fileLo = win * db->winN, fileN = db->nWin == 1 ? db->nFiles : (db->nFiles - fileLo < db->winN ? db->nFiles - fileLo : db->winN);
            if (extEv) { db->evStart = db->ev[4 * slot + 1]; db->evStop = db->ev[4 * slot + 2]; }
            if (direct) {
                DirK D;
                D.db = db->v;
                D.a.firstQ = db->d_firstQ; D.a.tileD = db->v.tileD; D.a.q_qs = d_qs; D.a.q_qe = d_qe; D.a.ctl = db->d_ctl; D.a.fix = db->d_fix;
                D.a.heavyS = db->d_heavy + IGD_HEAVY_MAX; D.a.farList = db->d_far; D.a.nq = (int)nq; D.a.v = v; D.a.epoch = db->epoch;
                D.a.rule = rule; D.a.promised = 1; D.a.sbCap = db->sbCap; D.a.wldsBytes = IGD_D_WLDS + 2 * db->sbCap;
                D.a.out = db->d_slab; D.a.hitsOut = (u64 *)d_hits; D.a.totalOut = (u64 *)d_total;
                if (useV) launch_sorted(db, igd_scan_direct<true>, db->grid, IGD_WG_DIR, (size_t)db->ldsDirect, st, D);
                else launch_sorted(db, igd_scan_direct<false>, db->grid, IGD_WG_DIR, (size_t)db->ldsDirect, st, D);
            } else
            launch_scan_any<true>(db, a, useV, packed, st, db->nWin > 1 ? win : -1);
            if (extEv && db->evStart) {                  // (no merge-join launch took the pair: cannot happen for this kind of batch)
                db->evStart = db->evStop = nullptr;
                HIPCHK(hipEventRecord(db->ev[4 * slot + 1], st)); HIPCHK(hipEventRecord(db->ev[4 * slot + 2], st));
            }
            if (slot >= 0 && last && !extEv) HIPCHK(hipEventRecord(db->ev[4 * slot + 2], st));
            // slab rows -> hits[]; the listed exact walks (same launch) add straight into hits[] and total
            ScanArgs w = a;
            w.out = (u64 *)d_hits;
            SortK Kt = K;
            Kt.a.sbCap = 0; Kt.a.wldsBytes = (int)tailWave; Kt.a.tailHistOff = tailHistOff;
            // IGD_REDUCE_GROUPS row groups sum the slab; the launch is filled up to IGD_TAIL_WGS workgroups (8 waves per SIMD),
            // which find out from the batch's control words that the tail has nothing for them -- or share a long
            // exact-walk list and the coverage of long queries, whose loops are chains of dependent loads
            const int gx = (fileN + IGD_TAIL_WG - 1) / IGD_TAIL_WG;
            dim3 rg(gx, (!last || IGD_REDUCE_GROUPS * gx >= IGD_TAIL_WGS) ? IGD_REDUCE_GROUPS : (IGD_TAIL_WGS + gx - 1) / gx);
            const int rows32 = (mode != 2 && packed) ? db->epoch : 0;      // the merge join's kernel leaves 32-bit rows (CNT32)
            if (useV)
                k_reduce_slabs<true><<<rg, IGD_TAIL_WG, last ? tailLds : 0, st>>>(Kt, db->d_slab, db->grid, fileN, (u64 *)d_hits + fileLo, (u64 *)d_total,
                                                               db->d_ctl, mode == 1 ? db->epoch : 0, w, db->d_fix, db->d_long, db->d_heavy, last ? valves : -1, rows32);
            else
                k_reduce_slabs<false><<<rg, IGD_TAIL_WG, last ? tailLds : 0, st>>>(Kt, db->d_slab, db->grid, fileN, (u64 *)d_hits + fileLo, (u64 *)d_total,
                                                                db->d_ctl, mode == 1 ? db->epoch : 0, w, db->d_fix, db->d_long, db->d_heavy, last ? valves : -1, rows32);
        }
    } else {
        a.out = (u64 *)d_hits;
        const SortK K = make_sortk(db, a);
        if (d_total) k_sum_hits<<<1, 256, 0, st>>>((const u64 *)d_hits, db->nFiles, (u64 *)d_total, -1);
        launch_scan_any<false>(db, a, useV, packed, st);
        if (slot >= 0) HIPCHK(hipEventRecord(db->ev[4 * slot + 2], st));
        {   // total is taken from the growth of sum(hits) here (k_sum_hits), not by the walk
            ScanArgs w = a;
            w.total = nullptr;
            SortK Kt = K;
            Kt.a.sbCap = 0; Kt.a.wldsBytes = IGD_WLDS_BYTES; Kt.a.tailHistOff = tailHistOff;
            if (useV) k_exact_walk<true><<<1024, 256, tailLds, st>>>(Kt, w, db->d_fix, db->d_long, db->d_heavy, valves);
            else k_exact_walk<false><<<1024, 256, tailLds, st>>>(Kt, w, db->d_fix, db->d_long, db->d_heavy, valves);
        }
        if (d_total) k_sum_hits<<<1, 256, 0, st>>>((const u64 *)d_hits, db->nFiles, (u64 *)d_total, +1);
    }
    if (pipeEv) HIPCHK(hipEventRecord(db->ev[4 * slot + 3], st));
    if (mode == 1) db->promised = db->epoch;
    db->lastMode = mode; db->lastPacked = packed ? 1 : 0;
    HIPCHK(hipGetLastError());
    guard.done = true;
    return IGD_HIP_OK;
}

extern "C" int igd_hip_sync(igd_hip_db *db, void *stream)
{
    if (!db) return IGD_HIP_ERR_ARG;
    if (db->inner) return igd_hip_sync(db->inner, stream ? stream : (void *)db->stream);   // (the promise is kept track of where the batch ran)
    HIPCHK(hipSetDevice(db->device));
    HIPCHK(hipStreamSynchronize(stream ? (hipStream_t)stream : db->stream));
    HIPCHK(hipGetLastError());
    if (db->promised) {
        // a batch ran under IGD_HIP_FLAG_SORTED: the device recorded whether the promise held
        // -- stickily: ANY promised batch since the last sync that was found unordered is reported
        int32_t ctl[4] = {0, 0, 0, 0};
        HIPCHK(hipMemcpy(ctl, db->d_ctl, sizeof ctl, hipMemcpyDeviceToHost));
        const bool broken = ctl[CTL_BROKEN] != 0;
        db->promised = 0;
        if (broken) {
            const int32_t zero = 0;
            HIPCHK(hipMemcpy(db->d_ctl + CTL_BROKEN, &zero, 4, hipMemcpyHostToDevice));
            snprintf(g_err, sizeof g_err, "igd_hip: queries passed with IGD_HIP_FLAG_SORTED were not ordered by "
                     "(contig, start); such a batch added nothing to hits");
            return IGD_HIP_ERR_UNSORTED;
        }
    }
    return IGD_HIP_OK;
}

// The same, polling: hipStreamSynchronize parks the host thread on the stream's completion signal and is woken by an
// interrupt -- tens of microseconds after the last kernel has ended -- while hipStreamQuery reads the signal.  A job of a
// few milliseconds (bench.py's 20 timed steps) notices its end ~0.1 ms sooner.  Burns a host core while it waits.
extern "C" int igd_hip_sync_spin(igd_hip_db *db, void *stream)
{
    if (!db) return IGD_HIP_ERR_ARG;
    HIPCHK(hipSetDevice(db->device));
    hipStream_t st = stream ? (hipStream_t)stream : db->stream;
    for (;;) {
        const hipError_t e = hipStreamQuery(st);
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) { set_err("hipStreamQuery", e, __FILE__, __LINE__); return IGD_HIP_ERR_DEVICE; }
    }
    return igd_hip_sync(db, stream);
}

static int ensure_qstage(igd_hip_db *db, int64_t nq)
{
    if (nq <= db->qcap) return IGD_HIP_OK;
    HIPCHK(hipDeviceSynchronize());
    if (db->d_qc) (void)hipFree(db->d_qc);
    if (db->d_qs) (void)hipFree(db->d_qs);
    if (db->d_qe) (void)hipFree(db->d_qe);
    db->d_qc = db->d_qs = db->d_qe = nullptr;
    db->qcap = 0;
    int rc;
    if ((rc = dalloc(&db->d_qc, (size_t)nq, nullptr)) != IGD_HIP_OK) return rc;
    if ((rc = dalloc(&db->d_qs, (size_t)nq, nullptr)) != IGD_HIP_OK) return rc;
    if ((rc = dalloc(&db->d_qe, (size_t)nq, nullptr)) != IGD_HIP_OK) return rc;
    db->qcap = nq;
    return IGD_HIP_OK;
}

extern "C" int igd_hip_search(igd_hip_db *db, const int32_t *ichr, const int32_t *qs, const int32_t *qe,
                              int64_t nq, int32_t v, int rule, int64_t *hits, int64_t *total)
{
    return igd_hip_search_ex(db, ichr, qs, qe, nq, v, rule, 0, hits, total);
}

// One slab of host queries -> db->d_hits / db->d_total (cleared first), in engine batches; nothing is copied back.
static int search_slab_resident(igd_hip_db *db, const int32_t *ichr, const int32_t *qs, const int32_t *qe,
                                int64_t nq, int32_t v, int rule, int flags)
{
    HIPCHK(hipSetDevice(db->device));
    hipStream_t st = db->stream;
    HIPCHK(hipMemsetAsync(db->d_hits, 0, (size_t)db->nFiles * 8, st));
    HIPCHK(hipMemsetAsync(db->d_total, 0, 8, st));
    const int64_t step = max_batch();
    for (int64_t q0 = 0; q0 < nq; q0 += step) {
        int64_t m = nq - q0 < step ? nq - q0 : step;
        const bool timing = db->timing;
        auto now = []() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; };
        double t0 = now();
        int rc = ensure_qstage(db, m);
        if (rc != IGD_HIP_OK) return rc;
        if (timing) { (void)hipStreamSynchronize(st); fprintf(stderr, "[igd timing]   search: query staging alloc     %7.1f ms\n", now() - t0); t0 = now(); }
        HIPCHK(hipMemcpyAsync(db->d_qc, ichr + q0, (size_t)m * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(db->d_qs, qs + q0, (size_t)m * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(db->d_qe, qe + q0, (size_t)m * 4, hipMemcpyHostToDevice, st));
        if (timing) { (void)hipStreamSynchronize(st); fprintf(stderr, "[igd timing]   search: H2D queries             %7.1f ms\n", now() - t0); t0 = now(); }
        rc = igd_hip_search_dev(db, db->d_qc, db->d_qs, db->d_qe, m, v, rule, flags, db->d_hits, db->d_total, st);
        if (rc != IGD_HIP_OK) return rc;
        rc = igd_hip_sync(db, st);                       // staging buffers are reused; promise checked
        if (timing) fprintf(stderr, "[igd timing]   search: workspace + kernels        %7.1f ms\n", now() - t0);
        if (rc == IGD_HIP_ERR_UNSORTED) {
            // the caller's order promise did not hold for this slice (it added nothing): redo it
            // with the device choosing the grouping
            flags &= ~(IGD_HIP_FLAG_SORTED | IGD_HIP_FLAG_SHORT);
            rc = igd_hip_search_dev(db, db->d_qc, db->d_qs, db->d_qe, m, v, rule, flags, db->d_hits, db->d_total, st);
            if (rc == IGD_HIP_OK) rc = igd_hip_sync(db, st);
        }
        if (rc != IGD_HIP_OK) return rc;
    }
    return IGD_HIP_OK;
}

extern "C" int igd_hip_search_ex(igd_hip_db *db, const int32_t *ichr, const int32_t *qs, const int32_t *qe,
                                 int64_t nq, int32_t v, int rule, int flags, int64_t *hits, int64_t *total)
{
    if (!db || !hits || nq < 0 || (nq > 0 && (!ichr || !qs || !qe))) {
        snprintf(g_err, sizeof g_err, "igd_hip_search: bad argument");
        return IGD_HIP_ERR_ARG;
    }
    if (total) *total = 0;
    if (nq == 0 || db->nFiles == 0) return IGD_HIP_OK;
    const int rc = search_slab_resident(db, ichr, qs, qe, nq, v, rule, flags);
    if (rc != IGD_HIP_OK) return rc;
    std::vector<int64_t> h((size_t)db->nFiles);
    int64_t tot = 0;
    HIPCHK(hipMemcpy(h.data(), db->d_hits, (size_t)db->nFiles * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&tot, db->d_total, 8, hipMemcpyDeviceToHost));
    for (int32_t f = 0; f < db->nFiles; f++) hits[f] += h[f];
    if (total) *total = tot;
    return IGD_HIP_OK;
}
