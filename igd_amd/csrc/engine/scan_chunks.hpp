// engine/scan_chunks.hpp -- part of igd_hip.hip (included there once; not a stand-alone header).
// The DIRECT step, query-partitioned (round 6): ONE kernel reads every query once -- no pre-pass at all.
// ------------------------------------------------------------------------------------------
// Round 5's DIRECT step was k_query_bounds<.., BONLY> (firstQ[] = the first query of every tile: 8 bytes per query read, 21-24 us
// for 1.25e7 queries) -> igd_scan_direct (units dealt to waves, each unit looks its tile's query range up in firstQ[] and reads
// those queries: 8 more bytes per query) -> k_reduce_slabs.  Here the QUERIES are dealt to the waves instead:
//   * wave w owns the contiguous range [nq * w / nwaves, nq * (w + 1) / nwaves) of the position-sorted batch and works through it
//     in passes of at most IGD_C_PASS (960) queries;
//   * a pass reads contig numbers and starts (8 bytes per query, coalesced), works out every query's key -- the global number
//     of its first tile, clamped into its contig as src/igd_search.c:459-464 does -- checks the order promise on (key, start)
//     with the pass's predecessor as seam, and cuts the pass where the key changes: <= 64 SEGMENTS (tile, first query, count);
//   * the segments' tiles are expanded into (unit, first query, count) items, one per lane, and every item is counted by
//     d_compute (scan_direct.hpp) exactly as round 5's kernel counted (unit, the tile's whole range): the rank method over the
//     unit's records, the next tile's first records riding along, the exceptions listed for the exact walks.  d_compute reads
//     the item's qs / qe itself (the starts a second time -- from L2, the wave has just read them).
// Why the sums are right: per record, hits = #{q: start < qe} - #{q: end <= qs} over the queries of a tile is additive over ANY
// partition of those queries, so the waves on either side of a seam inside a tile each add their own part (src/igd_search.c:
// 479-493: the reference counts query by query).  What the first unit of a tile does per query -- listing it for a walk of its
// later tiles, pushing it into the next tile's first records -- is per query as well.
// What goes away with the tile-keyed ranges: firstQ[], the bounds pass and its launch, the heavy-tile slices (a tile with 10^6
// queries is simply 10^6 / 960 items spread over all waves) and every add that is not to a workgroup's own LDS counters: a
// batch found out of order by ANY wave has added nothing when k_reduce_slabs looks at the mark.
// 32-bit counters: the host only takes this step when (queries per wave) x (records of the fullest tile + what rides along)
// stays below a wave's share of 2^32 (chunks_fit): no run-time guard, no far list.
// HBM bytes: 12 per query (contig, start, end: once) + 6 per record of a visited unit (+ the <= 64 appended) + descriptors.

#define IGD_C_PASS 960                                  // queries per pass: 15 blocks of 64; < the LDS array of query starts (sbCap = 1024)
#define IGD_C_BLOCKS (IGD_C_PASS / IGD_WAVE)
#define IGD_C_SEGS 64                                   // segments per pass: one per lane
#define IGD_C_HAND 512                                  // a wave's border moves to the next tile's first query when that lies within this many queries

template <bool USE_V>
__global__ __launch_bounds__(IGD_WG_DIR) __attribute__((amdgpu_waves_per_eu(IGD_WPE_DIR, IGD_WPE_DIR))) void igd_scan_chunks(DirK K)
{
    const DbView &db = K.db;
    const DirArgs &a = K.a;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr int wavesPerWG = IGD_WG_DIR / IGD_WAVE;
    const int nFiles = KARGD(db.nFiles);
    const int nCtg = KARGD(db.nCtg);
    const size_t hitBytes = ((size_t)nFiles * 4 + 15) & ~(size_t)15;
    const int wlds = KARGD(a.wldsBytes);
    unsigned int *hits = (unsigned int *)smem;
    unsigned short *sl = (unsigned short *)(smem + hitBytes + (size_t)wid * (size_t)wlds);
    unsigned int *hist = (unsigned int *)(sl + IGD_D_SL);
    unsigned short *sb = (unsigned short *)(hist + IGD_D_H);
    int32_t *sBase = (int32_t *)(smem + hitBytes + (size_t)wavesPerWG * (size_t)wlds);   // the two per-contig tables every query looks up
    int32_t *sNTile = sBase + nCtg;
    // What every batch owes its caller and the NEXT batch (k_query_bounds does it for the other steps): hits[] cleared under
    // IGD_HIP_FLAG_ZERO_FIRST -- nothing is added to it before this kernel has ended -- and the other parity's list counters.
    if (blockIdx.x == 0) {
        u64 *zh = KARGD(a.zeroHits), *zt = KARGD(a.zeroTotal);
        if (zh) for (int f = threadIdx.x; f < nFiles; f += IGD_WG_DIR) zh[f] = 0;
        if (threadIdx.x == 0) {
            if (zt) *zt = 0;
            int32_t *ctl = KARGD(a.ctl);
            ctl[CTL_NLONG + ((a.epoch + 1) & 1)] = 0;
            ctl[CTL_NFIX + ((a.epoch + 1) & 1)] = 0;
            ctl[CTL_BUDGET + ((a.epoch + 1) & 1)] = 0;
            ctl[CTL_NHEAVY + ((a.epoch + 1) & 1)] = 0;
            ctl[CTL_NHEAVYS + ((a.epoch + 1) & 1)] = 0;
            ctl[CTL_NFAR + ((a.epoch + 1) & 1)] = 0;
        }
    }
    for (int f = threadIdx.x; f < nFiles; f += IGD_WG_DIR) hits[f] = 0u;
    for (int k = lane; k < IGD_D_SL; k += IGD_WAVE) sl[k] = 65535;
    for (int k = lane; k < IGD_D_H; k += IGD_WAVE) hist[k] = 0u;
    {
        const int32_t *cb = KARGD(db.ctgBase), *cn = KARGD(db.ctgNTile);
        for (int c = threadIdx.x; c < nCtg; c += IGD_WG_DIR) { sBase[c] = cb[c]; sNTile[c] = cn[c]; }
    }
    __syncthreads();
    const int gwave = (int)blockIdx.x * wavesPerWG + wid;
    const long long nwaves = (long long)gridDim.x * wavesPerWG;
    const int sh = db.shift, nT = db.nT;
    // (the segment list lives in the array of query starts, which no unit is using while a pass is being cut up; the running
    // item counts are needed while units are counted: an area of their own behind it)
    int32_t *segG = (int32_t *)sb;                        // [64] a segment's tile
    int32_t *segP = segG + IGD_C_SEGS;                    // [65] a segment's first query (relative to the pass), closed by the pass's end
    int32_t *segI = (int32_t *)(sb + KARGD(a.sbCap));     // [64] running number of items up to and including a segment
    const unsigned long long below = (1ull << lane) - 1ull;
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void *)KARGD(a.q_ichr), 0, a.nq * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void *)a.q_qs, 0, a.nq * 4, 0x00020000);
    const int vo4 = lane * 4;
    auto key_of = [&](int c, int s, bool &ok) -> int {
        const bool cOk = (unsigned)c < (unsigned)nCtg;
        const int cb = cOk ? sBase[c] : 0, cm = cOk ? sNTile[c] - 1 : -1;
        const int n1r = tile_shift(s, sh);
        const int n1c = n1r < 0 ? 0 : (n1r > cm ? cm : n1r);
        ok = cOk && cm >= 0;                              // (a contig without tiles holds nothing: :462)
        return c < 0 ? -1 : (c >= nCtg ? nT : cb + n1c);
    };
    // ---- 0. the wave's range, handed over at tile boundaries ----
    // Nominally [nq * w / nwaves, nq * (w + 1) / nwaves).  A tile cut by such a border would be counted in two parts -- twice the
    // per-unit work (staging, prefix sums, the bisections of term B) for the same records -- so a border moves forward to the
    // first query of the next tile when one begins within IGD_C_HAND queries: handover(q) is a function of the queries around q
    // alone, and the two waves on either side of a border work it out alike.
    auto handover = [&](int q) -> int {
        if (q <= 0) return 0;
        if (q >= a.nq) return a.nq;
        int C[IGD_C_HAND / IGD_WAVE], S[IGD_C_HAND / IGD_WAVE];
#pragma unroll
        for (int p = 0; p < IGD_C_HAND / IGD_WAVE; p++) {
            C[p] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsC, vo4, (q + p * IGD_WAVE) * 4, 0);
            S[p] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsS, vo4, (q + p * IGD_WAVE) * 4, 0);
        }
        const int pc = (int)__builtin_amdgcn_raw_buffer_load_b32(rsC, 0, (q - 1) * 4, 0);
        const int ps = (int)__builtin_amdgcn_raw_buffer_load_b32(rsS, 0, (q - 1) * 4, 0);
        bool ok;
        int carryG = key_of(pc, ps, ok);
        carryG = __builtin_amdgcn_readfirstlane(ok ? carryG : -1);
        int found = -1;
#pragma unroll
        for (int p = 0; p < IGD_C_HAND / IGD_WAVE; p++) {
            if (found >= 0) continue;
            const int k = key_of(C[p], S[p], ok);
            const int g = ok ? k : -1;
            const int pg = __builtin_amdgcn_update_dpp(carryG, g, 0x138, 0xf, 0xf, false);
            carryG = __builtin_amdgcn_readlane(g, IGD_WAVE - 1);
            const unsigned long long m = __ballot(q + p * IGD_WAVE + lane < a.nq && g != pg);
            if (m) found = q + p * IGD_WAVE + __builtin_ctzll(m);
        }
        return found >= 0 ? found : q;
    };
    const int q0 = handover((int)((long long)a.nq * gwave / nwaves)), q1 = handover((int)((long long)a.nq * (gwave + 1) / nwaves));
    DRaw A, B;
    bool appDirty = false;
    bool broken = false;
    for (int cur = q0; cur < q1 && !broken;) {
        const int n = q1 - cur < IGD_C_PASS ? q1 - cur : IGD_C_PASS;
        // ---- 1. keys, order, segments ----
        int nseg = 0, nEff = n;
        {
            int C[IGD_C_BLOCKS], S[IGD_C_BLOCKS];
#pragma unroll
            for (int p = 0; p < IGD_C_BLOCKS; p++) {
                C[p] = 0; S[p] = 0;
                if (p * IGD_WAVE < n) {
                    C[p] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsC, vo4, (cur + p * IGD_WAVE) * 4, 0);
                    S[p] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsS, vo4, (cur + p * IGD_WAVE) * 4, 0);
                }
            }
            // the query before the pass: the seam of the order check (the batch's first query has none)
            int pc = -1, ps = INT_MIN;
            if (cur > 0) {
                pc = (int)__builtin_amdgcn_raw_buffer_load_b32(rsC, 0, (cur - 1) * 4, 0);
                ps = (int)__builtin_amdgcn_raw_buffer_load_b32(rsS, 0, (cur - 1) * 4, 0);
            }
            bool pok;
            int carryK = cur > 0 ? key_of(pc, ps, pok) : INT_MIN, carryS = cur > 0 ? ps : INT_MIN, carryG = INT_MIN, carryC = pc;
            carryK = __builtin_amdgcn_readfirstlane(carryK); carryS = __builtin_amdgcn_readfirstlane(carryS); carryC = __builtin_amdgcn_readfirstlane(carryC);
            bool bad = false;
#pragma unroll
            for (int p = 0; p < IGD_C_BLOCKS; p++) {
                // (no `break`: the loop must unroll completely -- C[] / S[] are registers only then)
                if (p * IGD_WAVE >= nEff || nseg > IGD_C_SEGS) continue;
                const int idx = p * IGD_WAVE + lane;
                const bool valid = idx < nEff;
                bool ok;
                const int k = key_of(C[p], S[p], ok);
                const int g = (ok && valid) ? k : -1;     // the tile whose units count the query (-1: none does)
                // lane i gets lane i - 1's value, lane 0 the last one of the block before (DPP wave_shr:1)
                const int pk = __builtin_amdgcn_update_dpp(carryK, k, 0x138, 0xf, 0xf, false);
                const int pq = __builtin_amdgcn_update_dpp(carryS, S[p], 0x138, 0xf, 0xf, false);
                const int pg = __builtin_amdgcn_update_dpp(carryG, g, 0x138, 0xf, 0xf, false);
                const int pcn = __builtin_amdgcn_update_dpp(carryC, C[p], 0x138, 0xf, 0xf, false);
                // the promise: keys never decrease, and inside one contig's tile neither do the starts (k_query_bounds' rule)
                bad = bad || (valid && (k < pk || (k == pk && C[p] == pcn && S[p] < pq)));
                carryC = __builtin_amdgcn_readlane(C[p], IGD_WAVE - 1);
                carryK = __builtin_amdgcn_readlane(k, IGD_WAVE - 1);
                carryS = __builtin_amdgcn_readlane(S[p], IGD_WAVE - 1);
                carryG = __builtin_amdgcn_readlane(g, IGD_WAVE - 1);
                const bool nb = valid && (idx == 0 || g != pg);
                const unsigned long long m = __ballot(nb);
                const int slot = nseg + __popcll(m & below);
                if (nb && slot < IGD_C_SEGS) { segG[slot] = g; segP[slot] = idx; }
                const int cnt = __popcll(m);
                if (nseg + cnt > IGD_C_SEGS) {            // more tiles than lanes: the pass ends where segment 65 would begin
                    const unsigned long long mc = __ballot(nb && slot == IGD_C_SEGS);
                    nEff = __builtin_amdgcn_readlane(idx, __builtin_ctzll(mc));
                    nseg = IGD_C_SEGS + 1;                // (the blocks behind are skipped; the queries of this block beyond the cut
                    continue;                             // are looked at again by the next pass, the order check included)
                }
                nseg += cnt;
            }
            if (nseg > IGD_C_SEGS) nseg = IGD_C_SEGS;
            if (__ballot(bad)) { d_mark_broken<true>(a, lane); broken = true; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // A pass ends where a tile ends: the window's last segment is cut short by the window (its tile goes on behind it)
            // unless the window ends with the wave's range -- left to the next pass, which starts at its first query.  (A tile
            // with more queries than a window holds is counted window by window: no boundary to wait for.)
            if (nEff == n && cur + n < q1 && nseg >= 2) { nseg--; nEff = segP[nseg]; }
            else if (lane == 0) segP[nseg] = nEff;       // closes the last segment
        }
        if (broken) break;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- 2. lane i = segment i: its tile's units ----
        int sg = -1, sf0 = 0, sc0 = 0;
        if (lane < nseg) {
            sg = segG[lane];
            const int pos = segP[lane], end = segP[lane + 1];
            sf0 = cur + pos; sc0 = end - pos;
        }
        __builtin_amdgcn_wave_barrier();
        const bool okSeg = sg >= 0 && sg < nT;
        int su0 = 0, snu = 0, sAppOff = 0, sAppMeta = 0;
        if (okSeg) {
            const int32_t *tu = KARGD(db.tileUnit0);
            su0 = tu[sg]; snu = tu[sg + 1] - su0;
            const int4 d = KARGD(a.tileD)[sg];
            sAppOff = d.x; sAppMeta = d.y;
        }
        const int incl = wave_inclusive_sum(snu), excl = incl - snu;
        const int total = __builtin_amdgcn_readlane(incl, IGD_WAVE - 1);
        segI[lane] = incl;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int ib = 0; ib < total; ib += IGD_WAVE) {
            // ---- 3. item ib + lane: which segment, which of its units ----
            DRegs L;
            L.offLo = L.n = L.jf = L.f0 = L.c0 = L.appOff = L.appMeta = 0;
            {
                const int k = ib + lane;
                int s = 0;                                // the first segment whose running count exceeds k
#pragma unroll
                for (int step = 32; step > 0; step >>= 1) s += (segI[s + step - 1] <= k) ? step : 0;
                s = s > 63 ? 63 : s;
                const int bp = s * 4;
                const int u0_ = __builtin_amdgcn_ds_bpermute(bp, su0), ex_ = __builtin_amdgcn_ds_bpermute(bp, excl);
                const int f0_ = __builtin_amdgcn_ds_bpermute(bp, sf0), c0_ = __builtin_amdgcn_ds_bpermute(bp, sc0);
                const int ao_ = __builtin_amdgcn_ds_bpermute(bp, sAppOff), am_ = __builtin_amdgcn_ds_bpermute(bp, sAppMeta);
                if (k < total) {
                    const Unit *up = KARGD(db.units) + (u0_ + (k - ex_));
                    const int4 ua = ((const int4 *)up)[0];
                    const int ujf = ((const int32_t *)up)[4];
                    L.offLo = ua.x; L.n = ua.w; L.jf = ujf;
                    L.f0 = f0_; L.c0 = c0_;
                    L.appOff = ao_; L.appMeta = am_;
                    // only a tile's first unit sees its queries when the tile holds no record (its placeholder)
                    if (ua.w == 0 && !(ujf & 1)) L.c0 = 0;
                }
            }
            unsigned long long m = __ballot(L.c0 != 0);
            int ka = -1, kb = -1;
            if (m) { ka = __builtin_ctzll(m); m &= m - 1; }
            if (m) { kb = __builtin_ctzll(m); m &= m - 1; }
            d_issue<USE_V>(db, a, L, ka < 0 ? 0 : ka, ka >= 0, lane, A);
            while (ka >= 0) {
                d_issue<USE_V>(db, a, L, kb < 0 ? 0 : kb, kb >= 0, lane, B);
                d_compute<USE_V, false, true, false>(db, a, L, ka, lane, A, hits, sl, hist, sb, INT_MIN, &appDirty);
                ka = -1;
                if (m) { ka = __builtin_ctzll(m); m &= m - 1; }
                d_issue<USE_V>(db, a, L, ka < 0 ? 0 : ka, ka >= 0, lane, A);
                if (kb >= 0) d_compute<USE_V, false, true, false>(db, a, L, kb, lane, B, hits, sl, hist, sb, INT_MIN, &appDirty);
                kb = -1;
                if (m) { kb = __builtin_ctzll(m); m &= m - 1; }
            }
        }
        __builtin_amdgcn_wave_barrier();
        cur += nEff;
    }
    __syncthreads();
    {
        unsigned int *row32 = (unsigned int *)KARGD(a.out) + (size_t)blockIdx.x * nFiles;
        for (int f = threadIdx.x; f < nFiles; f += IGD_WG_DIR) row32[f] = hits[f];
    }
}

// Can the batch take the query-partitioned step with 32-bit workgroup counters and no run-time guard?  A wave adds at most
// (its queries) x (records of a tile + what rides along with each of its units) to any ONE counter; IGD_WG_DIR / 64 waves share
// the workgroup's counters.
static bool chunks_fit(int64_t nq, int grid, int64_t maxTileRecords)
{
    const int64_t wavesPerWG = IGD_WG_DIR / IGD_WAVE, nwaves = (int64_t)grid * wavesPerWG;
    const int64_t perWave = (nq + nwaves - 1) / nwaves + 1;
    const int64_t perTile = maxTileRecords + (int64_t)IGD_D_APP * ((maxTileRecords + IGD_CHUNK - 1) / IGD_CHUNK + 1);
    return perWave * perTile < ((int64_t)1 << 32) / wavesPerWG;
}
