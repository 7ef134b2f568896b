// engine/scan_chunks.hpp -- part of igd_hip.hip (included there once; not a stand-alone header).
// The DIRECT step as a merge of two ordered streams (round 6): ONE kernel reads every query once -- no pre-pass at all.
// ------------------------------------------------------------------------------------------
// Round 5's DIRECT step was k_query_bounds<.., BONLY> (firstQ[] = the first query of every tile: 8 bytes per query read, 21-24 us
// for 1.25e7 queries) -> igd_scan_direct (units dealt to waves, each unit looks its tile's query range up in firstQ[] and reads
// those queries: 8 more bytes per query) -> k_reduce_slabs.  Here the QUERIES are dealt to the waves instead:
//   * wave w owns a contiguous range of the position-sorted batch -- nominally [nq * w / nwaves, nq * (w + 1) / nwaves), each
//     border moved forward to the first query of the next tile when one begins within IGD_C_HAND queries (handover(): a
//     function of the queries around the border alone, so that both neighbours work it out alike) -- and reads it as a STREAM:
//     blocks of 64 queries (contig, start, end: 12 bytes per query, coalesced), the next block always in flight;
//   * the first query of the stream names a tile (src/igd_search.c:459-464: first tile, clamped into the contig); the tile's
//     first unit is counted by d_compute<.., STREAM> (scan_direct.hpp) against the RUN of queries that belong to the tile --
//     found while they are counted: the run ends at the first query of another tile, at the wave's border or after
//     IGD_C_PASS queries (the LDS array of starts; the tile then simply goes on as the next item) -- the rank method over the
//     unit's records, the next tile's first records riding along, the exceptions listed for the exact walks, as in round 5;
//     a tile's further units (> 320 records) are counted against the now known range [f0, f0 + c0) by the round-5 code;
//   * a tile's records are in flight before its turn comes: the item after (tile g, unit r) is (g, r + 1) or -- what a dense
//     batch makes true nearly always -- (g + 1, 0); a wrong guess (a tile without queries) costs a round trip.  The per-tile
//     descriptors come from a 32-tile window of tileD[] in the wave's LDS.
// Why the sums are right: per record, hits = #{q: start < qe} - #{q: end <= qs} over the queries of a tile is additive over ANY
// partition of those queries (src/igd_search.c:479-493: the reference counts query by query), so a tile cut by a border or
// by IGD_C_PASS is counted in parts; what a tile's first unit does per query -- listing it for a walk of its later tiles,
// pushing it into the next tile's first records -- is per query as well.
// The order promise is verified on the way: keys of consecutive runs never decrease (a wave's first run against the query
// before its range), starts inside a run never decrease, every query of a run has the run's contig and tile.
// What goes away with the tile-keyed ranges: firstQ[], the bounds pass and its launch, the heavy-tile slices (a tile with 10^6
// queries is 10^6 / 960 items spread over all waves) and every add that is not to a workgroup's own LDS counters: a batch found
// out of order by ANY wave has added nothing when k_reduce_slabs looks at the mark.
// 32-bit counters: the host only takes this step when (queries per wave) x (records of the fullest tile + what rides along)
// stays below a wave's share of 2^32 (chunks_fit): no run-time guard, no far list.
// HBM bytes: 12 per query (contig, start, end: once) + 6 per record of a visited unit (+ the <= 64 appended) + 16 per tile.

#define IGD_C_HAND 512                                  // a wave's border moves to the next tile's first query when that lies within this many queries
#define IGD_C_WIN 32                                    // tiles of tileD[] a wave keeps in LDS

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
    int32_t *sBase = (int32_t *)(smem + hitBytes + (size_t)wavesPerWG * (size_t)wlds);   // the two per-contig tables every run's first query looks up
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
    const int sh = db.shift, nT = db.nT, nq = a.nq;
    CStream st;
    st.rsC = __builtin_amdgcn_make_buffer_rsrc((void *)KARGD(a.q_ichr), 0, nq * 4, 0x00020000);
    st.rsS = __builtin_amdgcn_make_buffer_rsrc((void *)a.q_qs, 0, nq * 4, 0x00020000);
    st.rsE = __builtin_amdgcn_make_buffer_rsrc((void *)a.q_qe, 0, nq * 4, 0x00020000);
    const int vo4 = lane * 4;
    // key of a query: the global number of its first tile, clamped into its contig (k_query_bounds' BONLY form: contig numbers
    // outside the database get -1 / nT); ok: a tile owns the query
    auto key_of = [&](int c, int s, bool &ok) -> int {
        const bool cOk = (unsigned)c < (unsigned)nCtg;
        const int cb = cOk ? sBase[c] : 0, cm = cOk ? sNTile[c] - 1 : -1;
        const int n1r = tile_shift(s, sh);
        const int n1c = n1r < 0 ? 0 : (n1r > cm ? cm : n1r);
        ok = cOk && cm >= 0;                              // (a contig without tiles holds nothing: :462)
        return c < 0 ? -1 : (c >= nCtg ? nT : cb + n1c);
    };
    // ---- the wave's range, handed over at tile boundaries ----
    // A tile cut by a border would be counted in two parts -- twice the per-unit work (staging, prefix sums, the bisections of
    // term B) for the same records -- so a border moves forward to the first query of the next tile when one begins within
    // IGD_C_HAND queries.  Both borders' windows are asked for at once.
    const int n0 = (int)((long long)nq * gwave / nwaves), n1 = (int)((long long)nq * (gwave + 1) / nwaves);
    int q0, q1;
    {
        constexpr int HB = IGD_C_HAND / IGD_WAVE;
        int C0[HB], S0[HB], C1[HB], S1[HB];
#pragma unroll
        for (int p = 0; p < HB; p++) {
            C0[p] = (int)__builtin_amdgcn_raw_buffer_load_b32(st.rsC, vo4, (n0 + p * IGD_WAVE) * 4, 0);
            S0[p] = (int)__builtin_amdgcn_raw_buffer_load_b32(st.rsS, vo4, (n0 + p * IGD_WAVE) * 4, 0);
            C1[p] = (int)__builtin_amdgcn_raw_buffer_load_b32(st.rsC, vo4, (n1 + p * IGD_WAVE) * 4, 0);
            S1[p] = (int)__builtin_amdgcn_raw_buffer_load_b32(st.rsS, vo4, (n1 + p * IGD_WAVE) * 4, 0);
        }
        // (the query before a border; a border at 0 or nq reads nothing it uses)
        const int pc0 = (int)__builtin_amdgcn_raw_buffer_load_b32(st.rsC, 0, (n0 > 0 ? n0 - 1 : 0) * 4, 0);
        const int ps0 = (int)__builtin_amdgcn_raw_buffer_load_b32(st.rsS, 0, (n0 > 0 ? n0 - 1 : 0) * 4, 0);
        const int pc1 = (int)__builtin_amdgcn_raw_buffer_load_b32(st.rsC, 0, (n1 > 0 ? n1 - 1 : 0) * 4, 0);
        const int ps1 = (int)__builtin_amdgcn_raw_buffer_load_b32(st.rsS, 0, (n1 > 0 ? n1 - 1 : 0) * 4, 0);
        auto handover = [&](int q, int pc, int ps, const int (&C)[HB], const int (&S)[HB]) -> int {
            if (q <= 0) return 0;
            if (q >= nq) return nq;
            bool ok;
            int carryG = key_of(pc, ps, ok);
            carryG = __builtin_amdgcn_readfirstlane(ok ? carryG : -1);
            int found = -1;
#pragma unroll
            for (int p = 0; p < HB; p++) {
                const int k = key_of(C[p], S[p], ok);
                const int g = ok ? k : -1;
                const int pg = __builtin_amdgcn_update_dpp(carryG, g, 0x138, 0xf, 0xf, false);
                carryG = __builtin_amdgcn_readlane(g, IGD_WAVE - 1);
                const unsigned long long m = __ballot(q + p * IGD_WAVE + lane < nq && g != pg);
                if (found < 0 && m) found = q + p * IGD_WAVE + __builtin_ctzll(m);
            }
            return found >= 0 ? found : q;
        };
        q0 = handover(n0, pc0, ps0, C0, S0);
        q1 = handover(n1, pc1, ps1, C1, S1);
    }
    if (q0 < q1) {
        // ---- the stream ----
        st.bpos = q0;
        st.C = (int)__builtin_amdgcn_raw_buffer_load_b32(st.rsC, vo4, q0 * 4, 0);
        st.S = (int)__builtin_amdgcn_raw_buffer_load_b32(st.rsS, vo4, q0 * 4, 0);
        st.E = (int)__builtin_amdgcn_raw_buffer_load_b32(st.rsE, vo4, q0 * 4, 0);
        c_stream_issue(st, lane);
        int lastKey = INT_MIN, lastS = INT_MIN;           // key and start of the query before the next run (at first: before the range -- the order check's seam)
        if (q0 > 0) {
            const int pc = (int)__builtin_amdgcn_raw_buffer_load_b32(st.rsC, 0, (q0 - 1) * 4, 0);
            const int ps = (int)__builtin_amdgcn_raw_buffer_load_b32(st.rsS, 0, (q0 - 1) * 4, 0);
            bool ok;
            lastKey = __builtin_amdgcn_readfirstlane(key_of(pc, ps, ok));
            lastS = __builtin_amdgcn_readfirstlane(ok ? ps : INT_MIN);
        }
        int off = 0;                                      // lane of the current block at which the next run begins
        // a window of IGD_C_WIN tiles of tileD[] in the wave's LDS (behind its array of query starts): tile wbase + i at win[i]
        int wbase = -(1 << 29);                          // (no window yet; g - wbase stays an int)
        int4 *win = (int4 *)(sb + KARGD(a.sbCap));
        const int4 *tileD = KARGD(a.tileD);
        auto window = [&](int g) {                        // (a round trip: once per IGD_C_WIN - 1 tiles)
            if (g >= wbase && g < wbase + IGD_C_WIN - 1) return;
            wbase = g;
            if (lane < IGD_C_WIN) win[lane] = (g + lane < nT) ? tileD[g + lane] : make_int4(0, 0, 0, 0);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };
        // one unit of tile g in the form d_compute takes it (the same unit in every lane, read with lane 0)
        auto unit_regs = [&](int g, int r, DRegs &L) {
            const int4 d = win[g - wbase];
            const int ao = __builtin_amdgcn_readfirstlane(d.x), am = __builtin_amdgcn_readfirstlane(d.y);
            const int of = __builtin_amdgcn_readfirstlane(d.z), cn = __builtin_amdgcn_readfirstlane(d.w);
            const int left = cn - r * IGD_CHUNK;
            L.offLo = of + r * IGD_CHUNK;
            L.n = left < IGD_CHUNK ? (left > 0 ? left : 0) : IGD_CHUNK;
            L.jf = ((g - sBase[(am >> 13) & 1023]) << 4) | (r == 0 ? 1 : 0);
            L.appOff = ao; L.appMeta = am;
            L.f0 = 0; L.c0 = 1;
        };
        // the record words of a unit, asked for ahead of its turn (no queries: the stream brings them) ...
        auto issue = [&](int g, int r, uint32_t (&W)[IGD_SLOTS + 1]) {
            DRegs L;
            unit_regs(g, r, L);
            const int n = L.n;
            const int appN = (L.jf & 1) ? (L.appMeta & 127) : 0;
            const unsigned offLo = (unsigned)L.offLo, appOff = (unsigned)L.appOff;
            const int end = (int)offLo + n, endA = (int)appOff + appN;
            const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)db.pse, 0, (int)((unsigned)end * 4u), 0x00020000);
            const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void *)db.pse, 0, appN ? (int)((unsigned)endA * 4u) : 0, 0x00020000);
#pragma unroll
            for (int q = 0; q < IGD_SLOTS; q++) W[q] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsA, vo4 + q * 256, (int)(offLo * 4u), 0);
            W[IGD_SLOTS] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsB, vo4, (int)(appOff * 4u), 0);
        };
        // ... and its dataset numbers at its turn: they are first looked at when the unit's counts are flushed, behind all its
        // batches (six registers that are not held while the unit waits)
        auto take = [&](int g, int r, const uint32_t (&W)[IGD_SLOTS + 1], DRegs &L, DRaw &R) {
            unit_regs(g, r, L);
            const int appN = (L.jf & 1) ? (L.appMeta & 127) : 0;
#pragma unroll
            for (int q = 0; q <= IGD_SLOTS; q++) R.a[q] = W[q];
            R.c0 = 1; R.f0 = 0; R.n = L.n; R.qs = 0; R.qe = 0;
            d_load_x<USE_V>(db, R.x, (unsigned)L.offLo, L.n, (unsigned)L.appOff, appN, lane);
        };
        DRaw A;
        DRegs LA;
        uint32_t WB[IGD_SLOTS + 1];                       // the record words of the unit expected next (in flight)
#pragma unroll
        for (int q = 0; q <= IGD_SLOTS; q++) WB[q] = 0u;
        int gB = -2, rB = 0;                              // ... which unit that is
        bool appDirty = false, broken = false;
        for (;;) {
            // ---- the next run: where the stream stands ----
            if (off >= IGD_WAVE) { c_stream_advance(st, lane); off = 0; }
            const int cur = st.bpos + off;
            if (cur >= q1) break;
            const int c_ = __builtin_amdgcn_readlane(st.C, off), s_ = __builtin_amdgcn_readlane(st.S, off);
            bool ok;
            const int k = key_of(c_, s_, ok);
            // keys never decrease; k == lastKey: a tile that goes on (behind a border or IGD_C_PASS queries) -- its starts go on too
            if (k < lastKey || (ok && k == lastKey && s_ < lastS)) { broken = true; break; }
            lastKey = k;
            if (!ok) {
                // a run no tile owns (contig outside the database, contig without tiles): its keys must not decrease either
                int carryK = k;
                bool cont = false;                        // (behind the run's first block lane 0 is compared with the block before)
                for (;;) {
                    bool okl;
                    const int kl = key_of(st.C, st.S, okl);
                    const int pk = __builtin_amdgcn_update_dpp(carryK, kl, 0x138, 0xf, 0xf, false);
                    const unsigned long long stop = __ballot(lane >= off && (okl || st.bpos + lane >= q1));
                    const int e = stop ? __builtin_ctzll(stop) : IGD_WAVE;
                    if (__ballot(lane >= off && lane < e && (lane > off || cont) && kl < pk)) broken = true;
                    if (e > off) lastKey = __builtin_amdgcn_readlane(kl, e - 1);
                    if (e < IGD_WAVE) { off = e; break; }
                    carryK = __builtin_amdgcn_readlane(kl, IGD_WAVE - 1);
                    c_stream_advance(st, lane);
                    off = 0;
                    cont = true;
                }
                lastS = INT_MIN;
                if (broken) break;
                continue;
            }
            // ---- tile k: its first unit against the run, its further units against the range the run turns out to be ----
            const int g = k;
            window(g);
            if (!(gB == g && rB == 0)) issue(g, 0, WB);   // (a wrong guess, or the wave's first tile: a round trip)
            take(g, 0, WB, LA, A);
            const int cnt = __builtin_amdgcn_readfirstlane(win[g - wbase].w);
            const int nu = cnt > 0 ? (cnt + IGD_CHUNK - 1) / IGD_CHUNK : 1;
            // what comes next is asked for now: the tile's second unit, or the next tile's first
            {
                const int gN = nu > 1 ? g : g + 1, rN = nu > 1 ? 1 : 0;
                gB = -2;
                if (gN < nT && gN - wbase < IGD_C_WIN) { issue(gN, rN, WB); gB = gN; rB = rN; }
            }
            const int meta = LA.appMeta;
            const int ctg = (meta >> 13) & 1023;
            const int j = LA.jf >> 4, cm = sNTile[ctg] - 1;
            const int f0 = cur;
            int offOut = off;
            const int c0 = d_compute<USE_V, false, true, false, true>(db, a, LA, 0, lane, A, hits, sl, hist, sb, INT_MIN, &appDirty,
                                                                      &st, off, q1, ctg, j == 0 ? INT_MIN : j, j == cm ? INT_MAX : j, &offOut, &lastS);
            off = offOut;
            if (c0 <= 0) { broken = true; break; }        // (cannot happen: the run's first query belongs to its tile)
            for (int r = 1; r < nu; r++) {
                // unit r was asked for while unit r - 1 was counted; behind it comes unit r + 1 or the next tile
                if (!(gB == g && rB == r)) issue(g, r, WB);
                take(g, r, WB, LA, A);
                {
                    const int gN = r + 1 < nu ? g : g + 1, rN = r + 1 < nu ? r + 1 : 0;
                    gB = -2;
                    if (gN < nT && gN - wbase < IGD_C_WIN) { issue(gN, rN, WB); gB = gN; rB = rN; }
                }
                LA.f0 = f0; LA.c0 = c0;
                A.f0 = f0; A.c0 = c0;
                // (the first 64 queries of the range: d_compute's first batch)
                A.qs = (int)__builtin_amdgcn_raw_buffer_load_b32(st.rsS, vo4, f0 * 4, 0);
                A.qe = (int)__builtin_amdgcn_raw_buffer_load_b32(st.rsE, vo4, f0 * 4, 0);
                d_compute<USE_V, false, true, false>(db, a, LA, 0, lane, A, hits, sl, hist, sb, INT_MIN, &appDirty);
            }
        }
        if (broken) d_mark_broken<true>(a, lane);
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
    const int64_t perWave = (nq + nwaves - 1) / nwaves + 1 + IGD_C_HAND;
    const int64_t perTile = maxTileRecords + (int64_t)IGD_D_APP * ((maxTileRecords + IGD_CHUNK - 1) / IGD_CHUNK + 1);
    return perWave * perTile < ((int64_t)1 << 32) / wavesPerWG;
}
