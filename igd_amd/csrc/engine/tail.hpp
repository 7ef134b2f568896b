// engine/tail.hpp -- part of igd_hip.hip (included there once; not a stand-alone header).
// exact walks, coverage, k_reduce_slabs (last launch of a batch), k_sum_hits
// ------------------------------------------------------------------------------------------
// exact_walk_body: what the scan kernel leaves out (the exact-walk list of the batch's path): a wave walks a listed
// query's tiles (WALK_*: which of them) on the EXACT arrays, 5 slots at a time, and adds into the workgroup's LDS
// counters of the batch's last launch -- the caller's global hits[] where the files do not fit -- and the batch total.
// Rare for the benchmark's queries; a batch of long ones lists every query (its last tile).
// The workgroup's LDS counters of the batch's last launch.  An LDS pointer by TYPE: through a plain `u64 *` (which may also
// be null, or point at global memory) the compiler emitted flat_atomic_add_x2 -- 138 of them in k_reduce_slabs, not one
// ds_add_u64 -- and a flat atomic that resolves to LDS goes the vector-memory way round (address check in the texture
// path, both counters) instead of straight to the LDS: the long queries' last launch spent most of its time there.
struct TailHist {
    igd_lds_u64 *p;
    bool on;
    __device__ __forceinline__ void add(int ix, u64 v) const { (void)__hip_atomic_fetch_add(p + ix, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
};

template <bool USE_V>
__device__ __forceinline__ void exact_walk_body(const DbView &db, const ScanArgs &a, const int2 *__restrict__ fixList,
                                                const int2 *__restrict__ longList, int gwave, int nwaves, int ctlv, const TailHist hist)
{
    // ctlv: the batch's control words, word i in lane i (ONE load by the caller: the walk and the two skew valves
    // would otherwise each wait for their own, one after the other, to find out that there is nothing to do)
    const bool uns = __builtin_amdgcn_readlane(ctlv, CTL_UNSORTED) == a.epoch;
    if (a.mode == 1 && uns) return;                      // broken promise: the batch adds nothing
    const bool sortedPath = a.mode == 1 || (a.mode == 0 && !uns);
    const int2 *list = sortedPath ? fixList : longList;
    const int nList = __builtin_amdgcn_readlane(ctlv, (sortedPath ? CTL_NFIX : CTL_NLONG) + (a.epoch & 1));
    const int lane = threadIdx.x & 63;
    u64 found = 0;
    // a wave takes `per` entries of the list at a time (one each while the list is shorter than the launch has waves): every
    // lane reads one entry's query and contig first, so that the chain of dependent loads is paid once per group, not per query
    int per = (nList + nwaves - 1) / nwaves;
    per = per > IGD_WAVE ? IGD_WAVE : per;
    for (int l0 = gwave * per; l0 < nList; l0 += nwaves * per) {
        const int cnt = nList - l0 < per ? nList - l0 : per;
        int vq = 0, vkind = -1, vqs = 0, vqe = 0, vn1 = 0, vj1 = -1, vbase = 0, vcnt = 0, voffLo = 0, voffHi = 0, vlob = 0;
        int vu0 = 0, vnu = 0, vqe2 = 0;                  // WALK_LAST over the compact image: the last tile's units, the query's qe'
        UnitRegs vur;
        vur.offLo = vur.offHi = vur.tile = vur.n = vur.jf = 0;
#pragma unroll
        for (int r = 0; r < 6; r++) vur.w[r] = 0;
        const bool cimg = USE_V ? a.packedWalk == 2 : a.packedWalk != 0;
        if (lane < cnt) {
            const int2 ent = list[l0 + lane];
            vq = ent.x; vkind = ent.y & 15;
            vqs = a.q_qs[vq]; vqe = a.q_qe[vq];
            const int cc = sortedPath ? ent.y >> 4 : a.q_ichr[vq];     // (k_query_bounds' entries carry the contig: a batch given as runs has no ichr[])
            vn1 = tile_of(db, (db.vshift >= 0 && vqs < 0) ? 0 : vqs);     // (re-tiled copy: see query_span)
            int n2 = tile_of(db, (int)((unsigned)vqe - 1u));
            const int mT = db.ctgNTile[cc] - 1;
            if (n2 > mT) n2 = mT;
            vbase = db.ctgBase[cc];
            vj1 = n2 > vn1 ? n2 : vn1;
            if (a.rule == IGD_HIP_RULE_NEST && db.tileCnt[vbase + vn1] == 0) vj1 = -1;   // :468 -- nothing to walk
            if (vj1 >= 0) {                              // ... and the first tile of the walk (the only one of WALK_LAST and WALK_FIRST)
                const int jf = vkind == WALK_LAST ? vj1 : vkind == WALK_REST ? vn1 + 1 : vn1;
                if (jf <= vj1 && cimg && vkind == WALK_LAST) {
                    vu0 = db.tileUnit0[vbase + jf]; vnu = db.tileUnit0[vbase + jf + 1] - vu0;
                    vcnt = vnu;                          // (0: an empty tile)
                    if (vnu > 0) vur = load_unit_regs(db.units + vu0);     // the tile's first unit (mostly its only one)
                    const int T0 = (int)((unsigned)jf * (unsigned)db.nbp);
                    vqe2 = vqe - T0;
                    vqe2 = (vqe2 < db.nbp ? vqe2 : db.nbp) + 1;
                } else
                if (jf <= vj1) {
                    vcnt = db.tileCnt[vbase + jf];
                    const int64_t o = db.tileOff[vbase + jf];
                    voffLo = (int)o; voffHi = (int)(o >> 32);
                    vlob = db.tileBd[vbase + jf];
                }
            }
        }
        if (cimg) {
            // The LAST tile of the group's long queries over the COMPACT image.  Such a query covers the tile from its start up
            // to qe, so a record counts iff it starts in the tile (s' >= 1: the copy that counts, :510-511) before qe (s' < qe');
            // its end is beyond the query's start by construction.  Records are ordered by start and every 64-record slot's
            // smallest s' is in the unit's descriptor -- fetched one per lane with the group's other look-ups -- so the slots at or
            // beyond qe' are never loaded (on average half the tile) and a record is 6 bytes (12 in the exact arrays).  What
            // bounds a walk is the round trip for its records: TWO walks are in flight per wave.
            struct Walk { uint32_t pa[IGD_SLOTS], px[IGD_SLOTS]; int qe2, live; };
            auto issue = [&](int e, Walk &w) {
                w.qe2 = __builtin_amdgcn_readlane(vqe2, e);
                const int n = __builtin_amdgcn_readlane(vur.n, e);
                const int64_t off = (int64_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(vur.offHi, e) << 32) |
                                              (unsigned)__builtin_amdgcn_readlane(vur.offLo, e));
                // Records are ordered by start, so the slots that can hold a hit are the first nOn.  Their loads are bounds-checked
                // buffer loads over [0, min(n, 64 nOn)): a lane beyond reads 0 (s' = 65535: beyond every qe') and touches no memory, and
                // -- the point -- no load sits behind a branch: behind `if (i < n)` the compiler waited for each slot's words before
                // it asked for the next slot's (s_waitcnt after every pair of loads), five round trips per walk where one will do.
                int nOn = 0;
#pragma unroll
                for (int r = 0; r < IGD_SLOTS; r++)
                    nOn += (r * IGD_WAVE < n && (int)(65535u - ((unsigned)__builtin_amdgcn_readlane(vur.w[r], e) & 0xFFFFu)) < w.qe2) ? 1 : 0;   // wave-uniform; monotone
                w.live = (1 << nOn) - 1;
                const int m = n < nOn * IGD_WAVE ? n : nOn * IGD_WAVE;
                const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)(db.pse + off), 0, m * 4, 0x00020000);
                if (USE_V) {
                    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void *)(db.pxv + off), 0, m * 4, 0x00020000);
#pragma unroll
                    for (int r = 0; r < IGD_SLOTS; r++) {
                        w.pa[r] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsA, lane * 4, r * 256, 0);
                        w.px[r] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsX, lane * 4, r * 256, 0);
                    }
                } else {
                    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void *)(db.px + off), 0, m * 2, 0x00020000);
#pragma unroll
                    for (int r = 0; r < IGD_SLOTS; r++) {
                        w.pa[r] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsA, lane * 4, r * 256, 0);
                        w.px[r] = (uint32_t)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsX, lane * 2, r * 128, 0);
                    }
                }
            };
            auto count = [&](const Walk &w) {
#pragma unroll
                for (int r = 0; r < IGD_SLOTS; r++) {
                    if (!(w.live & (1 << r))) continue;
                    const int s2 = (int)(65535u - (w.pa[r] & 0xFFFFu));
                    bool hit = s2 >= 1 && s2 < w.qe2;
                    if (USE_V) hit = hit && ((int)w.px[r] >> 16) >= a.v;
                    const int ix = (int)(w.px[r] & 0xFFFFu);
                    found += __popcll(__ballot(hit));
                    if (hit) { if (hist.on) hist.add(ix, 1ull); else atomicAdd(&a.out[ix], 1ull); }
                }
            };
            unsigned long long m = __ballot(lane < cnt && vkind == WALK_LAST && vj1 >= 0 && vnu > 0);
            // the further chunks of a tile of more than 320 records (rare): one after the other
            auto more = [&](int e) {
                for (int u = __builtin_amdgcn_readlane(vu0, e) + 1, ue = __builtin_amdgcn_readlane(vu0, e) + __builtin_amdgcn_readlane(vnu, e); u < ue; u++) {
                    const UnitRegs ur = load_unit_regs(db.units + u);          // the same unit in every lane
                    const int qe2 = __builtin_amdgcn_readlane(vqe2, e);
                    if ((int)(65535u - ((unsigned)__builtin_amdgcn_readfirstlane(ur.w[0]) & 0xFFFFu)) >= qe2) break;   // sorted: nothing here or behind
                    const int n = __builtin_amdgcn_readfirstlane(ur.n);
                    const int64_t off = (int64_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(ur.offHi) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(ur.offLo));
                    Walk C;
                    C.qe2 = qe2; C.live = 0;
#pragma unroll
                    for (int r = 0; r < IGD_SLOTS; r++) {
                        const int i = r * IGD_WAVE + lane;
                        C.pa[r] = 0xFFFFu; C.px[r] = 0u;
                        if (r * IGD_WAVE < n) {
                            C.live |= 1 << r;
                            const int64_t at = off + (i < n ? i : 0);
                            const uint32_t p0 = db.pse[at], x0 = USE_V ? db.pxv[at] : (uint32_t)db.px[at];
                            if (i < n) { C.pa[r] = p0; C.px[r] = x0; }
                        }
                    }
                    count(C);
                }
            };
            // FOUR walks in flight per wave (the tail's long-query work runs on a quarter of the launch's workgroups, see batch_tail)
            Walk W0, W1, W2, W3;
            int e0 = -1, e1 = -1, e2 = -1, e3 = -1;
#define IGD_WALK_NEXT(E, W) do { E = -1; if (m) { E = __builtin_ctzll(m); m &= m - 1; issue(E, W); } } while (0)
            IGD_WALK_NEXT(e0, W0); IGD_WALK_NEXT(e1, W1); IGD_WALK_NEXT(e2, W2); IGD_WALK_NEXT(e3, W3);
            while (e0 >= 0) {
                count(W0); more(e0); IGD_WALK_NEXT(e0, W0);
                if (e1 >= 0) { count(W1); more(e1); IGD_WALK_NEXT(e1, W1); }
                if (e2 >= 0) { count(W2); more(e2); IGD_WALK_NEXT(e2, W2); }
                if (e3 >= 0) { count(W3); more(e3); IGD_WALK_NEXT(e3, W3); }
                if (e0 < 0) {                              // slot 0 ran dry first: the others hold what is left
                    if (e1 >= 0) { count(W1); more(e1); e1 = -1; }
                    if (e2 >= 0) { count(W2); more(e2); e2 = -1; }
                    if (e3 >= 0) { count(W3); more(e3); e3 = -1; }
                }
            }
#undef IGD_WALK_NEXT
        }
        for (int e = 0; e < cnt; e++) {
            const int kind = __builtin_amdgcn_readlane(vkind, e), qs = __builtin_amdgcn_readlane(vqs, e), qe = __builtin_amdgcn_readlane(vqe, e);
            const int n1 = __builtin_amdgcn_readlane(vn1, e), base = __builtin_amdgcn_readlane(vbase, e);
            int j1 = __builtin_amdgcn_readlane(vj1, e);
            if (j1 < 0) continue;
            if (kind == WALK_LAST && cimg) continue;       // (done above, over the compact image)
            int j0 = n1;
            if (kind == WALK_LAST) j0 = j1;              // (n2 >= n1 + IGD_SHORT_TILES: the tiles between are counted by coverage_body)
            if (kind == WALK_FIRST) j1 = n1;
            if (kind == WALK_REST) { j0 = n1 + 1; j1 = j1 < n1 + IGD_SHORT_TILES - 1 ? j1 : n1 + IGD_SHORT_TILES - 1; }   // (igd_scan_direct: the later tiles the scan leaves out; a long query's tiles from n1+4 on: coverage + WALK_LAST)
            for (int j = j0; j <= j1; j = (kind == WALK_ALL && j < j1) ? j1 : j + 1) {   // (WALK_ALL: first and last tile, coverage_body has the rest)
                const int t = base + j;
                const int tcnt = j == j0 ? __builtin_amdgcn_readlane(vcnt, e) : db.tileCnt[t];
                if (tcnt == 0) continue;
                const int lob = (j == n1) ? INT_MIN : j == j0 ? __builtin_amdgcn_readlane(vlob, e) : db.tileBd[t];
                const int64_t toff = j == j0 ? (int64_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(voffHi, e) << 32) |
                                                         (unsigned)__builtin_amdgcn_readlane(voffLo, e))
                                             : db.tileOff[t];
                for (int rec0 = 0; rec0 < tcnt; rec0 += IGD_CHUNK) {
                    int st[IGD_SLOTS], en[IGD_SLOTS], ix[IGD_SLOTS], va[IGD_SLOTS];
#pragma unroll
                    for (int r = 0; r < IGD_SLOTS; r++) {
                        const int i = rec0 + r * IGD_WAVE + lane;
                        const bool ok = i < tcnt;
                        const int64_t at = toff + (ok ? i : 0);      // (every lane loads -- a lane beyond the tile its first record: no load behind a branch, see `issue` above)
                        const int s0 = db.start[at], e0 = db.end[at], x0 = db.idx[at];
                        st[r] = ok ? s0 : INT_MAX;
                        en[r] = ok ? e0 : INT_MIN;
                        ix[r] = ok ? x0 : 0;
                        if (USE_V) { const int v0 = db.value[at]; va[r] = ok ? v0 : INT_MIN; }
                    }
                    if (__builtin_amdgcn_readfirstlane(st[0]) >= qe) break;    // sorted: nothing further
#pragma unroll
                    for (int r = 0; r < IGD_SLOTS; r++) {
                        bool hit = (st[r] < qe) & (st[r] >= lob) & (en[r] > qs);
                        if (USE_V) hit = hit & (va[r] >= a.v);
                        found += __popcll(__ballot(hit));
                        if (hit) { if (hist.on) hist.add(ix[r], 1ull); else atomicAdd(&a.out[ix[r]], 1ull); }
                    }
                }
            }
        }
    }
    if (a.total && lane == 0 && found) atomicAdd(a.total, found);
}

// coverage_body: the tiles that long queries of a merge-join batch cover from end to end (IGD_COV_*).  Every wave takes
// a contiguous run of units, finds how many long queries cover its first tile (coarse sums + the fine differences of
// the tile's block) and keeps that count running from tile to tile; a unit of a covered tile adds count x 1 to hits[]
// for each of its records that starts in the tile (and passes the value filter).  The same launch zeroes the
// difference arrays the batch BEFORE this one used (its own last launch is done with them).
// (the reset is its own step: a batch that breaks its promise of order still has to clean up after the one before it)
__device__ __forceinline__ void coverage_reset(const DbView &db, int epoch, int gwave, int nwaves, int ctlv)
{
    const int other = (epoch & 1) ^ 1;
    const size_t covLen = IGD_COV_LEN(db.nT);
    for (int set = 0; set < 2; set++) {
        if (__builtin_amdgcn_readlane(ctlv, CTL_COV + set * 2 + other) != epoch - 1) continue;
        int32_t *old = db.cov + (size_t)(set * 2 + other) * covLen;
        for (size_t k = (size_t)gwave * IGD_WAVE + (threadIdx.x & 63); k < covLen; k += (size_t)nwaves * IGD_WAVE) old[k] = 0;
    }
}

#ifndef IGD_COV_CHUNK
#define IGD_COV_CHUNK 0       // units a wave takes at a time; 0: every wave's share in as few chunks as a wave holds descriptors for (<= 64 units each)
#endif
template <bool USE_V>
__device__ __forceinline__ void coverage_body(const DbView &db, const ScanArgs &a, u64 *__restrict__ d_hits,
                                              u64 *__restrict__ d_total, int gwave, int nwaves, int ctlv, const TailHist hist)
{
    const int lane = threadIdx.x & 63;
    const bool uns = __builtin_amdgcn_readlane(ctlv, CTL_UNSORTED) == a.epoch;
    if (a.mode == 1 && uns) return;                      // broken promise: the batch adds nothing
    const int set = (a.mode == 1 || (a.mode == 0 && !uns)) ? 0 : 1, par = a.epoch & 1;   // merge join / bucket path
    if (__builtin_amdgcn_readlane(ctlv, CTL_COV + set * 2 + par) != a.epoch) return;     // no long query in this batch
    const int32_t *diff = db.cov + (size_t)(set * 2 + par) * IGD_COV_LEN(db.nT), *coarse = diff + db.nT + 2;
    u64 found = 0;
    // A chunk costs a chain of round trips before its first record arrives (the units' descriptors, their tiles' borders, the
    // coverage count at its first tile), so a wave's share comes in as few chunks as possible: one, of nUnits / nwaves units,
    // for the benchmark's 190 000 units on 4096 waves (16 at a time -- three chunks for most waves, two for the rest -- took as long
    // as the longer chain: the launch's 4 waves per SIMD neither ran out of instructions to issue nor waited for the LDS).
    int chunk = IGD_COV_CHUNK;
    if (chunk == 0) {
        chunk = (db.nUnits + nwaves - 1) / nwaves;
        chunk = chunk < 16 ? 16 : (chunk > IGD_WAVE ? IGD_WAVE : chunk);
    }
    for (int u0 = gwave * chunk; u0 < db.nUnits; u0 += nwaves * chunk) {
        const int cnt = db.nUnits - u0 < chunk ? db.nUnits - u0 : chunk;
        // one unit per lane: its tile, and the number of long queries that cover that tile from end to end = (coarse +
        // fine prefix at the chunk's first tile) + the differences of the tiles since -- no chain of loads from unit to unit
        UnitRegs ur = load_unit_regs(db.units + u0 + (lane < cnt ? lane : 0));
        const int tile = ur.tile, tile0 = __builtin_amdgcn_readfirstlane(tile);
        const int bd = db.tileBd[tile];                  // (a covered tile is never the first of its contig)
        // (every load below is asked for with a clamped index and masked afterwards: a load per loop round, or behind `k < gap`,
        // is a round trip of its own -- up to 16 + 3 + 4 of them in a row at the head of every wave's share, LABNOTES R5-12)
        int p0;
        {
            const int blk = tile0 >> IGD_COV_SHIFT, t0 = blk << IGD_COV_SHIFT;
            int sum = 0;
            for (int c0 = 0; c0 < blk; c0 += 4 * IGD_WAVE) {            // coarse sums of the blocks before: four per lane in flight
                int v[4];
#pragma unroll
                for (int k = 0; k < 4; k++) { const int c = c0 + k * IGD_WAVE + lane; v[k] = coarse[c < blk ? c : 0]; }
#pragma unroll
                for (int k = 0; k < 4; k++) sum += c0 + k * IGD_WAVE + lane < blk ? v[k] : 0;
            }
            constexpr int PER = (1 << IGD_COV_SHIFT) / IGD_WAVE;        // the block's fine differences up to tile0: all of a lane's in flight
            int v[PER];
#pragma unroll
            for (int k = 0; k < PER; k++) { const int t = t0 + k * IGD_WAVE + lane; v[k] = diff[t <= tile0 ? t : tile0]; }
#pragma unroll
            for (int k = 0; k < PER; k++) sum += t0 + k * IGD_WAVE + lane <= tile0 ? v[k] : 0;
            for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
            p0 = sum;
        }
        const int before = __shfl_up(tile, 1);
        const int gap = (lane == 0 || lane >= cnt) ? 0 : tile - before;   // tiles since the unit before (0: same tile; > 1: empty tiles between)
        int d = 0;
        {
            int v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = diff[tile - k >= 0 ? tile - k : 0];
            if (gap <= 4) {
#pragma unroll
                for (int k = 0; k < 4; k++) d += k < gap ? v[k] : 0;
            }
        }
        for (unsigned long long m = __ballot(gap > 4); m; m &= m - 1) {   // a run of empty tiles (a centromere): summed by the whole wave
            const int src = __builtin_ctzll(m);
            const int hi = __builtin_amdgcn_readlane(tile, src), g = __builtin_amdgcn_readlane(gap, src);
            int sum = 0;
            for (int t = hi - g + 1 + lane; t <= hi; t += IGD_WAVE) sum += diff[t];
            for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
            if (lane == src) d = sum;
        }
        const int since = wave_inclusive_sum(d);          // (every lane takes part)
        const int cv = lane < cnt ? p0 + since : 0;
        unsigned long long m = __ballot(cv > 0);
        if (m == 0) continue;                            // nothing of this chunk is covered
        // What bounds a covered unit is the round trip for its records: TWO units are in flight per wave (the loads of the
        // next covered unit are issued before the current one is counted).
        struct Cov { int st[IGD_SLOTS], ix[IGD_SLOTS], va[USE_V ? IGD_SLOTS : 1]; int c, lob, nrem; };
        // (compact image: the records that start in the tile are the unit's records from number `pre` on -- Unit::pre -- so
        // only their dataset numbers are read: 2 bytes a record, 4 with the value, where the exact arrays cost 8 and 12)
        // This launch runs 4 waves per SIMD and was bound by the instructions it issues (120 per covered unit: 2 x 10^7 for a
        // batch whose long queries cover the genome): the compact path asks for the records [pre, n) through a descriptor
        // that ends at n -- lanes past it read 0 without a compare, slots past it are not visited -- and knows what the
        // unit adds to the batch total without counting it (c x (n - pre)): ~50.
        const bool cimg = USE_V ? a.packedWalk == 2 : a.packedWalk != 0;
        const int vo2 = lane * 2, vo4 = lane * 4;
        auto issue = [&](int e, Cov &w) {
            w.c = __builtin_amdgcn_readlane(cv, e);
            const int n = __builtin_amdgcn_readlane(ur.n, e);
            const int64_t off = (int64_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(ur.offHi, e) << 32) |
                                          (unsigned)__builtin_amdgcn_readlane(ur.offLo, e));
            if (cimg) {
                const int pre = __builtin_amdgcn_readlane(ur.pre, e);
                const int nrem = n > pre ? n - pre : 0;
                w.nrem = nrem;
                if (USE_V) {
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(db.pxv + off + pre), 0, nrem * 4, 0x00020000);
#pragma unroll
                    for (int r = 0; r < IGD_SLOTS; r++) w.ix[r] = (int)__builtin_amdgcn_raw_buffer_load_b32(rs, vo4, r * 256, 0);
                } else {
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(db.px + off + pre), 0, nrem * 2, 0x00020000);
#pragma unroll
                    for (int r = 0; r < IGD_SLOTS; r++) w.ix[r] = (int)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rs, vo2, r * 128, 0);
                }
                return;
            }
            w.lob = __builtin_amdgcn_readlane(bd, e);
#pragma unroll
            for (int r = 0; r < IGD_SLOTS; r++) {
                const int i = r * IGD_WAVE + lane;
                const int64_t at = off + (i < n ? i : 0);      // (every lane loads: no load behind a branch)
                const int s0 = db.start[at], x0 = db.idx[at];
                w.st[r] = i < n ? s0 : INT_MIN;
                w.ix[r] = i < n ? x0 : 0;
                if (USE_V) { const int v0 = db.value[at]; w.va[r] = i < n ? v0 : INT_MIN; }
            }
        };
        auto count = [&](const Cov &w) {
            if (cimg) {
                if (!USE_V) found += (u64)(unsigned)w.nrem * (u64)(unsigned)w.c;
#pragma unroll
                for (int r = 0; r < IGD_SLOTS; r++) {
                    if (r * IGD_WAVE >= w.nrem) break;   // (the same for all lanes)
                    bool in = lane < w.nrem - r * IGD_WAVE;
                    int ix = w.ix[r];
                    if (USE_V) {
                        in = in && (ix >> 16) >= a.v;    // (arithmetic shift: the signed 16-bit value)
                        ix &= 0xFFFF;
                        found += (u64)__popcll(__ballot(in)) * (u64)(unsigned)w.c;
                    }
                    if (in) { if (hist.on) hist.add(ix, (u64)(unsigned)w.c); else atomicAdd(&d_hits[ix], (u64)(unsigned)w.c); }
                }
                return;
            }
#pragma unroll
            for (int r = 0; r < IGD_SLOTS; r++) {
                bool in = w.st[r] >= w.lob;              // the copy of the record that counts (:510-511)
                if (USE_V) in = in && w.va[r] >= a.v;
                found += (u64)__popcll(__ballot(in)) * (u64)(unsigned)w.c;
                if (in) { if (hist.on) hist.add(w.ix[r], (u64)(unsigned)w.c); else atomicAdd(&d_hits[w.ix[r]], (u64)(unsigned)w.c); }
            }
        };
        Cov C0, C1, C2, C3;                              // FOUR covered units in flight per wave
        int e0 = -1, e1 = -1, e2 = -1, e3 = -1;
#define IGD_COV_NEXT(E, C) do { E = -1; if (m) { E = __builtin_ctzll(m); m &= m - 1; issue(E, C); } } while (0)
        IGD_COV_NEXT(e0, C0); IGD_COV_NEXT(e1, C1); IGD_COV_NEXT(e2, C2); IGD_COV_NEXT(e3, C3);
        while (e0 >= 0) {
            count(C0); IGD_COV_NEXT(e0, C0);
            if (e1 >= 0) { count(C1); IGD_COV_NEXT(e1, C1); }
            if (e2 >= 0) { count(C2); IGD_COV_NEXT(e2, C2); }
            if (e3 >= 0) { count(C3); IGD_COV_NEXT(e3, C3); }
            if (e0 < 0) {
                if (e1 >= 0) { count(C1); e1 = -1; }
                if (e2 >= 0) { count(C2); e2 = -1; }
                if (e3 >= 0) { count(C3); e3 = -1; }
            }
        }
#undef IGD_COV_NEXT
    }
    if (d_total && lane == 0 && found) atomicAdd(d_total, found);
}

// The batch's last launch.  Besides its own job it hosts the exact walks and the two skew valves (`valves` bit 0:
// bucket path, bit 1: merge join; bit 2: BIG image), all of which normally find nothing to do.  SortK comes first:
// the merge join's code reads its rarer arguments from the kernel-argument segment (KARG).
template <bool USE_V>
__device__ __forceinline__ void batch_tail(const SortK &K, const ScanArgs &wa, const int2 *__restrict__ fixList,
                                           const int2 *__restrict__ longList, const int32_t *__restrict__ heavyB, int valves,
                                           u64 *__restrict__ d_hits, u64 *__restrict__ d_total, unsigned char *smem, int gwave, int nwaves,
                                           int ctlv /* the batch's control words, word i in lane i */)
{
    const int lane = threadIdx.x & 63;
    // what the exact walks and the coverage find is counted in the workgroup's LDS first (when the files fit): a batch of
    // long queries makes one addition per (query, record) pair here
    TailHist hist;
    hist.p = (igd_lds_u64 *)(smem + (K.a.tailHistOff >= 0 ? K.a.tailHistOff : 0));
    hist.on = false;
    if (K.a.tailHistOff >= 0) {
        const bool uns = __builtin_amdgcn_readlane(ctlv, CTL_UNSORTED) == wa.epoch;
        const bool sortedPath = wa.mode == 1 || (wa.mode == 0 && !uns);
        const int nList = __builtin_amdgcn_readlane(ctlv, (sortedPath ? CTL_NFIX : CTL_NLONG) + (wa.epoch & 1));
        const bool cov = __builtin_amdgcn_readlane(ctlv, CTL_COV + (sortedPath ? 0 : 2) + (wa.epoch & 1)) == wa.epoch;
        if (!(wa.mode == 1 && uns) && (nList > 0 || cov)) {          // (the same answer in every wave of the launch)
            hist.on = true;
            for (int f = threadIdx.x; f < K.db.nFiles; f += blockDim.x) hist.p[f] = 0;
            __syncthreads();
        }
    }
    // (Tried: the long queries' work on a quarter of the launch's workgroups, to quarter the 3.9 x 10^6 atomics with which 2048
    // workgroups flush 1900 LDS counters each -- slower, 560 -> 874 us for 10^6 queries of 100-200 kbp: the walks want the waves.)
    // (Tried: half the waves of a SIMD taking their coverage sums before their walks, so that the two kinds of work overlap:
    // 108.5 vs 109.1 us for 10^5 queries of 100-200 kbp -- both are bound by the instructions the launch's 4 waves per SIMD issue.)
    if (!(IGD_EXP & 0x100000)) exact_walk_body<USE_V>(K.db, wa, fixList, longList, gwave, nwaves, ctlv, hist);
    if (!(IGD_EXP & 0x200000)) coverage_body<USE_V>(K.db, wa, d_hits, d_total, gwave, nwaves, ctlv, hist);
    if (hist.on) {
        __syncthreads();
        for (int f = threadIdx.x; f < K.db.nFiles; f += blockDim.x) {
            const u64 c = hist.p[f];
            if (c) atomicAdd(&d_hits[f], c);
        }
    }
    if (valves & 1) heavy_bucket_body<USE_V>(K.db, wa, heavyB, d_hits, d_total, gwave, nwaves, lane, ctlv);
    if (valves & 8) {                                    // the batch took the DIRECT step (scan_direct.hpp): its dense tiles and deferred units
        DirArgs d;
        d.firstQ = K.a.firstQ; d.tileD = K.db.tileD; d.q_qs = wa.q_qs; d.q_qe = wa.q_qe; d.ctl = K.a.ctlw; d.fix = nullptr;
        d.heavyS = K.a.heavyS; d.farList = K.a.farList; d.nq = wa.nq; d.v = wa.v; d.epoch = wa.epoch; d.rule = wa.rule;
        d.promised = 1; d.sbCap = 0; d.wldsBytes = K.a.wldsBytes; d.out = nullptr; d.hitsOut = d_hits; d.totalOut = d_total;
        direct_tail_body<USE_V>(K.db, d, smem + (size_t)(threadIdx.x >> 6) * (size_t)K.a.wldsBytes, gwave, nwaves, lane, ctlv);
    } else
    if (valves & 2) {
        unsigned char *wsm = smem + (size_t)(threadIdx.x >> 6) * (size_t)K.a.wldsBytes;
        if (valves & 4) heavy_sorted_body<USE_V, true>(K, d_hits, d_total, wsm, gwave, nwaves, lane, ctlv);
        else {
            // what the two valves count goes to the workgroup's LDS counters first (when the files fit and a valve has work:
            // the same answer in every wave of the launch) and to hits[] once per workgroup
            const int par = wa.epoch & 1;
            const bool work = (__builtin_amdgcn_readlane(ctlv, CTL_NHEAVYS + par) | __builtin_amdgcn_readlane(ctlv, CTL_NFAR + par)) != 0 &&
                              __builtin_amdgcn_readlane(ctlv, CTL_UNSORTED) != wa.epoch;
            igd_lds_u64 *lh = (K.a.tailHistOff >= 0 && work) ? hist.p : nullptr;
            if (lh) {
                __syncthreads();                         // (the walks' flush above has read the counters)
                for (int f = threadIdx.x; f < K.db.nFiles; f += blockDim.x) lh[f] = 0;
                __syncthreads();
            }
            if (!(IGD_EXP & 0x10000)) heavy_sorted_body<USE_V, false>(K, d_hits, d_total, wsm, gwave, nwaves, lane, ctlv, lh);
            if (!(IGD_EXP & 0x20000)) far_units_body<USE_V, false>(K, d_hits, d_total, wsm, gwave, nwaves, lane, ctlv, lh);   // (the lean build does not exist for BIG images)
            if (lh) {
                __syncthreads();
                for (int f = threadIdx.x; f < K.db.nFiles; f += blockDim.x) {
                    const u64 c = lh[f];
                    if (c) atomicAdd(&d_hits[f], c);
                }
            }
        }
    }
}

template <bool USE_V>
__global__ __launch_bounds__(256) void k_exact_walk(SortK K, ScanArgs a, const int2 *__restrict__ fixList,
                                                    const int2 *__restrict__ longList, const int32_t *__restrict__ heavyB, int valves)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int gwave = blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ctlv = (threadIdx.x & 63) < IGD_CTL_WORDS ? a.ctl[threadIdx.x & 63] : 0;
    coverage_reset(K.db, a.epoch, gwave, gridDim.x * (blockDim.x >> 6), ctlv);
    batch_tail<USE_V>(K, a, fixList, longList, heavyB, valves, a.out, nullptr, smem, gwave, gridDim.x * (blockDim.x >> 6), ctlv);
}

// slab rows -> int64 hits[] (+ batch total).  grid = (ceil(nFiles/IGD_TAIL_WG), IGD_REDUCE_GROUPS)
#ifndef IGD_TAIL_OCC
#define IGD_TAIL_ATTR
#else
#define IGD_TAIL_ATTR __attribute__((amdgpu_waves_per_eu(IGD_TAIL_OCC, IGD_TAIL_OCC)))   // A/B: registers cut to what IGD_TAIL_OCC waves per SIMD leave each
#endif
template <bool USE_V>
__global__ __launch_bounds__(IGD_TAIL_WG) IGD_TAIL_ATTR void k_reduce_slabs(SortK K, const u64 *__restrict__ slab, int rows, int nFiles,
                                                      u64 *__restrict__ hits, u64 *__restrict__ total,
                                                      const int32_t *__restrict__ ctl, int brokenIf,
                                                      ScanArgs wa, const int2 *__restrict__ fixList,
                                                      const int2 *__restrict__ longList, const int32_t *__restrict__ heavyB, int valves,
                                                      int rows32 /* != 0: igd_scan_sorted of epoch `rows32` wrote 32-bit rows (unless the batch went to the bucket path) */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ u64 red[IGD_TAIL_WG / IGD_WAVE];
    constexpr int WPB = IGD_TAIL_WG / IGD_WAVE;           // waves per workgroup
    // the batch's control words, word i in lane i: one load, in flight together with the slab rows
    const int ctlv = (threadIdx.x & 63) < IGD_CTL_WORDS ? ctl[threadIdx.x & 63] : 0;
    int f = blockIdx.x * IGD_TAIL_WG + threadIdx.x;
    u64 s = 0;
    if (f < nFiles && blockIdx.y < IGD_REDUCE_GROUPS) {  // (the workgroups beyond are there for the batch's tail only, see the launch)
        // (both kinds of rows are read before the control words say which kind this batch left: the loads are in flight
        // together, and a row of either kind lies inside the slab)
        u64 s64 = 0;
        unsigned long long s32 = 0;
        if (rows32) {
            const unsigned int *slab32 = (const unsigned int *)slab;
            for (int g = blockIdx.y; g < rows; g += IGD_REDUCE_GROUPS) s32 += slab32[(size_t)g * nFiles + f];
            if (__builtin_amdgcn_readlane(ctlv, CTL_UNSORTED) == rows32)   // the bucket path's batch: 64-bit rows
                for (int g = blockIdx.y; g < rows; g += IGD_REDUCE_GROUPS) s64 += slab[(size_t)g * nFiles + f];
            s = __builtin_amdgcn_readlane(ctlv, CTL_UNSORTED) == rows32 ? s64 : s32;
        } else
            for (int g = blockIdx.y; g < rows; g += IGD_REDUCE_GROUPS) s += slab[(size_t)g * nFiles + f];
    }
    // brokenIf != 0: the batch ran under IGD_HIP_FLAG_SORTED; if the device found it unsorted the
    // scan kernel wrote no slab, so nothing may be added
    if (valves >= 0) coverage_reset(K.db, wa.epoch, (blockIdx.y * gridDim.x + blockIdx.x) * WPB + (int)(threadIdx.x >> 6), gridDim.x * gridDim.y * WPB, ctlv);
    if (brokenIf != 0 && __builtin_amdgcn_readlane(ctlv, CTL_UNSORTED) == brokenIf) return;
    if (s) atomicAdd(&hits[f], s);
    if (total) {
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            u64 t = 0;
            for (int k = 0; k < WPB; k++) t += red[k];
            if (t) atomicAdd(total, t);
        }
    }
    if (valves < 0) return;                              // an earlier pass of a windowed batch: the tail rides in the last pass's launch
    // ... and the batch's exact-walk list and skew valves (normally empty) ride in the same launch
    const int nb = gridDim.x * gridDim.y;
    const int bid = blockIdx.y * gridDim.x + blockIdx.x;
    const int gwave = bid * WPB + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    batch_tail<USE_V>(K, wa, fixList, longList, heavyB, valves, wa.out, total, smem, gwave, nb * WPB, ctlv);   // (wa.out: the caller's hits[]; `hits` is a window of it in a windowed batch)
}

// without LDS counters the batch total is the growth of sum(hits): measured around the launch
__global__ __launch_bounds__(256) void k_sum_hits(const u64 *__restrict__ hits, int nFiles,
                                                  u64 *__restrict__ total, int sign)
{
    __shared__ u64 red[4];
    u64 s = 0;
    for (int f = threadIdx.x; f < nFiles; f += 256) s += hits[f];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 t = red[0] + red[1] + red[2] + red[3];
        if (sign > 0) atomicAdd(total, t);
        else atomicAdd(total, (u64)(-(int64_t)t));
    }
}
