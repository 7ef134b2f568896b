// engine/scan_tiles.hpp -- part of igd_hip.hip (included there once; not a stand-alone header).
// igd_scan_tiles (bucket path) and its skew valve
// ------------------------------------------------------------------------------------------
// The scan kernel.
//
// Work unit = <= IGD_CHUNK records of one tile.  A wave owns units gwave, gwave+nwaves, ...
//  * Descriptors: the Unit and the query range (merge join: firstQ[]; bucket: pairN/pairPos)
//    of the wave's next 64 units are fetched ONE PER LANE -- a two-level dependent load done
//    once, in parallel -- and broadcast with v_readlane when their turn comes, so no scalar-load
//    round trip sits in the per-unit path.
//  * Records: slot r of lane l is record r*64+l of the unit (coalesced loads).  PACKED: the
//    compact words are compared in place with 16-bit compares (s' low half, e' high half), 2
//    VGPRs per slot, and every wave keeps TWO units in flight (the loads of unit k+1 are issued
//    before unit k is compared: twice the bytes outstanding, compares overlap loads).  Exact
//    arrays (3 VGPRs per slot): one unit at a time.
//  * Compare, per (query, slot): lob <= start < qe && end > qs [&& value >= v]; lob = tile start
//    for a non-first tile is the reference's tS prefix skip (:510-511), start < qe is what its
//    bisection computes (:479-487).  Tiles are sorted by start, so once the first start of a
//    slot is >= qe the remaining slots cannot match: a wave-uniform loop exit.
//  * A hit is one ds_add_u64 into the workgroup's private LDS copy of hits[].
struct Raw {
    uint32_t a[IGD_SLOTS];       // PACKED: s' | e' << 16        exact: start
    int32_t b[IGD_SLOTS];        //                              exact: end
    int32_t x[IGD_SLOTS];        // idx, then idx * 8 (byte offset of the counter)
    int32_t w[IGD_SLOTS];        // value (USE_V)
    int32_t q0, q1, q2;          // first 64 candidates: merge join ichr,qs,qe ; bucket qs,qe,-
};

// A Unit held one-per-lane in VGPRs, and its wave-uniform broadcast.
struct UnitRegs { int32_t offLo, offHi, tile, n, jf, w[6], pre; };
__device__ __forceinline__ UnitRegs load_unit_regs(const Unit *p)
{
    const int4 a = ((const int4 *)p)[0], b = ((const int4 *)p)[1], c = ((const int4 *)p)[2];
    UnitRegs r;
    r.offLo = a.x; r.offHi = a.y; r.tile = a.z; r.n = a.w;
    r.jf = b.x; r.w[0] = b.y; r.w[1] = b.z; r.w[2] = b.w;
    r.w[3] = c.x; r.w[4] = c.y; r.w[5] = c.z; r.pre = c.w;
    return r;
}

struct ScanArgs {
    const int32_t *firstQ;       // merge join: [nT+1]
    const int32_t *pairN;        // bucket path: pairs per tile
    const int32_t *pairPos;      //              end of each tile's range in `pairs`
    const int2 *pairs;
    const int2 *walkList;        // exact-walk list of this batch's path (k_exact_walk only)
    const int32_t *ctl;
    const int32_t *q_ichr, *q_qs, *q_qe;
    const int32_t *q_w;          // merge join: per query (first global tile << 4 | span), from k_query_bounds
    int nq, v, rule, epoch;
    int mode;                    // 0: device decides (ctl[CTL_UNSORTED]); 1: sorted promised; 2: bucket
    u64 *out;                    // slab [grid][nFiles] (LDS counters) or the global hits[]
    u64 *total;                  // k_exact_walk: batch total (may be null)
    u64 *hitsOut;                // the caller's hits[] (the skew kernels add to it directly)
    int packedWalk;              // the exact walk of a long query's LAST tile may read the compact image: 1 (pse + px), 2 (pse + pxv: `-v`), 0 (no image)
};

// Issue the loads of unit kk.  BRANCH-FREE on purpose: every call issues exactly the same
// number of loads (out-of-range lanes and unvisited units read element 0 of an array instead of
// being skipped), so that the compiler can count them and wait for unit k with s_waitcnt
// vmcnt(N) while the loads of unit k+1 stay in flight.  A conditional load would force vmcnt(0)
// and serialise the two buffers.  The masking happens in compute_unit.
template <bool SORTED, bool USE_V, bool PACKED>
__device__ __forceinline__ void issue_unit(const DbView &db, const ScanArgs &a, const UnitRegs &L, int Lr0,
                                           int Lr1, int kk, int lane, Raw &R)
{
    const int r0 = __builtin_amdgcn_readlane(Lr0, kk), r1 = __builtin_amdgcn_readlane(Lr1, kk);
    const bool active = SORTED ? (r1 > r0) : (r0 > 0);
    const int n = active ? __builtin_amdgcn_readlane(L.n, kk) : 0;
    const int64_t off = (int64_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(L.offHi, kk) << 32) |
                                  (unsigned)__builtin_amdgcn_readlane(L.offLo, kk));
    const int64_t base = active ? off : 0;
#if IGD_BUFFER_LOADS
    if (PACKED) {
        // Buffer loads with a per-unit descriptor: hardware bounds checking returns 0 for lanes past
        // the unit's last record (0 is the "never matches" word) and for unvisited units (n = 0) no
        // memory is touched at all; the per-lane part of the address is just lane*4 + r*256.
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)(db.pse + base), 0, n * 4, 0x00020000);
        const int vo4 = lane * 4, vo2 = lane * 2;
        if (USE_V) {
            const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void *)(db.pxv + base), 0, n * 4, 0x00020000);
#pragma unroll
            for (int r = 0; r < IGD_SLOTS; r++) {
                R.a[r] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsA, vo4, r * 256, 0);   // slot offset: an immediate
                R.x[r] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsX, vo4, r * 256, 0);
            }
        } else {
            const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void *)(db.px + base), 0, n * 2, 0x00020000);
#pragma unroll
            for (int r = 0; r < IGD_SLOTS; r++) {
                R.a[r] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsA, vo4, r * 256, 0);
                R.x[r] = (int)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsX, vo2, r * 128, 0);
            }
        }
    } else
#endif
    {
    // uniform base pointers + (slot*64 + lane): the loads need no per-lane address arithmetic.
    // Lanes past the unit's last record read the next unit's records (the arrays are padded by
    // one chunk); compute_unit discards them.
    const uint32_t *pa = db.pse + base;
    const uint16_t *pxx = db.px + base;
    const uint32_t *pvv = USE_V ? db.pxv + base : nullptr;
#pragma unroll
    for (int r = 0; r < IGD_SLOTS; r++) {
        const int i = r * IGD_WAVE + lane;
        const int64_t at = base + (i < n ? i : 0);
        if (PACKED) {
            R.a[r] = pa[i];
            if (USE_V) R.x[r] = (int)pvv[i];             // idx | value << 16: one word, one register
            else R.x[r] = (int)pxx[i];
        } else {
            R.a[r] = (uint32_t)db.start[at];
            R.b[r] = db.end[at];
            R.x[r] = db.idx[at];
            if (USE_V) R.w[r] = db.value[at];
        }
    }
    }
    if (SORTED) {
        int i = r0 + lane;
        i = (active && i < r1) ? i : 0;
        R.q0 = a.q_w[i];
        R.q1 = a.q_qs[i];
        R.q2 = a.q_qe[i];
    } else {
        const int i = (active && lane < r0) ? r1 - r0 + lane : 0;
        const int2 pr = a.pairs[i];
        R.q0 = pr.x; R.q1 = pr.y;
    }
}

// Per-query parameters of the compare, computed for 64 candidate queries at once (one per lane)
// and broadcast to the wave one query at a time.
// PACKED: ONE word  (65536 - qe') | qs' << 16  (see k_pack_units); a record matches when both
//         halves of its word are >= the halves of the query word.
// exact : p0 = lob (INT_MIN first tile / tile start), p1 = qe - lob, p2 = qs; a record matches
//         when (unsigned)(start - p0) < p1 && end > p2 -- one subtraction and one unsigned compare
//         give lob <= start < qe together.
// No scalar-ALU work is needed per record slot, which matters: a CU has a single scalar unit.
typedef unsigned short igd_u16x2 __attribute__((ext_vector_type(2)));


// one query against the unit's slots: cnt[r] += hit   (no branches, no exec masking, no LDS)
template <bool USE_V, bool PACKED>
__device__ __forceinline__ void match_raw(const Raw &R, int (&cnt)[IGD_SLOTS], int p0, int p1, int p2, int v)
{
#pragma unroll
    for (int r = 0; r < IGD_SLOTS; r++) {
#if IGD_EXP_NOMATCH
        asm volatile("" ::"v"(R.a[r]), "v"(R.x[r]));
        continue;
#endif
        if (PACKED) {
            igd_u16x2 rec, qw;
            __builtin_memcpy(&rec, &R.a[r], 4);
            __builtin_memcpy(&qw, &p0, 4);
            const igd_u16x2 mx = __builtin_elementwise_max(rec, qw);   // v_pk_max_u16
            uint32_t mxw;
            __builtin_memcpy(&mxw, &mx, 4);
            cnt[r] += mxw == R.a[r] ? 1 : 0;             // both halves already >= the query's
        } else {
            const uint32_t d = R.a[r] - (uint32_t)p0;
            int t = d < (uint32_t)p1 ? R.b[r] : INT_MIN;
            if (USE_V) t = R.w[r] >= v ? t : INT_MIN;
            cnt[r] += t > p2 ? 1 : 0;
        }
    }
}

// One slot of a unit against the queries of a batch of <= 64 (word P0 per lane) that pass against the slot's summary word
// W (and are `live`): cnt += hit.  The loop over the picked queries is written out: left to the compiler it keeps the mask
// in VCC but clears its bit with a shift and an and-not (or an add -1 and an and) and tests it with a compare -- 5 scalar
// + 4 vector instructions per (slot, query); here s_bitset0 and the branch on VCC itself: 3 + 4, and the kernels that run
// it are bound by instruction issue (the headline scan: 68.5 -> 65 us).  Wait states (gfx950): a packed op's result read
// by the next VALU 1, an SGPR written by a VALU (v_readlane, v_cmp) read by a VALU 2.
template <bool LIVE>
__device__ __forceinline__ void match_slot_asm(int &cnt, uint32_t W, int P0, uint32_t rec, unsigned long long live)
{
    int t, q, x;
    unsigned long long c;
    if (LIVE)
        asm volatile("v_pk_max_u16 %[x], %[W], %[P0]\n\t"
                     "s_nop 0\n\t"
                     "v_cmp_eq_u32_e32 vcc, %[W], %[x]\n\t"
                     "s_and_b64 vcc, vcc, %[live]\n\t"
                     "s_cbranch_vccz 2f\n"
                     "1:\n\t"
                     "s_ff1_i32_b64 %[t], vcc\n\t"
                     "v_readlane_b32 %[q], %[P0], %[t]\n\t"
                     "s_bitset0_b64 vcc, %[t]\n\t"
                     "s_nop 0\n\t"
                     "v_pk_max_u16 %[x], %[rec], %[q]\n\t"
                     "s_nop 0\n\t"
                     "v_cmp_eq_u32_e64 %[c], %[rec], %[x]\n\t"
                     "s_nop 1\n\t"
                     "v_addc_co_u32_e64 %[cnt], %[c], 0, %[cnt], %[c]\n\t"
                     "s_cbranch_vccnz 1b\n"
                     "2:"
                     : [cnt] "+v"(cnt), [t] "=&s"(t), [q] "=&s"(q), [x] "=&v"(x), [c] "=&s"(c)
                     : [W] "s"(W), [P0] "v"(P0), [rec] "v"(rec), [live] "s"(live)
                     : "vcc", "scc");
    else
        asm volatile("v_pk_max_u16 %[x], %[W], %[P0]\n\t"
                     "s_nop 0\n\t"
                     "v_cmp_eq_u32_e32 vcc, %[W], %[x]\n\t"
                     "s_cbranch_vccz 2f\n"
                     "1:\n\t"
                     "s_ff1_i32_b64 %[t], vcc\n\t"
                     "v_readlane_b32 %[q], %[P0], %[t]\n\t"
                     "s_bitset0_b64 vcc, %[t]\n\t"
                     "s_nop 0\n\t"
                     "v_pk_max_u16 %[x], %[rec], %[q]\n\t"
                     "s_nop 0\n\t"
                     "v_cmp_eq_u32_e64 %[c], %[rec], %[x]\n\t"
                     "s_nop 1\n\t"
                     "v_addc_co_u32_e64 %[cnt], %[c], 0, %[cnt], %[c]\n\t"
                     "s_cbranch_vccnz 1b\n"
                     "2:"
                     : [cnt] "+v"(cnt), [t] "=&s"(t), [q] "=&s"(q), [x] "=&v"(x), [c] "=&s"(c)
                     : [W] "s"(W), [P0] "v"(P0), [rec] "v"(rec)
                     : "vcc");
}

// Compact image: the queries of `live` (one per lane, word P0) against the unit, slot by slot.  A
// query is compared with the records of a slot only if its word passes against the slot's summary
// W[r] -- the same packed test, done for 64 queries at once; on the benchmark that leaves 1.8 of
// 4.9 slots per (query, unit).  cnt[r] += hit; no exec masking, no LDS.
__device__ __forceinline__ void match_slots(const Raw &R, int (&cnt)[IGD_SLOTS], const uint32_t (&W)[IGD_SLOTS], int P0,
                                            unsigned long long live)
{
#if IGD_ASM_MATCH && !IGD_EXP_NOMATCH
#pragma unroll
    for (int r = 0; r < IGD_SLOTS; r++) match_slot_asm<true>(cnt[r], W[r], P0, R.a[r], live);
    return;
#endif
    igd_u16x2 qv;
    __builtin_memcpy(&qv, &P0, 4);
#pragma unroll
    for (int r = 0; r < IGD_SLOTS; r++) {
        igd_u16x2 wv;
        __builtin_memcpy(&wv, &W[r], 4);
        const igd_u16x2 mw = __builtin_elementwise_max(wv, qv);
        uint32_t mww;
        __builtin_memcpy(&mww, &mw, 4);
        unsigned long long m = __ballot(mww == W[r]) & live;
#if IGD_EXP_NOMATCH
        asm volatile("" ::"v"(R.a[r]), "v"(R.x[r]));
        continue;
#endif
        while (m) {
            const int src = __builtin_ctzll(m);
            m &= m - 1;
            const int q = __builtin_amdgcn_readlane(P0, src);
            igd_u16x2 rec, qw;
            __builtin_memcpy(&rec, &R.a[r], 4);
            __builtin_memcpy(&qw, &q, 4);
            const igd_u16x2 mx = __builtin_elementwise_max(rec, qw);   // v_pk_max_u16
            uint32_t mxw;
            __builtin_memcpy(&mxw, &mx, 4);
            cnt[r] += mxw == R.a[r] ? 1 : 0;             // both halves already >= the query's
        }
    }
}

template <bool SORTED, bool USE_V, bool PACKED, bool WIN = false>
__device__ __forceinline__ void compute_unit(const DbView &db, const ScanArgs &a, const UnitRegs &L, int Lr0,
                                             int Lr1, int kk, int lane, Raw &R, u64 *hits, u64 *found = nullptr)
{
    const int r0 = __builtin_amdgcn_readlane(Lr0, kk), r1 = __builtin_amdgcn_readlane(Lr1, kk);
    const bool active = SORTED ? (r1 > r0) : (r0 > 0);
    if (!active) return;
    const int un = __builtin_amdgcn_readlane(L.n, kk);
    if (un == 0) return;                                 // placeholder of an empty tile
    const int jf = __builtin_amdgcn_readlane(L.jf, kk);
    const int uj = jf >> 4;
    const int T = (int)((unsigned)uj * (unsigned)db.nbp);
    const int bd = uj == 0 ? INT_MIN : T;                // tile start; "no lower bound" in tile 0 (src/igd_search.c:496,529)
    // slot summaries (see k_pack_units): the largest word a record of the slot could have
    uint32_t W[IGD_SLOTS];
    if (PACKED) {
#pragma unroll
        for (int r = 0; r < IGD_SLOTS; r++) W[r] = (uint32_t)__builtin_amdgcn_readlane(L.w[r], kk);
    }
    int cnt[IGD_SLOTS];
#pragma unroll
    for (int r = 0; r < IGD_SLOTS; r++) {
        cnt[r] = 0;
        // lanes past the unit's last record hold someone else's data: make them unmatchable;
        // (compact image) so are records that fail the value filter -- v is fixed for the batch
        bool drop = (IGD_BUFFER_LOADS && PACKED) ? false : (r * IGD_WAVE + lane >= un);
        if (PACKED && USE_V) {
            drop = drop || (R.x[r] >> 16) < a.v;         // arithmetic shift: the signed 16-bit value
            R.x[r] &= 0xFFFF;
        }
        if (drop) R.a[r] = PACKED ? 0u : (uint32_t)INT_MAX;
    }
    int nLater = 0;                                      // covering queries for which this is NOT the first tile
    if (SORTED) {
        const int ut = __builtin_amdgcn_readlane(L.tile, kk);   // global tile number of this unit
        // rule NEST: a query whose FIRST tile is empty is dead (src/igd_search.c:468); which of the
        // previous tiles are empty is a property of the database (flag bits 1..3 of the unit)
        const int deadk = a.rule == IGD_HIP_RULE_NEST ? (jf & 15) : 0;
        for (int p = r0; p < r1; p += IGD_WAVE) {
            int w = (p + lane < r1) ? R.q0 : -1, qs_ = R.q1, qe_ = R.q2;
            if (p != r0) {
                const int i = p + lane;
                const bool in = i < r1;
                w = in ? a.q_w[i] : -1;
                qs_ = in ? a.q_qs[i] : 0;
                qe_ = in ? a.q_qe[i] : 0;
            }
            const int k = ut - (w >> 4);                 // 0: this is the query's first tile
            // k == 0 with qe <= T is left to k_exact_walk when the compact image is read;
            // tiles further than IGD_SHORT_TILES-1 behind are too (long queries).  The span is clamped
            // to the query's own contig, so k <= span also says "same contig".
            const bool later = k > 0 && k < IGD_SHORT_TILES && (w & 15) >= k && !((deadk >> k) & 1);
            const bool covers = w >= 0 && ((k == 0 && !(PACKED && qe_ <= T)) || later);
            int P0, P1 = 0, P2 = 0;
            if (PACKED) P0 = query_word(qs_, qe_, k == 0, T, db.nbp);
            else {
                P0 = k == 0 ? INT_MIN : bd;
                P1 = (int)((unsigned)qe_ - (unsigned)P0);
                P2 = qs_;
            }
            unsigned long long m = __ballot(covers);
            if (PACKED) {
                nLater += __popcll(__ballot(covers && later));
                match_slots(R, cnt, W, P0, m);
            } else {
                while (m) {
                    const int src = __builtin_ctzll(m);
                    m &= m - 1;
                    match_raw<USE_V, PACKED>(R, cnt, __builtin_amdgcn_readlane(P0, src),
                                             __builtin_amdgcn_readlane(P1, src), __builtin_amdgcn_readlane(P2, src), a.v);
                }
            }
        }
    } else {
        const int np = r0, pend = r1;
        for (int p = pend - np; p < pend; p += IGD_WAVE) {
            int m = pend - p;
            if (m > IGD_WAVE) m = IGD_WAVE;
            int px_ = R.q0, py_ = R.q1;
            if (p != pend - np) {
                const int2 pr = (lane < m) ? a.pairs[p + lane] : make_int2(0, INT_MIN);
                px_ = pr.x; py_ = pr.y;
            }
            const bool first = px_ >= bd;                // tile 0: bd = INT_MIN, always first
            int P0, P1 = 0, P2 = 0;
            if (PACKED) {
                P0 = query_word(px_, py_, first, T, db.nbp);
                nLater += __popcll(__ballot(lane < m && !first));
            } else {
                P0 = first ? INT_MIN : bd;
                P1 = (int)((unsigned)py_ - (unsigned)P0);
                P2 = px_;
            }
            if (PACKED) match_slots(R, cnt, W, P0, m >= IGD_WAVE ? ~0ull : ((1ull << m) - 1ull));
            else
                for (int k = 0; k < m; k++)
                    match_raw<USE_V, PACKED>(R, cnt, __builtin_amdgcn_readlane(P0, k), __builtin_amdgcn_readlane(P1, k),
                                             __builtin_amdgcn_readlane(P2, k), a.v);
        }
    }
    // one LDS atomic per record that was hit, with the number of queries that hit it
    // records that start before the tile (s' = 0, low half 65535) were matched by every
    // "later tile" query, none of which may count them (the reference's tS skip, :510-511);
    // most units have no such query at all
    if (PACKED && nLater != 0) {
#pragma unroll
        for (int r = 0; r < IGD_SLOTS; r++) cnt[r] -= (R.a[r] & 0xFFFFu) == 0xFFFFu ? nLater : 0;
    }
#pragma unroll
    for (int r = 0; r < IGD_SLOTS; r++) {
        const int c = cnt[r];
        if (WIN) {                                       // this pass counts the files of its window only
            const unsigned x = (unsigned)R.x[r] - (unsigned)db.fileLo;
            if (c && x < (unsigned)db.nFiles) atomicAdd((u64 *)((char *)hits + ((size_t)x << 3)), (u64)(unsigned)c);
        } else
        if (c) atomicAdd((u64 *)((char *)hits + ((size_t)R.x[r] << 3)), (u64)(unsigned)c);
    }
    if (found) {                                         // skew valves: the batch total is kept by the caller of this unit
        int t = 0;
#pragma unroll
        for (int r = 0; r < IGD_SLOTS; r++) t += cnt[r];
        for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o);
        if (lane == 0 && t) atomicAdd(found, (u64)(unsigned)t);
    }
}

// SORTED = true : merge join over the caller's ordered arrays (firstQ[])
// SORTED = false: bucketed pairs
// In the device-decides mode both are enqueued and the one that does not apply returns at once.
// WIN: more files than LDS counters (15 360): the batch is scanned once per window of files, each pass counting its own
// (db.fileLo, db.nFiles = the window); per-record global atomics -- the alternative -- run at 2.4e10 per second on this
// part whatever their scope (tools/atomic_bench.hip): 1.05 ms per 10^6 queries where a pass takes 0.08
template <bool SORTED, bool USE_V, bool LDS_HITS, bool PACKED, bool WIN = false>
__global__ __launch_bounds__(IGD_WG, IGD_WPE) void igd_scan_tiles(DbView db, ScanArgs a)
{
    {
        const bool uns = __builtin_amdgcn_readfirstlane(a.ctl[CTL_UNSORTED]) == a.epoch;
        if (SORTED ? uns : (a.mode == 0 && !uns)) return;     // not this kernel's batch
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64 *hits = LDS_HITS ? (u64 *)smem : a.out;
    if (LDS_HITS) {
        for (int f = threadIdx.x; f < db.nFiles; f += IGD_WG) hits[f] = 0;
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const int wavesPerWG = IGD_WG / IGD_WAVE;
    // readfirstlane: the wave index is uniform; everything derived from it stays in SGPRs
    const int gwave = blockIdx.x * wavesPerWG + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nwaves = gridDim.x * wavesPerWG;
    Raw A, B;
    // issue-priority balancing between the waves of a SIMD (see igd_scan_sorted): a wave lowers its priority as it
    // gets through its share, so that the eight waves of a SIMD finish together instead of oldest first
    const int myUnits = (db.nUnits - gwave + nwaves - 1) / nwaves;
    const int quarter = (myUnits + 3) >> 2;
    int prioAt = quarter, prioLevel = 3, done = 0;
#if IGD_OPT_PRIO
    __builtin_amdgcn_s_setprio(3);
#endif

    for (int ub = gwave; ub < db.nUnits; ub += nwaves * IGD_WAVE) {
        UnitRegs L;
        int Lr0 = 0, Lr1 = 0;
        {
            const long long mi = (long long)ub + (long long)lane * nwaves;
            if (mi < db.nUnits) {
                L = load_unit_regs(db.units + mi);
                if (SORTED) {
                    const int lj = L.jf >> 4;
                    const int lb = lj < IGD_SHORT_TILES - 1 ? lj : IGD_SHORT_TILES - 1;
                    if (L.n > 0) {
                        Lr0 = a.firstQ[L.tile - lb];
                        Lr1 = a.firstQ[L.tile + 1];
                    }
                } else if (L.n > 0) {
                    Lr0 = a.pairN[L.tile];
                    if (Lr0 < 0) Lr0 = 0;                 // listed for heavy_bucket_body
                    Lr1 = a.pairPos[L.tile];
                }
            }
        }
        int cntU = (int)(((long long)db.nUnits - ub + nwaves - 1) / nwaves);
        if (cntU > IGD_WAVE) cntU = IGD_WAVE;
        if (PACKED) {
            issue_unit<SORTED, USE_V, PACKED>(db, a, L, Lr0, Lr1, 0, lane, A);
            for (int kk = 0; kk < cntU; kk += 2) {
                if (kk + 1 < cntU) issue_unit<SORTED, USE_V, PACKED>(db, a, L, Lr0, Lr1, kk + 1, lane, B);
                compute_unit<SORTED, USE_V, PACKED, WIN>(db, a, L, Lr0, Lr1, kk, lane, A, hits);
                if (kk + 2 < cntU) issue_unit<SORTED, USE_V, PACKED>(db, a, L, Lr0, Lr1, kk + 2, lane, A);
                if (kk + 1 < cntU) compute_unit<SORTED, USE_V, PACKED, WIN>(db, a, L, Lr0, Lr1, kk + 1, lane, B, hits);
#if IGD_OPT_PRIO
                done += 2;
                if (done >= prioAt) {
                    prioAt += quarter;
                    prioLevel--;
                    if (prioLevel == 2) __builtin_amdgcn_s_setprio(2);
                    else if (prioLevel == 1) __builtin_amdgcn_s_setprio(1);
                    else __builtin_amdgcn_s_setprio(0);
                }
#endif
            }
        } else {
            for (int kk = 0; kk < cntU; kk++) {
                issue_unit<SORTED, USE_V, PACKED>(db, a, L, Lr0, Lr1, kk, lane, A);
                compute_unit<SORTED, USE_V, PACKED, WIN>(db, a, L, Lr0, Lr1, kk, lane, A, hits);
#if IGD_OPT_PRIO
                done += 1;
                if (done >= prioAt) {
                    prioAt += quarter;
                    prioLevel--;
                    if (prioLevel == 2) __builtin_amdgcn_s_setprio(2);
                    else if (prioLevel == 1) __builtin_amdgcn_s_setprio(1);
                    else __builtin_amdgcn_s_setprio(0);
                }
#endif
            }
        }
    }

    if (LDS_HITS) {
        __syncthreads();
        u64 *row = a.out + (size_t)blockIdx.x * db.nFiles;
        for (int f = threadIdx.x; f < db.nFiles; f += IGD_WG) row[f] = hits[f];
    }
}

// deal_items: the work items of a skew valve's listed tiles, dealt round-robin to all waves of the hosting launch.  The
// tiles are looked up 64 at a time, one per lane (look(h) -> the tile's number of items, 0 for h < 0; it keeps what it
// found in lane variables), and one modulo per group finds the wave's first item; fn(lane of the tile, item within the
// tile) then runs for every item of this wave.  (A loop over the tiles with a chain of dependent loads and two 64-bit
// remainders per tile and wave made 1000 listed tiles cost EVERY wave of the launch 0.75 ms.)
template <typename LOOK, typename FN>
__device__ __forceinline__ void deal_items(int nH, int gwave, int nwaves, int lane, LOOK look, FN fn)
{
    long long base = 0;
    for (int h0 = 0; h0 < nH; h0 += IGD_WAVE) {
        const int items = look(h0 + lane < nH ? h0 + lane : -1);
        const int incl = wave_inclusive_sum(items);
        const int total = __builtin_amdgcn_readlane(incl, IGD_WAVE - 1);
        long long r = ((long long)gwave - base) % nwaves;
        if (r < 0) r += nwaves;
        for (long long g = r; g < total; g += nwaves) {
            const int hh = __popcll(__ballot(incl <= (int)g));                     // the tile (lane) that holds item g of the group
            fn(hh, (int)g - (hh ? __builtin_amdgcn_readlane(incl, hh - 1) : 0));
        }
        base += total;
    }
}

// ------------------------------------------------------------------------------------------
// heavy_bucket_body: the bucket path's skew valve.  The tile chunk is the unit of work, so a batch whose queries pile
// up in a few tiles (10^6 unordered queries in ONE tile: 62 ms) would be serialised on the waves that own them.
// k_split_fine lists the tiles with more than IGD_HEAVY_PAIRS pairs and hides them from igd_scan_tiles (negative
// pair count); here every (unit of the tile, slice of IGD_HEAVY_PAIRS pairs) is one work item, dealt round-robin to
// all waves of the hosting launch (the batch's last kernel: k_reduce_slabs / k_exact_walk -- a launch of its own would
// cost every batch 4 us), compared exactly like any other unit (compute_unit) and added to hits[] and the batch
// total with global atomics.  Nothing listed: one load per wave.
template <bool USE_V>
__device__ __forceinline__ void heavy_bucket_body(const DbView &db, const ScanArgs &a, const int32_t *__restrict__ heavy,
                                                  u64 *__restrict__ d_hits, u64 *__restrict__ d_total, int gwave, int nwaves, int lane,
                                                  int ctlv)
{
    int nH = __builtin_amdgcn_readlane(ctlv, CTL_NHEAVY + (a.epoch & 1));
    if (nH == 0) return;
    if (nH > IGD_HEAVY_MAX) nH = IGD_HEAVY_MAX;
    int lnp = 0, lpend = 0, lu0 = 0, lnu = 0;
    deal_items(nH, gwave, nwaves, lane,
        [&](int h) {
            lnp = lpend = lu0 = lnu = 0;
            if (h < 0) return 0;
            const int t = heavy[h];
            lnp = -a.pairN[t]; lpend = a.pairPos[t];
            lu0 = db.tileUnit0[t]; lnu = db.tileUnit0[t + 1] - lu0;
            return lnu * ((lnp + IGD_HEAVY_PAIRS - 1) / IGD_HEAVY_PAIRS);
        },
        [&](int hh, int it) {
            const int np = __builtin_amdgcn_readlane(lnp, hh), pend = __builtin_amdgcn_readlane(lpend, hh);
            const int u0 = __builtin_amdgcn_readlane(lu0, hh), nu = __builtin_amdgcn_readlane(lnu, hh);
            const int u = u0 + it % nu, sl = it / nu;
            const int p1 = sl * IGD_HEAVY_PAIRS + IGD_HEAVY_PAIRS < np ? sl * IGD_HEAVY_PAIRS + IGD_HEAVY_PAIRS : np;
            const UnitRegs L = load_unit_regs(db.units + u);                      // the same unit in every lane
            const int Lr0 = p1 - sl * IGD_HEAVY_PAIRS, Lr1 = pend - np + p1;      // pairs of the slice, end of the slice
            Raw A;
            issue_unit<false, USE_V, true>(db, a, L, Lr0, Lr1, 0, lane, A);
            compute_unit<false, USE_V, true>(db, a, L, Lr0, Lr1, 0, lane, A, d_hits, d_total);
        });
}
