// engine/scan_direct.hpp -- part of igd_hip.hip (included there once; not a stand-alone header).
// The DIRECT step of a dense, position-sorted batch: the queries' words are never written -- the scan reads the queries.
// ------------------------------------------------------------------------------------------
// What it replaces.  The ordinary sorted step is k_query_bounds (every query read once: 12 bytes in, 4.3 out) ->
// igd_scan_sorted (reads the 4.3) -> k_reduce_slabs.  For one GPU's slab of BASELINE config 4 -- 1.25e7 queries, 130 .. 530
// per visited tile -- the pre-pass costs 47 us beside a scan of 56 .. 94 us.  Here:
//   k_query_bounds<.., BONLY>  the BOUNDS alone: firstQ[t] = the first query of every tile, and the check that the keys
//                     (contig, clamped first tile) never decrease -- starts and contig numbers read, nothing else read,
//                     computed or stored: 21 .. 24 us.  (Tried first: no pass at all -- firstQ[] by bisection per tile,
//                     then by 64-ary searches per 64 / 256 tiles with the last steps in LDS: O(T log Q) probes, but every
//                     probe in a page of its own -- 59 / 46 / 38 us for the benchmark's 188 505 tiles.  LABNOTES.md.)
//   igd_scan_direct   the rank method of igd_scan_sorted (scan_sorted.hpp) with the queries' words derived in registers
//                     from q_qs / q_qe (8 bytes per query, read once by the wave that counts them), the order of the
//                     starts inside a tile verified where they are read (each query belongs to exactly one tile's range
//                     and is checked by that tile's first unit), and the LATER tiles PUSHED instead of pulled: the first
//                     unit of tile t appends the first 64 records that START in tile t+1 to its own sorted start array
//                     (s' + W), so that ONE bisection per query with the unclamped key min(qe - T0, 2W) + 1 places it
//                     among both, and the prefix sum over one more slot yields the later-tile counts
//                     (src/igd_search.c:495-531: a later tile counts the records with tile start <= start < qe; end > qs
//                     holds by construction).  Queries that reach further than the appended records cover -- or beyond
//                     tile t+1 -- are listed for an exact walk of their later tiles (WALK_REST; long ones also WALK_LAST +
//                     the coverage arrays, as k_query_bounds lists them).
// Semantics kept: tile range and clamps src/igd_search.c:459-464, rule NEST :468 (an empty first tile ends the query:
// its first unit is a placeholder with n == 0 and pushes nothing), later tiles :495-531, rule FLAT :635-691.
// Chosen by the host for batches under IGD_HIP_FLAG_SORTED | IGD_HIP_FLAG_SHORT that are dense (>= 28 queries per tile
// on average) over a compact image with power-of-two tiles of <= 2^14 bp; everything else takes the ordinary step.
// Both promises are VERIFIED: disorder marks the batch broken (it adds nothing, igd_hip_sync reports it), a query
// longer than promised only costs time (its later tiles are walked exactly).
// Measured (one MI355X, step = all three launches): slab 0 of 8 / 4 / 2 of config 4's sorted set 112 -> 89, 127 -> 107,
// 156 -> 145 us; the unsharded 1.25e7-query batch (66 per tile) 215 -> 214 us -- its scan is 181 against 149 us there: two
// batches per unit, the second nearly empty, and a sixth slot per unit cost what the pre-pass saves.

#ifndef IGD_D_EXP
#define IGD_D_EXP 0             // measurement only (WRONG counts): 1 no term B, 2 no search in term A, 4 no prefix sums, 8 no flush, 16 no order check, 32 no records of the next tile
#endif
#ifndef IGD_WG_DIR
#define IGD_WG_DIR IGD_WG_RANK  // threads per workgroup / waves per SIMD of igd_scan_direct (2 workgroups per CU)
#define IGD_WPE_DIR IGD_WPE_RANK
#endif
#define IGD_D_SL 512                                    // u16 entries per wave: 320 own starts, 64 appended, 128 x 65535
#define IGD_D_H 392                                     // u32 entries per wave: histogram over positions 0 .. 384
#define IGD_D_WLDS (IGD_D_SL * 2 + IGD_D_H * 4)
#define IGD_D_APP 64                                    // records of the next tile that ride with a tile's first unit
#define WALK_REST 4     // the tiles n1+1 .. min(n2, n1+3) of a query, exactly (listed by igd_scan_direct)

// Per tile, built at open (k_tile_desc): x = index of the first record that STARTS in tile t+1 (its records from number
// `pre` on), y = how many of them ride along (<= IGD_D_APP) | (there are more) << 7 | (tile t+1 exists in the contig) << 8
// | min(w, 15) << 9 | contig << 13, z = contig, w = (last tile of the contig) - (this tile's number in it).
__global__ void k_tile_desc(DbView db, int4 *__restrict__ out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= db.nT) return;
    const int u0 = db.tileUnit0[t];
    const int j = UNIT_J(db.units[u0]);
    // contig of the tile: the one whose base is t - j
    int lo = 0, hi = db.nCtg - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (db.ctgBase[mid] <= t - j) lo = mid; else hi = mid - 1; }
    while (lo + 1 < db.nCtg && db.ctgBase[lo + 1] == t - j && db.ctgNTile[lo] == 0) lo++;    // (contigs without tiles share a base)
    const int rem = db.ctgNTile[lo] - 1 - j;
    int4 d = make_int4(0, 0, lo, rem);
    if (rem > 0) {
        int pre = 0;
        for (int u = db.tileUnit0[t + 1]; u < db.tileUnit0[t + 2]; u++) pre += db.units[u].pre;
        const int cnt = db.tileCnt[t + 1] - pre;
        d.x = (int)(db.tileOff[t + 1] + pre);
        d.y = (cnt < IGD_D_APP ? cnt : IGD_D_APP) | (cnt > IGD_D_APP ? 128 : 0) | 256;
    }
    d.y |= ((rem < 15 ? rem : 15) << 9) | (lo << 13);      // what the scan keeps per unit: one word (nCtg <= 1024)
    out[t] = d;
}

struct DirArgs {
    const int32_t *firstQ;       // [nT + 1] (k_query_bounds<.., BONLY>)
    const int4 *tileD;           // [nT] (k_tile_desc)
    const int32_t *q_qs, *q_qe;
    int32_t *ctl;
    int2 *fix;                   // the batch's exact-walk list (CTL_NFIX)
    int32_t *heavyS, *farList;   // tiles with more than IGD_HEAVY_FIRST queries / units that could overflow a 32-bit counter: the batch's last launch
    int nq, v, epoch, rule, promised;
    int sbCap, wldsBytes;
    u64 *out;                    // slab rows (32-bit)
    u64 *hitsOut, *totalOut;     // the caller's hits[] and batch total (global adds of the last launch)
};
struct DirK { DbView db; DirArgs a; };
// As in igd_scan_sorted (KARG): what the per-unit loop does not need -- the pointers of the descriptor phase and of the rare
// listings -- is read from the kernel-argument segment WHERE it is needed instead of sitting in (or being spilled from)
// scalar registers all along.  KA = false: the argument structs are ordinary memory (the batch's last launch).
#define KARGD(field) karg_load<decltype(((DirK *)0)->field)>((unsigned)offsetof(DirK, field))
#define DA(field) (KA ? KARGD(a.field) : a.field)
#define DD(field) (KA ? KARGD(db.field) : db.field)

// One unit per lane for the wave's next 64 units
struct DRegs { int32_t offLo, n, jf, f0, c0, appOff, appMeta; };
// One unit in flight
struct DRaw {
    uint32_t a[IGD_SLOTS + 1];   // record words; [IGD_SLOTS]: the appended records of the next tile
    int32_t x[IGD_SLOTS + 1];    // dataset numbers (| value << 16)
    int32_t qs, qe;              // the first 64 queries of the unit's range
    int32_t c0, f0, n;           // wave-uniform (the rest of the descriptor is read from the lane that holds it when the unit's turn comes)
};

// the dataset numbers (| value << 16 with the filter) of a unit's records and of the appended ones
template <bool USE_V>
__device__ __forceinline__ void d_load_x(const DbView &db, int32_t (&x)[IGD_SLOTS + 1], unsigned offLo, int n, unsigned appOff, int appN, int lane)
{
    const int vo4 = lane * 4, vo2 = lane * 2;
    const int end = (int)offLo + n, endA = (int)appOff + appN;
    if (USE_V) {
        const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void *)db.pxv, 0, (int)((unsigned)end * 4u), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void *)db.pxv, 0, appN ? (int)((unsigned)endA * 4u) : 0, 0x00020000);
#pragma unroll
        for (int r = 0; r < IGD_SLOTS; r++) x[r] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsX, vo4 + r * 256, (int)(offLo * 4u), 0);
        x[IGD_SLOTS] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsY, vo4, (int)(appOff * 4u), 0);
    } else {
        const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void *)db.px, 0, (int)((unsigned)end * 2u), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void *)db.px, 0, appN ? (int)((unsigned)endA * 2u) : 0, 0x00020000);
#pragma unroll
        for (int r = 0; r < IGD_SLOTS; r++) x[r] = (int)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsX, vo2 + r * 128, (int)(offLo * 2u), 0);
        x[IGD_SLOTS] = (int)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsY, vo2, (int)(appOff * 2u), 0);
    }
}

template <bool USE_V>
__device__ __forceinline__ void d_issue(const DbView &db, const DirArgs &a, const DRegs &L, int kk, bool valid, int lane, DRaw &R)
{
    const int kq = kk & 63;
    int c0 = __builtin_amdgcn_readlane(L.c0, kq);
    if (!valid) c0 = 0;
    const int f0 = __builtin_amdgcn_readlane(L.f0, kq);
    const int n = c0 ? __builtin_amdgcn_readlane(L.n, kq) : 0;
    const int jf = __builtin_amdgcn_readlane(L.jf, kq);
    const int meta = __builtin_amdgcn_readlane(L.appMeta, kq);
    const int appN = (c0 && (jf & 1)) ? (meta & 127) : 0;
    R.c0 = c0; R.f0 = f0; R.n = n;
    const unsigned offLo = (unsigned)__builtin_amdgcn_readlane(L.offLo, kq);
    const unsigned appOff = (unsigned)__builtin_amdgcn_readlane(L.appOff, kq);
    const int vo4 = lane * 4;
    const int end = (int)offLo + n, endA = (int)appOff + appN;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)db.pse, 0, (int)((unsigned)end * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void *)db.pse, 0, appN ? (int)((unsigned)endA * 4u) : 0, 0x00020000);
#pragma unroll
    for (int r = 0; r < IGD_SLOTS; r++) R.a[r] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsA, vo4 + r * 256, (int)(offLo * 4u), 0);
    R.a[IGD_SLOTS] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsB, vo4, (int)(appOff * 4u), 0);
    d_load_x<USE_V>(db, R.x, offLo, n, appOff, appN, lane);
    // the first 64 queries of the tile.  ONE descriptor per array for the whole kernel (its end = the batch's end): lanes past
    // the tile's last query read the queries that follow -- every use is masked by the tile's count -- and an unvisited unit
    // (c0 = 0) is pushed out of range: no memory access.
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void *)a.q_qs, 0, a.nq * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc((void *)a.q_qe, 0, a.nq * 4, 0x00020000);
    const int so = c0 ? f0 * 4 : 0x7FFFFF00;
    R.qs = (int)__builtin_amdgcn_raw_buffer_load_b32(rsS, vo4, so, 0);
    R.qe = (int)__builtin_amdgcn_raw_buffer_load_b32(rsE, vo4, so, 0);
}

// The batch was found out of order: nothing it has counted or is still going to count is added (k_reduce_slabs and the
// last launch's bodies look at the mark first).
template <bool KA>
__device__ __forceinline__ void d_mark_broken(const DirArgs &a, int lane)
{
    int32_t *ctl = DA(ctl);
    if (lane == 0) { ctl[CTL_UNSORTED] = a.epoch; if (a.promised) ctl[CTL_BROKEN] = a.epoch; }
}

// The batch's last launch cannot list a query for the exact walk (that list is being walked by then): the few queries of
// its units that need one -- see the listing in d_compute -- have their tiles j0 .. j1 walked here, on the exact arrays, by
// the whole wave (lob: :510-511; the first tile n1 has none).
template <bool USE_V>
__device__ __forceinline__ void d_walk_exact(const DbView &db, const DirArgs &a, int c, int qs, int qe, int n1, int j0, int j1, int lane)
{
    const int base = db.ctgBase[c];
    u64 found = 0;
    for (int j = j0; j <= j1; j++) {
        const int t = base + j;
        const int tcnt = db.tileCnt[t];
        if (tcnt == 0) continue;
        const int lob = j == n1 ? INT_MIN : db.tileBd[t];
        const int64_t toff = db.tileOff[t];
        for (int rec0 = 0; rec0 < tcnt; rec0 += IGD_WAVE) {
            const int i = rec0 + lane;
            const bool ok = i < tcnt;
            const int st = ok ? db.start[toff + i] : INT_MAX, en = ok ? db.end[toff + i] : INT_MIN;
            if (__builtin_amdgcn_readfirstlane(st) >= qe) break;               // sorted: nothing further
            bool hit = (st < qe) & (st >= lob) & (en > qs);
            if (USE_V) hit = hit && db.value[toff + (ok ? i : 0)] >= a.v;
            found += __popcll(__ballot(hit));
            if (hit) atomicAdd(&a.hitsOut[db.idx[toff + i]], 1ull);
        }
    }
    if (a.totalOut && lane == 0 && found) atomicAdd(a.totalOut, found);
}

// One unit against the queries [f0, f0 + c0) of its tile.  GLOBAL: counts go to the caller's 64-bit hits[] with global
// atomics (the batch's last launch: slices of very dense tiles, units that could overflow a 32-bit counter), else to the
// workgroup's 32-bit LDS counters.  prevQ: the start of the query before f0 when that one belongs to the same tile's range
// (a slice), else INT_MIN.
template <bool USE_V, bool GLOBAL, bool KA>
__device__ __forceinline__ void d_compute(const DbView &db, const DirArgs &a, const DRegs &L, int kk, int lane, DRaw &R, unsigned int *hits32,
                                          unsigned short *sl, unsigned int *hist, unsigned short *sb, int prevQ, bool *appDirty)
{
    const int c0 = R.c0;
    if (c0 <= 0) return;
    const int un = R.n, f0 = R.f0, jf = __builtin_amdgcn_readlane(L.jf, kk);
    const int meta = __builtin_amdgcn_readlane(L.appMeta, kk);
    const int W = db.nbp, sh = db.shift;
    const int j = jf >> 4;
    const bool first = jf & 1;                           // the tile's first unit: it checks, lists and pushes
    const int T0 = (int)((unsigned)j * (unsigned)W);
    const bool appMore = (meta & 128) != 0, nextTile = (meta & 256) != 0;
    const int ctg = (meta >> 13) & 1023;
    int rem = (meta >> 9) & 15;                          // tiles left in the contig behind this one, capped at 15 ...
    if (rem == 15) rem = DD(ctgNTile)[ctg] - 1 - j;      // ... the true number where it matters (a query of 16+ tiles)
    const bool dead = a.rule == IGD_HIP_RULE_NEST && un == 0;       // (:468; every unit of an empty tile is its placeholder)
    if (!first && un == 0) return;
    if (!first && dead) return;
    int cnt[IGD_SLOTS + 1];
#pragma unroll
    for (int r = 0; r <= IGD_SLOTS; r++) cnt[r] = 0;
    int32_t (&X)[IGD_SLOTS + 1] = R.x;
    // ---- the unit's starts, and behind them (first unit) the starts of the next tile's first records, + W ----
    // Lanes past the unit's last record hold W + 1 there: above every own start (<= W), not above any appended one (>= W + 1),
    // so the array stays sorted and a query that ends inside the tile is placed at the unit's end as before.
    const bool push = first && !dead && nextTile && !(IGD_D_EXP & 32);
    const unsigned padv = push ? (unsigned)W + 1u : 65535u;
    // (s' = the inverted low half of the record word; a lane without a record holds 0 there, i.e. 65535 inverted: one `not`
    // and one 16-bit `min` with the padding value per slot)
    const unsigned short padh = (unsigned short)padv;
#pragma unroll
    for (int r = 0; r < IGD_SLOTS; r++) {
        const unsigned short sv = (unsigned short)~R.a[r];
        sl[r * IGD_WAVE + lane] = sv < padh ? sv : padh;
    }
    if (push) {
        const unsigned sA = 65535u - (R.a[IGD_SLOTS] & 0xFFFFu);          // s' in [1, W] (these records start in their tile)
        sl[IGD_SLOTS * IGD_WAVE + lane] = (unsigned short)(R.a[IGD_SLOTS] != 0u ? (unsigned)W + sA : 65535u);
    } else if (*appDirty) sl[IGD_SLOTS * IGD_WAVE + lane] = 65535;       // (what the unit before left there)
    *appDirty = push;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // a query is served by the appended records iff every record of tile t+1 that it can count is among them
    int covKey = 0;                                      // keys up to this one are
    if (push) covKey = appMore ? W + (int)(65535u - ((unsigned)__builtin_amdgcn_readlane((int)R.a[IGD_SLOTS], IGD_D_APP - 1) & 0xFFFFu)) : 2 * W + 1;   // (more than ride along: all IGD_D_APP lanes hold one)
    // ... in terms of d = qe - T0: a query of the first unit with W < d <= dlim is served (when the appended records are all
    // there are and tile t+1 is the contig's last, whatever its end: n2 is clamped, :463)
    const int dlim = !first ? INT_MAX : (rem == 0 || dead) ? INT_MAX : appMore ? covKey - 1 : rem == 1 ? INT_MAX : 2 * W;
    const int capF = push ? 2 * W : W;                   // the usual batch's keys: min(d, capF) + 1
    const bool inLds = c0 < a.sbCap;
    int nFirst = 0, nBack = 0;
    bool disorder = false;
    int carryQ = prevQ;
    const int vo4 = lane * 4;
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void *)a.q_qs, 0, a.nq * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc((void *)a.q_qe, 0, a.nq * 4, 0x00020000);
    const bool rankAny = un != 0 || push;                // (a placeholder with nothing to push only checks its queries)
    // One batch of 64 queries.  (A lambda called for the first batch in straight-line code and for the others in a loop: a
    // load inside a loop makes the compiler wait for ALL loads in flight at the loop's head -- the NEXT unit's records, just
    // issued, included -- so a first batch inside the loop made every unit wait out a memory round trip: 233 against 150 us.)
    auto batch = [&](const int p, const int qs_, const int qe_) {
        const int idx = p + lane;
        const int a_ = qs_ - T0, d_ = qe_ - T0;
        if (first && !(IGD_D_EXP & 16)) {                // the order of the starts, where they are read
            // lane i gets lane i-1's start, lane 0 keeps the last start of the batch before (DPP wave_shr:1 -- one instruction)
            const int pq = __builtin_amdgcn_update_dpp(carryQ, qs_, 0x138, 0xf, 0xf, false);
            disorder = disorder || (idx < c0 && qs_ < pq);
            carryQ = __builtin_amdgcn_readlane(qs_, IGD_WAVE - 1);
        }
        // ---- the usual batch: queries of this tile, none inverted, that end inside it or are served by the records that ride
        // along -- nothing to mask or list ----
        const bool there = idx < c0;
        if (__ballot(there && !((unsigned)a_ < (unsigned)W && d_ >= (a_ > 1 ? a_ : 1) && d_ <= dlim)) == 0ull) {
            if (rankAny) {
                const int pos = (IGD_D_EXP & 2) ? (d_ & 255) : lds_lower_bound(sl, (d_ < capF ? d_ : capF) + 1);
                if (there) atomicAdd(&hist[pos], 1u);
            }
            nFirst += __popcll(__ballot(there));
            if (inLds && there) sb[idx] = (unsigned short)(a_ + 1);
            return;
        }
        bool inTile = there && (unsigned)a_ < (unsigned)W;
        // a query that starts outside the contig's tiles sits in the range of its first / last tile and counts nothing (:462);
        // anywhere else it is out of order
        bool front = there && a_ < 0;
        const bool back = there && a_ >= W;
        if (j == 0) {                                    // (a start above -W lies in tile 0 by C division, :459)
            const bool neg0 = front && a_ > -W;
            inTile = inTile || neg0;
            front = front && !neg0;
        }
        if (first) disorder = disorder || (front && j != 0) || (back && rem != 0);
        int qs1 = a_ + 1;
        qs1 = qs1 < 1 ? 1 : qs1;                         // (a start above -W lies in tile 0 by C division)
        const int qe1 = (d_ < W ? d_ : W) + 1;
        const bool good = inTile && d_ >= 1 && qe1 >= qs1;
        // later tiles (:495-531): n2 = min((qe - 1) / W, last tile of the contig)
        int span = (d_ - 1) >> sh;
        span = span > rem ? rem : span;
        const bool reach = good && span >= 1 && !dead;
        int key = qe1;
        bool far = false;
        if (first) {
            const int k2 = (d_ < 2 * W ? d_ : 2 * W) + 1;
            const bool served = reach && span == 1 && k2 <= covKey;
            far = reach && !served;
            key = served ? k2 : key;
            const bool wfirst = inTile && d_ <= 0;       // reaches back over the tile's start: exact starts needed (WALK_FIRST)
            const bool wlong = reach && span >= IGD_SHORT_TILES;
            const unsigned long long mf = __ballot(far), ml = __ballot(wlong), mw = __ballot(wfirst);
            if (GLOBAL) {
                // (the last launch: the exact-walk list is being walked already -- these queries' tiles are walked right here)
                for (unsigned long long m2 = mf | mw; m2; m2 &= m2 - 1) {
                    const int src = __builtin_ctzll(m2);
                    const int s_ = __builtin_amdgcn_readlane(qs_, src), e_ = __builtin_amdgcn_readlane(qe_, src);
                    const int sp = __builtin_amdgcn_readlane(span, src);
                    if ((mw >> src) & 1) d_walk_exact<USE_V>(db, a, ctg, s_, e_, j, j, j, lane);
                    else d_walk_exact<USE_V>(db, a, ctg, s_, e_, j, j + 1, j + sp, lane);
                }
            } else
            if (mf | ml | mw) {
                // (rare under the caller's promise of short queries: one returning atomic per wave and batch)
                const int nf = __popcll(mf), nl = __popcll(ml), nw = __popcll(mw);
                int32_t *ctl = DA(ctl);
                int2 *fix = DA(fix);
                int at = 0;
                if (lane == 0) at = atomicAdd(&ctl[CTL_NFIX + (a.epoch & 1)], nf + nl + nw);
                at = __builtin_amdgcn_readfirstlane(at);
                const unsigned long long below = (1ull << lane) - 1ull;
                const int q = f0 + idx;
                if (far) fix[at + __popcll(mf & below)] = make_int2(q, WALK_REST | (ctg << 4));
                if (wlong) {
                    fix[at + nf + __popcll(ml & below)] = make_int2(q, WALK_LAST | (ctg << 4));
                    if (span > IGD_SHORT_TILES) {        // tiles n1+4 .. n2-1 are covered from end to end (cover_tiles)
                        const int nT = DD(nT);
                        const int g0 = DD(ctgBase)[ctg] + j, ta = g0 + IGD_SHORT_TILES, tb = g0 + span;
                        int32_t *diff = DD(cov) + (size_t)(a.epoch & 1) * IGD_COV_LEN(nT), *coarse = diff + nT + 2;
                        atomicAdd(&diff[ta], 1); atomicAdd(&diff[tb], -1);
                        if ((ta >> IGD_COV_SHIFT) != (tb >> IGD_COV_SHIFT)) { atomicAdd(&coarse[ta >> IGD_COV_SHIFT], 1); atomicAdd(&coarse[tb >> IGD_COV_SHIFT], -1); }
                        ctl[CTL_COV + (a.epoch & 1)] = a.epoch;
                    }
                }
                if (wfirst) fix[at + nf + nl + __popcll(mw & below)] = make_int2(q, WALK_FIRST | (ctg << 4));
            }
        }
        if (rankAny) {
            // term A: the query's end among the sorted starts
            const int pos = lds_lower_bound(sl, key);
            if (good) atomicAdd(&hist[pos], 1u);
        }
        nFirst += __popcll(__ballot(good));
        nBack += __popcll(__ballot(back));
        // the exceptions: queries of this tile that are inverted or reach back over its start.  They stay in the ordered list
        // of starts that term B bisects and are taken out again one by one; an inverted query's own hits are added.
        unsigned long long x = __ballot(inTile && !good);
        if (x && un != 0) {
            const int wq = d_ >= 1 ? (int)((unsigned)(65536 - qe1) | ((unsigned)qs1 << 16)) : (int)IGD_NEVER;
            while (x) {
                const int src = __builtin_ctzll(x);
                x &= x - 1;
                const int s_ = __builtin_amdgcn_readlane(qs1, src), w_ = __builtin_amdgcn_readlane(wq, src);
#pragma unroll
                for (int r = 0; r < IGD_SLOTS; r++) {
                    igd_u16x2 rec, qw;
                    __builtin_memcpy(&rec, &R.a[r], 4);
                    __builtin_memcpy(&qw, &w_, 4);
                    const igd_u16x2 mx = __builtin_elementwise_max(rec, qw);
                    uint32_t mxw;
                    __builtin_memcpy(&mxw, &mx, 4);
                    cnt[r] += (s_ > (int)(R.a[r] >> 16) ? 1 : 0) + (mxw == R.a[r] ? 1 : 0);
                }
            }
        }
        if (inLds && there) sb[idx] = (unsigned short)(inTile ? (qs1 > 65535 ? 65535 : qs1) : (front ? 1 : 65535));
    };
    {
        // the next 64 queries are on their way while these are searched (past the batch's last query: no access)
        int qsN = (int)__builtin_amdgcn_raw_buffer_load_b32(rsS, vo4, (f0 + IGD_WAVE) * 4, 0);
        int qeN = (int)__builtin_amdgcn_raw_buffer_load_b32(rsE, vo4, (f0 + IGD_WAVE) * 4, 0);
        batch(0, R.qs, R.qe);
        for (int p = IGD_WAVE; p < c0; p += IGD_WAVE) {
            const int qs_ = qsN, qe_ = qeN;
            qsN = (int)__builtin_amdgcn_raw_buffer_load_b32(rsS, vo4, (f0 + p + IGD_WAVE) * 4, 0);
            qeN = (int)__builtin_amdgcn_raw_buffer_load_b32(rsE, vo4, (f0 + p + IGD_WAVE) * 4, 0);
            batch(p, qs_, qe_);
        }
    }
    if (__ballot(disorder)) d_mark_broken<KA>(a, lane);
    if (un == 0 && !push) return;                        // (a placeholder that had nothing to push: it only checked its queries)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // term B: #{q of this tile: qs' > e'} = c0 - #{qs' <= e'}: every record bisects the tile's ordered query starts
    if (un != 0 && !(IGD_D_EXP & 1)) {
        const int levels = 32 - __builtin_clz((unsigned)c0), top = 1 << levels;
        if (inLds) {
            for (int k = c0 + lane; k < top - 1; k += IGD_WAVE) sb[k] = 65535;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            typedef __attribute__((address_space(3))) const unsigned short *lds_u16;
            const unsigned sb0 = (unsigned)(size_t)(lds_u16)sb;
            unsigned Q[IGD_SLOTS], E[IGD_SLOTS];
            const unsigned q0 = sb0 + (unsigned)top - 2u;
#pragma unroll
            for (int r = 0; r < IGD_SLOTS; r++) { Q[r] = q0; E[r] = R.a[r] >> 16; }
            int vq[IGD_SLOTS];
            for (int S = top >> 1; S > 1; S >>= 1) {
                const int up = S, dn = -S;
#pragma unroll
                for (int r = 0; r < IGD_SLOTS; r++) vq[r] = (int)*(lds_u16)(size_t)Q[r];
#pragma unroll
                for (int r = 0; r < IGD_SLOTS; r++) Q[r] += (unsigned)(vq[r] <= (int)E[r] ? up : dn);
            }
#pragma unroll
            for (int r = 0; r < IGD_SLOTS; r++) vq[r] = (int)*(lds_u16)(size_t)Q[r];
            const unsigned zero = sb0 + 2u * (unsigned)c0;
#pragma unroll
            for (int r = 0; r < IGD_SLOTS; r++) cnt[r] += ((int)(Q[r] - zero) >> 1) + (vq[r] <= (int)E[r] ? 1 : 0);
        } else {
            int pos[IGD_SLOTS];
#pragma unroll
            for (int r = 0; r < IGD_SLOTS; r++) pos[r] = 0;
            for (int step = top >> 1; step > 0; step >>= 1) {
                int vq[IGD_SLOTS];
#pragma unroll
                for (int r = 0; r < IGD_SLOTS; r++) {
                    const int at = pos[r] + step - 1;
                    vq[r] = at < c0 ? a.q_qs[f0 + at] : INT_MAX;
                }
                // qs' <= e' on the raw starts: a start before the tile (qs' = 1) is below every e', one beyond it above
#pragma unroll
                for (int r = 0; r < IGD_SLOTS; r++) pos[r] += ((long long)vq[r] <= (long long)(int)(R.a[r] >> 16) + T0 - 1) ? step : 0;
            }
#pragma unroll
            for (int r = 0; r < IGD_SLOTS; r++) cnt[r] -= c0 - pos[r];
        }
    }
    // term A: #{q: p_q <= i} = inclusive prefix sum of the histogram over the positions; slot IGD_SLOTS = the appended records,
    // which only the queries that reach them (p_q beyond the unit's 320 positions) count
    int carry = 0;
#pragma unroll
    for (int r = 0; r <= IGD_SLOTS; r++) {
        if (r == IGD_SLOTS && !push) { cnt[r] = 0; break; }              // (no query is placed beyond the unit's own positions then)
        const int h = (int)hist[r * IGD_WAVE + lane];
        hist[r * IGD_WAVE + lane] = 0u;
        const int inc = (IGD_D_EXP & 4) ? h : wave_inclusive_sum(h);
        cnt[r] += nFirst - (carry + inc) + (r < IGD_SLOTS ? nBack : 0);
        carry += __builtin_amdgcn_readlane(inc, 63);
        bool keep = true;
        if (USE_V) { keep = (X[r] >> 16) >= a.v; X[r] &= 0xFFFF; }
        if (R.a[r] == 0u || !keep) cnt[r] = 0;
    }
    if (USE_V && !push) X[IGD_SLOTS] = 0;
    if (lane == 0) hist[IGD_SLOTS * IGD_WAVE] = 0u;      // (position 320 of a unit without appended records / 384)
    if (push && lane == 0) hist[(IGD_SLOTS + 1) * IGD_WAVE] = 0u;
    if (GLOBAL) {
        int t = 0;
#pragma unroll
        for (int r = 0; r <= IGD_SLOTS; r++) {
            const int c = cnt[r];
            t += c;
            if (c) atomicAdd((u64 *)((char *)a.hitsOut + ((size_t)X[r] << 3)), (u64)(unsigned)c);
        }
        if (a.totalOut) {
            for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o);
            if (lane == 0 && t) atomicAdd(a.totalOut, (u64)(unsigned)t);
        }
    } else {
#pragma unroll
        for (int r = 0; r <= IGD_SLOTS; r++) {
            if (IGD_D_EXP & 8) { asm volatile("" ::"v"(cnt[r]), "v"(X[r])); continue; }
            if (cnt[r]) atomicAdd(hits32 + X[r], (unsigned)cnt[r]);
        }
    }
}

// The DIRECT scan kernel: 2 workgroups of 12 waves per CU (6 waves per SIMD, like the full build of igd_scan_sorted).
template <bool USE_V>
__global__ __launch_bounds__(IGD_WG_DIR) __attribute__((amdgpu_waves_per_eu(IGD_WPE_DIR, IGD_WPE_DIR))) void igd_scan_direct(DirK K)
{
    const DbView &db = K.db;
    const DirArgs &a = K.a;
    if (__builtin_amdgcn_readfirstlane(KARGD(a.ctl)[CTL_UNSORTED]) == a.epoch) return;    // the bounds kernel found the keys out of order
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nFiles = KARGD(db.nFiles);
    const size_t hitBytes = ((size_t)nFiles * 4 + 15) & ~(size_t)15;
    unsigned int *hits = (unsigned int *)smem;
    unsigned short *sl = (unsigned short *)(smem + hitBytes + (size_t)wid * (size_t)KARGD(a.wldsBytes));
    unsigned int *hist = (unsigned int *)(sl + IGD_D_SL);
    unsigned short *sb = (unsigned short *)(hist + IGD_D_H);
    for (int f = threadIdx.x; f < nFiles; f += IGD_WG_DIR) hits[f] = 0u;
    for (int k = lane; k < IGD_D_SL; k += IGD_WAVE) sl[k] = 65535;
    for (int k = lane; k < IGD_D_H; k += IGD_WAVE) hist[k] = 0u;
    __syncthreads();
    constexpr int wavesPerWG = IGD_WG_DIR / IGD_WAVE;
    const int gwave = (int)blockIdx.x * wavesPerWG + wid;
    const int nwaves = gridDim.x * wavesPerWG;
    unsigned spent = 0u;
    const unsigned budget = 0xFFFFFFFFu / (unsigned)wavesPerWG;
    DRaw A, B;
    bool appDirty = false;
    for (int ub = gwave; ub < db.nUnits; ub += nwaves * IGD_WAVE) {
        DRegs L;
        L.offLo = L.n = L.jf = L.f0 = L.c0 = L.appOff = L.appMeta = 0;
        {
            const long long mi = (long long)ub + (long long)lane * nwaves;
            if (mi < db.nUnits) {
                const int32_t *firstQ = KARGD(a.firstQ);
                const UnitRegs u = load_unit_regs(KARGD(db.units) + mi);
                L.offLo = u.offLo; L.n = u.n; L.jf = u.jf;
                L.f0 = firstQ[u.tile];
                L.c0 = firstQ[u.tile + 1] - L.f0;
                const int4 d = KARGD(a.tileD)[u.tile];
                L.appOff = d.x; L.appMeta = d.y;
                // only a tile's first unit sees its queries when the tile holds no record (its placeholder) ...
                if (u.n == 0 && !(u.jf & 1)) L.c0 = 0;
                // ... a range that runs backwards is disorder; it is reported by the unit that finds it
                if (L.c0 < 0) { int32_t *ctl = KARGD(a.ctl); ctl[CTL_UNSORTED] = a.epoch; if (a.promised) ctl[CTL_BROKEN] = a.epoch; L.c0 = 0; }
                // a tile with very many queries is shared out over all waves of the batch's last launch, in slices
                if (L.c0 > IGD_HEAVY_FIRST) {
                    if (u.jf & 1) KARGD(a.heavyS)[atomicAdd(&KARGD(a.ctl)[CTL_NHEAVYS + (a.epoch & 1)], 1)] = u.tile;
                    L.c0 = 0;
                }
            }
        }
        // The workgroup's counters are 32-bit.  A unit adds at most (its queries) x (its records + the appended ones) to any one of
        // them; every wave keeps the sum of that bound over its units against its share of 2^32, and the unit that would take
        // it beyond -- in practice: none -- is left, whole, to the batch's last launch (64-bit global adds).
        unsigned long long m = __ballot(L.c0 != 0);
        {
            unsigned long long mm = m;
            while (mm) {
                const int k = __builtin_ctzll(mm);
                mm &= mm - 1;
                const unsigned bound = (unsigned)__builtin_amdgcn_readlane(L.c0, k) * (unsigned)(__builtin_amdgcn_readlane(L.n, k) + IGD_D_APP);
                if (bound > budget - spent) {
                    if (lane == 0) KARGD(a.farList)[atomicAdd(&KARGD(a.ctl)[CTL_NFAR + (a.epoch & 1)], 1)] = ub + k * nwaves;
                    if (lane == k) L.c0 = 0;
                    m &= ~(1ull << k);
                } else spent += bound;
            }
        }
        int ka = -1, kb = -1;
        if (m) { ka = __builtin_ctzll(m); m &= m - 1; }
        if (m) { kb = __builtin_ctzll(m); m &= m - 1; }
        d_issue<USE_V>(db, a, L, ka < 0 ? 0 : ka, ka >= 0, lane, A);
        while (ka >= 0) {
            d_issue<USE_V>(db, a, L, kb < 0 ? 0 : kb, kb >= 0, lane, B);
            d_compute<USE_V, false, true>(db, a, L, ka, lane, A, hits, sl, hist, sb, INT_MIN, &appDirty);
            ka = -1;
            if (m) { ka = __builtin_ctzll(m); m &= m - 1; }
            d_issue<USE_V>(db, a, L, ka < 0 ? 0 : ka, ka >= 0, lane, A);
            if (kb >= 0) d_compute<USE_V, false, true>(db, a, L, kb, lane, B, hits, sl, hist, sb, INT_MIN, &appDirty);
            kb = -1;
            if (m) { kb = __builtin_ctzll(m); m &= m - 1; }
        }
    }
    __syncthreads();
    {
        const int nf = KARGD(db.nFiles);
        unsigned int *row32 = (unsigned int *)KARGD(a.out) + (size_t)blockIdx.x * nf;
        for (int f = threadIdx.x; f < nf; f += IGD_WG_DIR) row32[f] = hits[f];
    }
}

// What igd_scan_direct left to the batch's last launch: (unit, slice of IGD_HEAVY_SLICE queries) items of the tiles it listed
// as too dense for one wave, and the units whose counts might not fit the workgroup's 32-bit counters -- dealt to all
// waves, added to the caller's hits[] and the batch total with global atomics.  `wsm`: this wave's LDS area (IGD_D_WLDS).
template <bool USE_V>
__device__ __forceinline__ void direct_tail_body(const DbView &db, const DirArgs &a, unsigned char *wsm, int gwave, int nwaves, int lane, int ctlv)
{
    if (__builtin_amdgcn_readlane(ctlv, CTL_UNSORTED) == a.epoch) return;
    int nH = __builtin_amdgcn_readlane(ctlv, CTL_NHEAVYS + (a.epoch & 1));
    const int nF = __builtin_amdgcn_readlane(ctlv, CTL_NFAR + (a.epoch & 1));
    if (nH == 0 && nF == 0) return;
    if (nH > IGD_HEAVYS_MAX) nH = IGD_HEAVYS_MAX;
    unsigned short *sl = (unsigned short *)wsm;
    unsigned int *hist = (unsigned int *)(sl + IGD_D_SL);
    unsigned short *sb = (unsigned short *)(hist + IGD_D_H);
    for (int k = lane; k < IGD_D_SL; k += IGD_WAVE) sl[k] = 65535;
    for (int k = lane; k < IGD_D_H; k += IGD_WAVE) hist[k] = 0u;
    bool appDirty = false;
    DirArgs b = a;
    b.sbCap = 0;                                         // (no LDS array of query starts in the last launch: term B bisects q_qs[])
    auto run = [&](int u, int f0, int c0, int prevQ) {
        const UnitRegs ur = load_unit_regs(db.units + u);                   // the same unit in every lane
        DRegs L;
        L.offLo = ur.offLo; L.n = ur.n; L.jf = ur.jf; L.f0 = f0; L.c0 = c0;
        const int4 d = b.tileD[ur.tile];
        L.appOff = d.x; L.appMeta = d.y;
        if (ur.n == 0 && !(ur.jf & 1)) return;
        DRaw A;
        d_issue<USE_V>(db, b, L, 0, true, lane, A);
        d_compute<USE_V, true, false>(db, b, L, 0, lane, A, nullptr, sl, hist, sb, prevQ, &appDirty);
    };
    int lf0 = 0, lc0 = 0, lu0 = 0, lnu = 0;
    deal_items(nH, gwave, nwaves, lane,
        [&](int h) {
            lf0 = lc0 = lu0 = lnu = 0;
            if (h < 0) return 0;
            const int tl = b.heavyS[h];
            lf0 = b.firstQ[tl]; lc0 = b.firstQ[tl + 1] - lf0;
            lu0 = db.tileUnit0[tl]; lnu = db.tileUnit0[tl + 1] - lu0;
            return lnu * ((lc0 + IGD_HEAVY_SLICE - 1) / IGD_HEAVY_SLICE);
        },
        [&](int hh, int it) {
            const int f0 = __builtin_amdgcn_readlane(lf0, hh), c0 = __builtin_amdgcn_readlane(lc0, hh);
            const int u0 = __builtin_amdgcn_readlane(lu0, hh), nu = __builtin_amdgcn_readlane(lnu, hh);
            const int u = u0 + it % nu, sc = it / nu;
            const int fs = f0 + sc * IGD_HEAVY_SLICE;
            const int cs = c0 - sc * IGD_HEAVY_SLICE < IGD_HEAVY_SLICE ? c0 - sc * IGD_HEAVY_SLICE : IGD_HEAVY_SLICE;
            run(u, fs, cs, sc > 0 ? b.q_qs[fs - 1] : INT_MIN);
        });
    for (int i = gwave; i < nF; i += nwaves) {
        const int u = __builtin_amdgcn_readfirstlane(b.farList[i]);
        const int tl = __builtin_amdgcn_readfirstlane(db.units[u].tile);
        const int f0 = b.firstQ[tl];
        run(u, f0, b.firstQ[tl + 1] - f0, INT_MIN);
    }
}
#undef DA
#undef DD
