// engine/scan_sorted.hpp -- part of igd_hip.hip (included there once; not a stand-alone header).
// igd_scan_sorted: the merge join (pairwise and rank builds) -- the dominant kernel
// ------------------------------------------------------------------------------------------
// igd_scan_sorted: the merge join over the compact image -- the dominant kernel of a position-sorted
// batch.  Same unit / slot / summary scheme as igd_scan_tiles above, but fed by k_query_bounds'
// per-query words, which take everything that depends on one QUERY out of the per-unit path:
//   * the queries whose FIRST tile is the unit's tile are read as ready-made compare words (qw0, one
//     bounds-checked buffer load per 64 of them); nothing is computed per candidate;
//   * the queries of the up-to-3 tiles before it are looked at only if k_query_bounds marked the tile
//     (spill[]: some query covers it as a later tile), 29 % of the units on the benchmark;
//   * the visiting rule (NEST: an empty first tile ends the query) is already folded into the words.
// Two ways to count a unit's overlaps, chosen per unit:
//   pairwise  (few queries per tile): every query that passes a slot's summary word is broadcast and
//             compared with the slot's 64 records: v_readlane, v_pk_max_u16, v_cmp, v_addc;
//   rank      (>= IGD_DENSE_MIN first-tile queries): O((R + Q) log) instead of O(R Q).  For queries with
//             qs <= qe a record is missed for exactly one of two reasons -- it starts at or after the
//             query's end (A) or ends at or before its start (B) -- so per record
//                 hits = #queries - #{q: qe' <= s'} - #{first-tile q: qs' > e'}.
//             A: every query bisects the unit's sorted starts (staged in LDS) for p = #{records: s' < qe'}
//                and adds 1 to a histogram at p; a prefix sum over the records gives #{q: p_q <= i};
//             B: the first-tile queries of a tile are consecutive in the caller's array and ordered by
//                start, so every record bisects q_qs[] for its own end.
//             Queries that are inverted (qe < qs) or masked out (IGD_NEVER) would be counted twice or
//             wrongly: they are taken out of both terms and compared pairwise.
#ifndef IGD_DENSE_MIN
#define IGD_DENSE_MIN 32
#endif
#ifndef IGD_LEAN_SMALL
#define IGD_LEAN_SMALL 1        // the lean build's units of <= 64 / <= 192 records take the pairwise path over one / three slots
#endif

#define IGD_WLDS_S 512                                  // u16 entries per wave: the unit's sorted s' (+ sentinels)
#define IGD_WLDS_H 328                                  // u32 entries per wave: histogram over record positions 0..320
#define IGD_WLDS_BYTES (IGD_WLDS_S * 2 + IGD_WLDS_H * 4)

struct SortArgs {
    const int32_t *firstQ;       // [nT+1] first query of each tile
    const int32_t *spill;        // [nT]   == epoch: some query covers the tile as a later tile
    const int32_t *qw0, *later;  // per-query first-tile words; later-tile words, compacted per later block (k_query_bounds)
    const int32_t *lpos;         // [nT+1] entries of its later block before query firstQ[t]
    const int2 *laterHdr;        // per later block: (entries, last tile covered as a later tile)
    int lbShift;                 // log2(queries per later block): 8, 10 or 12 (k_query_bounds<VEC, ., WGT>)
    const int32_t *q_qs;         // the caller's query starts (rank method: exceptions, and tiles with more queries than sbCap)
    const int32_t *ctl;
    int nq, v, epoch, mode, rule;
    int sbCap, wldsBytes;        // rank method: u16 entries of a wave's sorted-query-start array / bytes of a wave's LDS area
    int32_t *ctlw, *heavyS;      // control words (writable) and the list of tiles left to heavy_sorted_body
    int32_t *farList;            // [nUnits] units the lean build leaves to far_units_body (unit number | its tile is in heavyS << 31)
    int noList;                  // a later pass of a windowed batch: heavy tiles and far units are left out as in the first pass, which listed them
    int tailHistOff;             // the last launch's per-workgroup u64 counters for exact walks and coverage: byte offset in its dynamic LDS (< 0: none)
    u64 *out;                    // slab [grid][nFiles] (LDS counters) or the global hits[]
};

// The two merge-join kernels take ONE argument struct, and read everything their inner loop does not need -- a dozen
// pointers of the rarer paths -- from the kernel-argument segment WHERE it is needed (KARG): with 8 waves per SIMD a
// wave has 80 scalar registers, and values loaded at kernel entry would sit in (or be spilled from) them all along.
struct SortK { DbView db; SortArgs a; u64 *hitsOut, *totalOut; };
typedef const __attribute__((address_space(4))) char *karg_ptr;
template <typename T>
__device__ __forceinline__ T karg_load(unsigned off)
{
    karg_ptr p = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));                           // opaque: the scalar load below stays in the branch it is written in
    return *(const __attribute__((address_space(4))) T *)(p + off);
}
#define KARG(field) karg_load<decltype(((SortK *)0)->field)>((unsigned)offsetof(SortK, field))

// A unit's descriptor and query ranges, one unit per lane (broadcast with v_readlane when its turn comes)
// la / ln: the tile's later-tile candidates -- the later[] entries of the queries of the (up to) 3 tiles before it -- as
// k_query_bounds' lpos[] places them: nA words from index la on, then nB words from the start of the block that holds
// query f0 (the range crossed a block boundary).  ln = nA | nB << 13 | (global tile & 3) << 26 | far << 28; far: more
// than 64 words, or more than one boundary crossed -- such a unit walks the blocks (far_later); 0: no candidates.
#define IGD_LN_A(ln) ((ln) & 8191)
#define IGD_LN_B(ln) (((ln) >> 13) & 8191)
#define IGD_LN_G2(ln) (((ln) >> 26) & 3)
#define IGD_LN_FAR(ln) (((ln) >> 28) & 1)
struct SRegs { int32_t offLo, offHi, n, jf, w[IGD_SLOTS], f0, c0, la, ln; };
typedef __attribute__((address_space(3))) u64 igd_lds_u64;      // a 64-bit counter in the workgroup's LDS (the batch's last launch)

struct Raw2 {
    uint32_t a[IGD_SLOTS];       // s' | e' << 16 (inverted s', see k_pack_units)
    int32_t x[IGD_SLOTS];        // idx (| value << 16)
    // The unit's candidates form ONE list: its nl later-tile entries first (not far: nl = nA + nB <= 64), then its c0
    // first-tile queries; the first 64 of the list come with the records:
    int32_t q;                   // lanes nl ..: first-tile words (already un-inverted; IGD_NEVER where there is none)
    int32_t lw;                  // lanes 0 .. nl-1: later[] entries (0 where there is none)
    int32_t c0, ln, f0, n;       // wave-uniform (SGPRs): the unit's query ranges and record count, kept from the issue
};

// Branch-free on purpose (see issue_unit): the same number of loads whatever the unit looks like, so that
// the compiler counts them (s_waitcnt vmcnt(N)) and the next unit's loads stay in flight during a compare.
// A unit nobody asks about (or kk past the wave's last unit) gets descriptors of size 0: no memory access.
// Descriptors: the hardware range check covers voffset + soffset + immediate, so every array keeps ONE base for the
// whole kernel (loop-invariant SGPRs) and a unit only moves soffset (= its first byte) and num_records (= its end):
// two scalar instructions per array instead of a 64-bit address computation.  BIG = the image is beyond the 4 GiB a
// 32-bit soffset reaches (> 2^30 records): per-unit base addresses, as igd_scan_tiles does.
// PLAIN (the lean build's own loop): kk < 64 names a lane that holds a unit of the round or zeros (a round is IGD_ROUND =
// 62 units: the two lanes past it stay empty, so the loop's look-ahead needs neither a `valid` flag nor a wrap), and no
// far unit gets here (they are listed for far_units_body) -- a dozen scalar instructions per unit that a kernel bound by
// instruction issue does not have to spare.
template <bool USE_V, bool BIG, bool PLAIN = false>
__device__ __forceinline__ void s_issue(const DbView &db, const SortArgs &a, const SRegs &L, int kk, bool valid, int lane, Raw2 &R)
{
    const int kq = PLAIN ? kk : kk & 63;
    int c0 = __builtin_amdgcn_readlane(L.c0, kq), ln = __builtin_amdgcn_readlane(L.ln, kq);
    if (!PLAIN && !valid) { c0 = 0; ln = 0; }
    const int f0 = __builtin_amdgcn_readlane(L.f0, kq);
    // (SRegs::n is 0 already for a unit nobody asks about: set where the round's descriptors are read)
    const int n = (PLAIN || (c0 | ln)) ? __builtin_amdgcn_readlane(L.n, kq) : 0;
    R.c0 = c0; R.ln = ln; R.f0 = f0; R.n = n;
    const unsigned offLo = (unsigned)__builtin_amdgcn_readlane(L.offLo, kq);
    const int vo4 = lane * 4, vo2 = lane * 2;
    if (BIG) {
        const int64_t off = (int64_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(L.offHi, kq) << 32) | offLo);
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)(db.pse + off), 0, n * 4, 0x00020000);
        if (USE_V) {
            const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void *)(db.pxv + off), 0, n * 4, 0x00020000);
#pragma unroll
            for (int r = 0; r < IGD_SLOTS; r++) {
                R.a[r] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsA, vo4, r * 256, IGD_NT_AUX);
                R.x[r] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsX, vo4, r * 256, IGD_NT_AUX);
            }
        } else {
            const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void *)(db.px + off), 0, n * 2, 0x00020000);
#pragma unroll
            for (int r = 0; r < IGD_SLOTS; r++) {
                R.a[r] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsA, vo4, r * 256, IGD_NT_AUX);
                R.x[r] = (int)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsX, vo2, r * 128, IGD_NT_AUX);
            }
        }
    } else {
        const int end = (int)offLo + n;                   // < 2^30 records: byte offsets fit 32 bits
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)db.pse, 0, (int)((unsigned)end * 4u), 0x00020000);
        if (USE_V) {
            const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void *)db.pxv, 0, n ? (int)((unsigned)(end + IGD_CHUNK) * 4u) : 0, 0x00020000);   // (see below)
#pragma unroll
            for (int r = 0; r < IGD_SLOTS; r++) {
                R.a[r] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsA, vo4 + r * 256, (int)(offLo * 4u), IGD_NT_AUX);
                R.x[r] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsX, vo4 + r * 256, (int)(offLo * 4u), IGD_NT_AUX);
            }
        } else {
            // (the dataset numbers are NOT cut off at the unit's end: the lanes past it -- whose record words are 0, so they
            // count nothing -- then name the datasets of the records that follow instead of all naming dataset 0.  Their
            // "+ 0" LDS atomics queued up for that ONE counter: on a database of small tiles -- 30 records: four and a half
            // of a unit's five slots empty -- the waves spent 81 % of their cycles waiting for the LDS.  The arrays are
            // padded by a chunk; a unit nobody asks about still touches no memory)
            const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void *)db.px, 0, n ? (int)((unsigned)(end + IGD_CHUNK) * 2u) : 0, 0x00020000);
#pragma unroll
            for (int r = 0; r < IGD_SLOTS; r++) {
                R.a[r] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsA, vo4 + r * 256, (int)(offLo * 4u), IGD_NT_AUX);
                R.x[r] = (int)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsX, vo2 + r * 128, (int)(offLo * 2u), IGD_NT_AUX);
            }
        }
    }
    // The first 64 entries of the unit's candidate list (see Raw2): its later-tile entries in lanes 0 .. nl-1 -- one run
    // of later[] words, or two when the candidate range crosses a block boundary; none for the 71 % of the units no query
    // reaches as a later tile -- and behind them the first-tile words.  Lanes outside either run are out of the buffers'
    // range: they read 0 (no memory access at all when a run is empty).
    const bool isFar = !PLAIN && IGD_LN_FAR(ln);
    const int nA = isFar ? 0 : IGD_LN_A(ln), nB = isFar ? 0 : IGD_LN_B(ln);
    const int nl = nA + nB;
    const int b0 = c0 < IGD_WAVE - nl ? c0 : IGD_WAVE - nl;
    // (PLAIN, image below 4 GiB: the end of the first batch's words was worked out with the round's descriptors -- SRegs::offHi)
    const int qEnd = (PLAIN && !BIG) ? __builtin_amdgcn_readlane(L.offHi, kq) : (f0 + b0) * 4;
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void *)a.qw0, 0, qEnd, 0x00020000);
    if (PLAIN && nl == 0) {                                                            // 71 % of the units: no later[] words to ask for
        R.q = ~(int)__builtin_amdgcn_raw_buffer_load_b32(rs0, vo4, f0 * 4, 0);
        R.lw = 0;
        return;
    }
    int voq = vo4;
    if (nl) voq = lane < nl ? 0x7FFFFF00 : vo4 - nl * 4;                               // (the later-tile lanes: far out of range)
    R.q = ~(int)__builtin_amdgcn_raw_buffer_load_b32(rs0, voq, f0 * 4, 0);             // before / past the tile's queries: ~0 = IGD_NEVER
    const int la = nl ? __builtin_amdgcn_readlane(L.la, kq) : 0;
    const int aB = (int)((unsigned)(f0 >> a.lbShift) << a.lbShift);                    // first entry of the block that holds query f0
    const int thr = nB ? nA : IGD_WAVE;                                                // lanes from here on read the second run
    const __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc((void *)a.later, 0, (nB ? aB + nB : la + nA) * 4, 0x00020000);
    R.lw = (int)__builtin_amdgcn_raw_buffer_load_b32(rsL, vo4 + (lane < thr ? la : aB - nA) * 4, 0, 0);
}

// The queries of one batch of <= 64 candidates (word `P0` per lane, IGD_NEVER where there is none) against the
// unit: per slot, the summary word picks the queries that can hit it at all (one compare for all 64), and only
// those are broadcast and compared.  cnt[r] += hit; no exec masking, no LDS.
template <bool ASM>
__device__ __forceinline__ void match_words(const Raw2 &R, int (&cnt)[IGD_SLOTS], const uint32_t (&W)[IGD_SLOTS], int P0)
{
#if IGD_ASM_MATCH && !(IGD_EXP & 2)
    if (ASM) {                                           // (match_slot_asm, scan_tiles.hpp: the loop written out)
#pragma unroll
        for (int r = 0; r < IGD_SLOTS; r++) match_slot_asm<false>(cnt[r], W[r], P0, R.a[r], 0ull);
        return;
    }
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
        unsigned long long m = __ballot(mww == W[r]);
#if IGD_EXP & 2
        asm volatile("" ::"v"(R.a[r]), "s"(m));
        m = 0;
#endif
        while (m) {
            const int src = __builtin_ctzll(m);
            m &= ~(1ull << src);                         // s_bitset0_b64
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

// later-tile word (k_query_bounds: later[]) -> compare word for this tile (IGD_NEVER when the query does not reach it)
// qe' of a later-tile word in this tile (meaningful where `covers`).  g2 = the tile's global number & 3; an entry is
// never 0, and a load outside the candidates' run returns 0.
__device__ __forceinline__ int later_end(int nbp, int e, int g2, int deadk, bool inRange, bool &covers)
{
    const int k = (g2 - (e >> 20)) & 3;                  // tiles between the query's first tile and this one (1..3)
    // rule NEST: a query whose FIRST tile is empty is dead (src/igd_search.c:468); deadk bit k = tile j-k is empty
    covers = inRange && e != 0 && k != 0 && ((e >> 18) & 3) >= k && !((deadk >> k) & 1);
    const int rel = (e & 0x3FFFF) - __mul24(k, nbp);     // qe - T for this tile
    return (rel < nbp ? rel : nbp) + 1;                  // qe'
}
__device__ __forceinline__ int later_word(int nbp, int e, int g2, int deadk, bool inRange, bool &covers)
{
    const int rel = later_end(nbp, e, g2, deadk, inRange, covers);
    return covers ? (int)((unsigned)(65536 - rel) | (1u << 16)) : (int)IGD_NEVER;
}

// #{entries of the wave's sorted s' array that are < key}: 9 dependent LDS reads; entries past the unit's
// records hold 65535 (> every key), so no bounds are needed
__device__ __forceinline__ int lds_lower_bound(const unsigned short *sl, int key)
{
    // carried as the LDS byte address of sl[pos]: a step is read (immediate offset), compare, select, add
    typedef __attribute__((address_space(3))) const unsigned short *lds_u16;
    const unsigned base = (unsigned)(size_t)(lds_u16)sl;
    unsigned P = base;
#pragma unroll
    for (int step = 256; step > 0; step >>= 1) P += ((int)*(lds_u16)(size_t)(P + 2u * (unsigned)(step - 1)) < key) ? 2u * (unsigned)step : 0u;
    return (int)((P - base) >> 1);
}

// A load the compiler's wait-count bookkeeping does not see (it is waited for right here).  For the seldom-taken
// branches of the full build's compare phase: a tracked load inside a loop makes the compiler wait for ALL vector loads
// in flight at the loop's head (s_waitcnt vmcnt(0)) -- the next unit's records included -- on every pass, taken or
// not, and the dense batches this build is for run every unit through those loops.
__device__ __forceinline__ int load_now(const int32_t *p)
{
    int v;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// Where the later-tile candidates of a tile lie in later[] (SRegs::la / ln): the entries of the queries [fl, f0) of the
// (up to) lb tiles before it, from lpos[] and the block headers.  Only for a tile some query reaches (spill[]).
template <bool KA>
__device__ __forceinline__ void later_range(const SortArgs &a, int tile, int lb, int fl, int f0, int &la, int &ln)
{
    la = 0; ln = 0;
    if (fl >= f0) return;
    const int32_t *lpos = KA ? KARG(a.lpos) : a.lpos;
    const int sh = a.lbShift;
    const int pA = lpos[tile - lb], pB = lpos[tile];
    const int bA = fl >> sh, bB = f0 >> sh;
    int nA = pB - pA, nB = 0;
    if (bA != bB) { nA = (KA ? KARG(a.laterHdr) : a.laterHdr)[bA].x - pA; nB = pB; }
    const int far = (bB - bA > 1 || nA + nB > IGD_WAVE) ? 1 : 0;
    if (nA | nB | far) {
        la = (bA << sh) + pA;
        ln = nA | (nB << 13) | ((tile & 3) << 26) | (far << 28);
    }
}

// A unit whose later-tile candidates do not come with its records (IGD_LN_FAR: more than 64 entries -- tiles behind a
// very dense one -- or a candidate range that crosses more than one block boundary) walks them here: nA entries from
// index la on, every entry of the blocks in between (their number: laterHdr[]), the first nB of the block that holds
// query f0.  FN(entries) is called per batch of <= 64 (0 in the lanes past a run's end).
template <bool KA, typename FN>
__device__ __forceinline__ void far_later(const SortArgs &a, int la, int ln, int f0, int lane, FN fn)
{
    const int sh = a.lbShift;
    const int bA = la >> sh, bB = f0 >> sh;
    const int32_t *later = KA ? KARG(a.later) : a.later;
    const int2 *hdr = KA ? KARG(a.laterHdr) : a.laterHdr;
    for (int b = bA; b <= bB; b++) {
        const int from = b == bA ? la : (b << sh);
        int cnt;
        if (b == bA) cnt = IGD_LN_A(ln);
        else if (b == bB) cnt = IGD_LN_B(ln);
        else cnt = KA ? __builtin_amdgcn_readfirstlane(load_now(&hdr[b].x)) : hdr[b].x;
        for (int p = 0; p < cnt; p += IGD_WAVE) {
            const int at = from + (p + lane < cnt ? p + lane : 0);
            const int e = KA ? load_now(later + at) : later[at];
            fn(p + lane < cnt ? e : 0);
        }
    }
}

// RANK = false: the lean build for batches that are sparse on average (the host decides by queries per tile): no rank
// method in the kernel at all -- its registers would burden the pairwise path, which is what such a batch runs --
// and a tile that is dense after all goes to heavy_sorted_body from IGD_LEAN_FIRST first-tile queries on.
// FEW: the database has one file (1) / up to eight (2): builds of their own, so that the usual one pays nothing for them;
// 3: a window of a database with more files than LDS counters (lanes without hits stay out as with 2: half the lanes have none)
template <bool USE_V, bool CNT32, bool RANK, bool LDSH = false, int FEW = 0>
__device__ __forceinline__ void s_compute(const DbView &db, const SortArgs &a, const SRegs &L, int kk, int lane, Raw2 &R,
                                          u64 *hits, unsigned short *sl, unsigned int *hist, unsigned short *sb, bool rankOK,
                                          u64 *found = nullptr, unsigned *spent = nullptr, unsigned budget = 0u, igd_lds_u64 *lhits = nullptr)
{
    const int c0 = R.c0, ln = R.ln;
    const int un = R.n;
    if (un == 0) return;                                 // placeholder of an empty tile, or nobody asks about this unit (s_issue: n = 0 then)
    const int f0 = R.f0;
    int cnt[IGD_SLOTS];
#pragma unroll
    for (int r = 0; r < IGD_SLOTS; r++) cnt[r] = 0;
    int nLater = 0;                                      // covering queries for which this is NOT the first tile
    bool keep[IGD_SLOTS];                                // record passes the value filter (USE_V)
#pragma unroll
    for (int r = 0; r < IGD_SLOTS; r++) {
        keep[r] = true;
        if (USE_V) {
            keep[r] = (R.x[r] >> 16) >= a.v;             // arithmetic shift: the signed 16-bit value
            R.x[r] &= 0xFFFF;
        }
    }
    // later tiles: how many entries lead the candidate list (0: none, or a `far` unit, which walks them separately), the
    // low bits of the unit's global tile number and which of the 3 tiles before it are empty (rule NEST)
    const bool far = RANK && IGD_LN_FAR(ln) && !(IGD_EXP & 8);   // (the lean build lists its far units: far_units_body)
    const int nl = ((RANK && IGD_LN_FAR(ln)) || (IGD_EXP & 8) != 0) ? 0 : IGD_LN_A(ln) + IGD_LN_B(ln);
    int g2 = 0, deadk = 0;
    if (ln) {
        g2 = IGD_LN_G2(ln);
        deadk = a.rule == IGD_HIP_RULE_NEST ? (__builtin_amdgcn_readlane(L.jf, kk) & 14) : 0;
    }
    const int nE = nl + c0;                              // entries of the candidate list
#if IGD_EXP & 4
    {
#pragma unroll
        for (int r = 0; r < IGD_SLOTS; r++) asm volatile("" ::"v"(R.a[r]), "v"(R.x[r]));
        asm volatile("" ::"v"(R.q));
        return;
    }
#endif
    if (!RANK || !(rankOK && c0 >= IGD_DENSE_MIN)) {
        // ---- pairwise ---------------------------------------------------------------------------
        uint32_t W[IGD_SLOTS];                           // the slots' summary words (read here: the rank method has no use for them)
#pragma unroll
        for (int r = 0; r < IGD_SLOTS; r++) W[r] = (uint32_t)__builtin_amdgcn_readlane(L.w[r], kk);
        if (USE_V) {
#pragma unroll
            for (int r = 0; r < IGD_SLOTS; r++) if (!keep[r]) R.a[r] = 0u;    // the word nothing matches
        }
        int w = R.q;
        if (nl) {                                        // the first nl lanes of the first batch hold later-tile entries
            bool covers;
            const int lw = later_word(db.nbp, R.lw, g2, deadk, lane < nl, covers);
            nLater = __popcll(__ballot(covers));
            w = lane < nl ? lw : w;
        }
        // the first batch came with the records; most units have no other (one scalar test), a dense tile is a chain of
        // them, each on its way while the one before it is compared
        if (nE > IGD_WAVE) {
            int wn = (IGD_WAVE + lane < nE) ? ~a.qw0[f0 + IGD_WAVE + lane - nl] : (int)IGD_NEVER;
            match_words<IGD_ASM_MATCH == 1 || !RANK>(R, cnt, W, w);
            for (int p = IGD_WAVE; p < nE; p += IGD_WAVE) {
                w = wn;
                wn = (p + IGD_WAVE + lane < nE) ? ~a.qw0[f0 + p + IGD_WAVE + lane - nl] : (int)IGD_NEVER;
                match_words<IGD_ASM_MATCH == 1 || !RANK>(R, cnt, W, w);
            }
        } else if (nE > 0) match_words<IGD_ASM_MATCH == 1 || !RANK>(R, cnt, W, w);
        if (far)
            far_later<RANK>(a, __builtin_amdgcn_readlane(L.la, kk), ln, f0, lane, [&](int e) {
                bool covers;
                const int lw = later_word(db.nbp, e, g2, deadk, true, covers);
                nLater += __popcll(__ballot(covers));
                match_words<IGD_ASM_MATCH == 1 || !RANK>(R, cnt, W, lw);
            });
        // records that start before the tile (s' = 0, low half 65535) were matched by every "later tile"
        // query, none of which may count them (the reference's tS skip, :510-511)
        if (nLater != 0) {
#pragma unroll
            for (int r = 0; r < IGD_SLOTS; r++) cnt[r] -= (R.a[r] & 0xFFFFu) == 0xFFFFu ? nLater : 0;
        }
    } else {
        // ---- rank ---------------------------------------------------------------------------------
#define IGD_TILE_START ((int)((unsigned)(__builtin_amdgcn_readlane(L.jf, kk) >> 4) * (unsigned)db.nbp))   /* only the seldom-taken branches need it */
#pragma unroll
        for (int r = 0; r < IGD_SLOTS; r++)
            sl[r * IGD_WAVE + lane] = (unsigned short)(65535u - (R.a[r] & 0xFFFFu));   // lanes past the unit: 65535
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const bool inLds = c0 < a.sbCap;           // the tile's query starts fit the wave's LDS array (a power of two)
        int nFirst = 0;
        // One batch of <= 64 entries of the candidate list, entry p + lane in each lane: word w for a first-tile query
        // (IGD_NEVER where the lane has none), later[] entry e for a later-tile one (first batch only: WITH_LATER).
        // Term A: every covering query bisects the unit's starts with its end and adds 1 to the histogram there.
        // the exceptions of a batch (lanes x: first-tile queries whose word is IGD_NEVER or inverted; idx: which of the tile's
        // queries, -1 in lanes that hold none): qs2 <- their true start, their contribution to term B undone, an inverted
        // query's own hits added
        auto batchFix = [&](const int w, int &qs2, const int idx, unsigned long long x) {
            int t = load_now(KARG(a.q_qs) + (idx >= 0 ? f0 + idx : f0)) - IGD_TILE_START + 1;   // = qs' for a query of this tile; beyond it: clamped
            if (idx < 0) t = 65535;
            t = t < 1 ? 1 : (t > 65535 ? 65535 : t);
            qs2 = t;
            while (x) {
                const int src = __builtin_ctzll(x);
                x &= ~(1ull << src);
                const int s_ = __builtin_amdgcn_readlane(qs2, src), wq = __builtin_amdgcn_readlane(w, src);
#pragma unroll
                for (int r = 0; r < IGD_SLOTS; r++) {
                    igd_u16x2 rec, qw;
                    __builtin_memcpy(&rec, &R.a[r], 4);
                    __builtin_memcpy(&qw, &wq, 4);
                    const igd_u16x2 mx = __builtin_elementwise_max(rec, qw);
                    uint32_t mxw;
                    __builtin_memcpy(&mxw, &mx, 4);
                    cnt[r] += (s_ > (int)(R.a[r] >> 16) ? 1 : 0) + (mxw == R.a[r] ? 1 : 0);   // undo term B; an inverted query's own hits
                }
            }
        };
        auto batchA = [&](const int w, const int e, const int p, const bool withLater) {
            const int idx = p + lane - nl;               // which of the tile's own queries (first batch: < 0 in the later-tile lanes)
            const bool there = idx >= 0 && idx < c0;
            const int qe2 = 65536 - (w & 0xFFFF);
            int qs2 = (int)((unsigned)w >> 16);
            const bool good = (unsigned)w != IGD_NEVER && qe2 >= qs2;   // (IGD_NEVER wherever the lane has no first-tile query)
            int key = qe2;
            bool add = good;
            if (withLater) {
                bool covers;
                const int le = later_end(db.nbp, e, g2, deadk, lane < nl, covers);
                key = covers ? le : key;
                add = add || covers;
                nLater += __popcll(__ballot(covers));
            }
            const int pos = (IGD_EXP & 128) ? (key & 255) : lds_lower_bound(sl, key);
            if (add) atomicAdd(&hist[pos], 1u);
            nFirst += __popcll(__ballot(good));
            // the exceptions: masked-out (IGD_NEVER) or inverted queries.  They stay in the ordered list of starts that
            // term B bisects -- with their TRUE start, so that it stays ordered -- and are taken out again one by one
            const unsigned long long x = __ballot(there && !good);
            if (x) batchFix(w, qs2, there ? idx : -1, x);
            if (inLds && there) sb[idx] = (unsigned short)qs2;
        };
        // The batches after the first are fetched one ahead: the load of batch k + 1 is issued before batch k is searched
        // and its word first touched after (530 queries per tile -- one GPU's slab of an 8-GPU job -- are 9 batches, and
        // a load waited for on the spot made each of them a memory round trip).  Two batches per pass of the loop, so
        // that no loaded word is carried around it.
        // A batch that lies wholly inside the tile's own queries (no later-tile lanes, no lanes past the last query -- all
        // batches but the first and the last of a tile with hundreds of queries): nothing to mask, nothing to select
        auto batchIn = [&](const int w, const int p) {
            const int idx = p + lane - nl;
            const int qe2 = 65536 - (w & 0xFFFF);
            int qs2 = (int)((unsigned)w >> 16);
            const bool good = (unsigned)w != IGD_NEVER && qe2 >= qs2;
            const int pos = (IGD_EXP & 128) ? (qe2 & 255) : lds_lower_bound(sl, qe2);
            if (good) atomicAdd(&hist[pos], 1u);
            const unsigned long long gm = __ballot(good);
            nFirst += __popcll(gm);
            if (gm != ~0ull) batchFix(w, qs2, idx, ~gm);   // masked-out or inverted queries: rare
            if (inLds) sb[idx] = (unsigned short)qs2;
        };
        // The batches after the first are fetched one ahead -- the load of batch k + 1 is issued before batch k is searched
        // and its word first touched after (530 queries per tile -- one GPU's slab of an 8-GPU job -- are 9 batches, and a
        // load waited for on the spot made each of them a memory round trip) -- by a bounds-checked load that costs no
        // vector instruction: per-lane offset lane * 4, the batch's first word in the scalar offset, and the lanes past the
        // tile's last query read 0 = ~IGD_NEVER.  (The later-tile entries all sit in the first batch: nl <= 64.)
        if (nE <= IGD_WAVE) batchA(R.q, R.lw, 0, nl != 0);   // (nothing to fetch ahead)
        else {
            const __amdgpu_buffer_rsrc_t rsq = __builtin_amdgcn_make_buffer_rsrc((void *)a.qw0, 0, (f0 + c0) * 4, 0x00020000);
            const int vo4 = lane * 4;
            int wn = ~(int)__builtin_amdgcn_raw_buffer_load_b32(rsq, vo4, (f0 + IGD_WAVE - nl) * 4, 0);
            batchA(R.q, R.lw, 0, nl != 0);
            int p = IGD_WAVE;
            for (; p + IGD_WAVE <= nE; p += IGD_WAVE) {
                const int w = wn;
                wn = ~(int)__builtin_amdgcn_raw_buffer_load_b32(rsq, vo4, (f0 + p + IGD_WAVE - nl) * 4, 0);   // (past the end: size-0 access)
                batchIn(w, p);       // (two of these side by side, their searches advancing in the same steps: no faster)
            }
            if (p < nE) batchA(wn, 0, p, false);
        }
        if (far)
            far_later<true>(a, __builtin_amdgcn_readlane(L.la, kk), ln, f0, lane, [&](int e) {
                bool covers;
                const int key = later_end(db.nbp, e, g2, deadk, true, covers);
#if IGD_EXP & 512
                nLater += __popcll(__ballot(covers));
                return;
#endif
                const int pos = (IGD_EXP & 128) ? (key & 255) : lds_lower_bound(sl, key);
                if (covers) atomicAdd(&hist[pos], 1u);
                nLater += __popcll(__ballot(covers));
            });
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // term B: #{first-tile q: qs' > e'} = c0 - #{qs' <= e'}: every record bisects the tile's ordered query starts
        {
            const int levels = 32 - __builtin_clz((unsigned)c0), top = 1 << levels;   // top = 2^levels > c0 >= IGD_DENSE_MIN, c0 < 2^30
            int pos[IGD_SLOTS];
            bool inLdsDone = false;
#pragma unroll
            for (int r = 0; r < IGD_SLOTS; r++) pos[r] = 0;
            if (IGD_EXP & 64) {
            } else
            if (inLds) {
                // the array is padded to top - 1 entries with 65535 (> every e'): no bounds in the loop, whose five
                // chains of dependent LDS reads then run side by side
                for (int k = c0 + lane; k < top - 1; k += IGD_WAVE) sb[k] = 65535;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // Positions are carried as LDS byte addresses and the steps are written out with their strides as immediate
                // offsets, entered at the tile's first stride (top / 2): read, compare, select, add per chain and step.
                // (Tried: term A's search of the last batch of queries advanced in the same steps -- six reads in flight
                // instead of five, nine dependent steps fewer per unit -- and it was no faster.)
                typedef __attribute__((address_space(3))) const unsigned short *lds_u16;
                const unsigned sb0 = (unsigned)(size_t)(lds_u16)sb;
                // The probes are carried as LDS byte addresses: with stride S the probe is entry pos + S - 1; taking the step
                // moves the next probe (stride S / 2) up by S / 2 entries, not taking it moves it down by S / 2 -- so a step is
                // read, compare, select +-S bytes, add, in a loop with a wave-uniform trip count.  (The steps used to be
                // written out behind a switch over the tile's first stride: every case label was a merge point for which
                // the compiler copied the five positions -- 55 moves per unit, 20 of them in the cases a tile of 66 queries
                // skips: 135 vector instructions where 7 steps need 105.)
                unsigned Q[IGD_SLOTS], E[IGD_SLOTS];
                const unsigned q0 = sb0 + (unsigned)top - 2u;            // entry top / 2 - 1
#pragma unroll
                for (int r = 0; r < IGD_SLOTS; r++) { Q[r] = q0; E[r] = R.a[r] >> 16; }
                int vq[IGD_SLOTS];
                for (int S = top >> 1; S > 1; S >>= 1) {                 // S = byte distance to the next probe
                    const int up = S, dn = -S;
#pragma unroll
                    for (int r = 0; r < IGD_SLOTS; r++) vq[r] = (int)*(lds_u16)(size_t)Q[r];
#pragma unroll
                    for (int r = 0; r < IGD_SLOTS; r++) Q[r] += (unsigned)(vq[r] <= (int)E[r] ? up : dn);   // compare, select, add
                }
#pragma unroll
                for (int r = 0; r < IGD_SLOTS; r++) vq[r] = (int)*(lds_u16)(size_t)Q[r];   // the last probe is the position itself
                // cnt -= c0 - pos with pos = (Q - sb0) / 2 + (last probe taken), in one go: subtract, halve, add with carry
                const unsigned zero = sb0 + 2u * (unsigned)c0;
#pragma unroll
                for (int r = 0; r < IGD_SLOTS; r++) cnt[r] += ((int)(Q[r] - zero) >> 1) + (vq[r] <= (int)E[r] ? 1 : 0);
                inLdsDone = true;
            } else {                                     // more queries than the LDS array holds: bisect q_qs[] itself
                const int32_t *q_qs = KARG(a.q_qs);
                const int T = IGD_TILE_START;
                for (int step = top >> 1; step > 0; step >>= 1) {
                    int vq[IGD_SLOTS];
#pragma unroll
                    for (int r = 0; r < IGD_SLOTS; r++) {
                        const int at = pos[r] + step - 1;
                        vq[r] = at < c0 ? q_qs[f0 + at] : INT_MAX;
                    }
#pragma unroll
                    for (int r = 0; r < IGD_SLOTS; r++) pos[r] += vq[r] <= (int)(R.a[r] >> 16) + T - 1 ? step : 0;   // qs' <= e'
                }
            }
            if (!inLdsDone) {
#pragma unroll
                for (int r = 0; r < IGD_SLOTS; r++) cnt[r] -= c0 - pos[r];
            }
        }
        // term A: #{q: p_q <= i} = inclusive prefix sum of the histogram over the record positions
        int carry = 0;
#pragma unroll
        for (int r = 0; r < IGD_SLOTS; r++) {
            const int h = (int)hist[r * IGD_WAVE + lane];
            hist[r * IGD_WAVE + lane] = 0u;
            const int inc = (IGD_EXP & 256) ? h : wave_inclusive_sum(h);
            const bool prefix = (R.a[r] & 0xFFFFu) == 0xFFFFu;   // starts before the tile: later-tile queries do not count it
            cnt[r] += nFirst + (prefix ? 0 : nLater) - (carry + inc);
            carry += __builtin_amdgcn_readlane(inc, 63);
            if (R.a[r] == 0u || !keep[r]) cnt[r] = 0;    // no record here (loads past the unit's end return 0; a record word has e' >= 1) / fails the value filter
        }
        if (lane == 0) hist[IGD_SLOTS * IGD_WAVE] = 0u;  // p = 320: queries beyond every record of a full unit
#undef IGD_TILE_START
    }
    // CNT32 (the workgroup's LDS counters are 32-bit): one 32-bit LDS atomic per slot, for all lanes -- a lane without
    // hits adds 0 (lanes past the unit: to counter 0), which costs LDS lanes but none of the compare / exec-mask
    // instructions that skipping them would.  No counter can wrap: a unit adds at most (its candidate queries) x (its
    // records) to any of them, every wave keeps the sum of that bound over its units, and the unit that would take the
    // wave beyond its share of 2^32 (and any `far` unit, whose candidates are not counted beforehand) adds to the caller's
    // 64-bit hits[] with global atomics instead -- as do the builds without LDS counters (one atomic per record hit).
    if (FEW == 3) {                                      // a window of files (see igd_scan_tiles, WIN): the others' records count nothing in this pass
#pragma unroll
        for (int r = 0; r < IGD_SLOTS; r++) {
            const unsigned x = (unsigned)R.x[r] - (unsigned)db.fileLo;
            const bool in = x < (unsigned)db.nFiles;
            cnt[r] = in ? cnt[r] : 0;
            R.x[r] = in ? (int)x : 0;
        }
    }
    bool direct = !CNT32;
    // (the lean build needs no guard: the host has bounded what its units -- <= IGD_LEAN_FIRST + 64 candidates each, the
    // far ones are not its own -- can add up to: launch_scan)     // the lean build: the host has bounded what its units -- <= IGD_LEAN_FIRST + 64
                                                         // candidates each, far ones apart -- can add up to (launch_scan)
    if (CNT32 && RANK) {
        // candidates of the unit: its own queries + its later-tile entries (a far unit: at most the two runs it knows
        // plus every entry of the blocks between them)
        long long cand = nE;
        if (IGD_LN_FAR(ln)) {
            const int bA = __builtin_amdgcn_readlane(L.la, kk) >> a.lbShift, bB = f0 >> a.lbShift;
            cand += IGD_LN_A(ln) + IGD_LN_B(ln) + (bB - bA > 1 ? (long long)(bB - bA - 1) << a.lbShift : 0);
        }
        const long long bound = cand * un;                // (cand <= 2^25, un <= 320)
        direct = bound > (long long)(budget - *spent);
        if (!direct) *spent += (unsigned)bound;
    }
    if (!direct && FEW == 1) {
        // one file: every lane names counter 0 -- 64 LDS atomics on one address are 64 passes (10^6 queries against a
        // database of one 2 x 10^7-record file: scan kernel 212 us, 54 with sixteen files) -- so the wave adds once
        int s = 0;
#pragma unroll
        for (int r = 0; r < IGD_SLOTS; r++) s += cnt[r];
        s = wave_inclusive_sum(s);
        if (lane == IGD_WAVE - 1 && s) atomicAdd((unsigned int *)hits, (unsigned)s);
    } else
    if (!direct) {
        constexpr bool few = FEW != 0;                   // a handful of files: lanes without hits stay out (see below)
#pragma unroll
        for (int r = 0; r < IGD_SLOTS; r++) {
#if IGD_EXP & 1
            asm volatile("" ::"v"(cnt[r]), "v"(R.x[r]));
            continue;
#endif
            // lean build: every lane adds, 0 included (no compare / exec masking -- that build is bound by instruction issue);
            // full build: lanes without hits stay out -- the lanes past a unit's end all name counter 0, and the dozens
            // of them in a unit's last slot would queue up for ONE address in an LDS the rank method keeps busy
            if (!RANK && !few && !USE_V) {
                // (the dataset number is a zero-extended 16-bit load the compiler masks again before it scales it: one
                // v_mad_u32_u16 does both)
                unsigned off;
                asm("v_mad_u32_u16 %0, %1, 4, 0" : "=v"(off) : "v"(R.x[r]));
                atomicAdd((unsigned int *)((char *)hits + off), (unsigned)cnt[r]);
            } else
            if ((!RANK && !few) || cnt[r]) atomicAdd((unsigned int *)hits + R.x[r], (unsigned)cnt[r]);
        }
    } else
    if (LDSH && lhits) {
        // the skew valves of the batch's last launch: into the workgroup's 64-bit LDS counters, flushed once per workgroup.
        // (Straight to hits[] these adds were the valves' cost: a piled-up tile's ~2000 slices all add to the SAME few hundred
        // counters -- a dozen lines of hits[] -- and requests for one line are served one after the other: 58 of the 125 us of
        // 10^6 queries inside one tile were far_units_body's slices waiting for each other's atomics.)
#pragma unroll
        for (int r = 0; r < IGD_SLOTS; r++) {
            const int c = cnt[r];
            if (c) (void)__hip_atomic_fetch_add(lhits + R.x[r], (u64)(unsigned)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    } else {
        u64 *gh = CNT32 ? KARG(hitsOut) : hits;          // (CNT32: the slab rows hold 32-bit counts, this unit's go to hits[] itself)
        if (CNT32) found = KARG(totalOut);
#pragma unroll
        for (int r = 0; r < IGD_SLOTS; r++) {
            const int c = cnt[r];
            if (c) atomicAdd((u64 *)((char *)gh + ((size_t)R.x[r] << 3)), (u64)(unsigned)c);
        }
    }
    if (found) {                                         // skew valve: the batch total is kept by the caller of this unit
        int t = 0;
#pragma unroll
        for (int r = 0; r < IGD_SLOTS; r++) t += cnt[r];
        for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o);
        if (lane == 0 && t) atomicAdd(found, (u64)(unsigned)t);
    }
}

// The lean build's unit of at most NS x 64 records (NS = 1 or 3 of the IGD_SLOTS slots): the pairwise path of s_compute over
// the slots that can hold a record at all.  A unit costs the same instructions whether it is full or holds 30 records --
// five summary words read out, five slots tested, five counter adds -- and on databases of small tiles (a clustered one: half of
// the units of the roadmap-scale database hold 129 .. 192 records, a small one's units 32 on average) that fixed cost is the
// kernel: both the vector and the ONE scalar unit of a CU are busy with it.  (LDS counters, 32-bit, more than 8 files: the
// usual lean build; everything else takes s_compute.)
template <bool USE_V, int NS>
__device__ __forceinline__ void s_compute_small(const DbView &db, const SortArgs &a, const SRegs &L, int kk, int lane, Raw2 &R, u64 *hits)
{
    const int c0 = R.c0, ln = R.ln, f0 = R.f0;
    int cnt[NS];
#pragma unroll
    for (int r = 0; r < NS; r++) cnt[r] = 0;
    if (USE_V) {
#pragma unroll
        for (int r = 0; r < NS; r++) {
            if ((R.x[r] >> 16) < a.v) R.a[r] = 0u;       // fails the value filter: the word nothing matches
            R.x[r] &= 0xFFFF;
        }
    }
    const int nl = IGD_LN_A(ln) + IGD_LN_B(ln);          // (a far unit never gets here: the lean build lists it)
    int g2 = 0, deadk = 0, nLater = 0;
    if (ln) {
        g2 = IGD_LN_G2(ln);
        deadk = a.rule == IGD_HIP_RULE_NEST ? (__builtin_amdgcn_readlane(L.jf, kk) & 14) : 0;
    }
    const int nE = nl + c0;
    uint32_t W[NS];
#pragma unroll
    for (int r = 0; r < NS; r++) W[r] = (uint32_t)__builtin_amdgcn_readlane(L.w[r], kk);
    int w = R.q;
    if (nl) {
        bool covers;
        const int lw = later_word(db.nbp, R.lw, g2, deadk, lane < nl, covers);
        nLater = __popcll(__ballot(covers));
        w = lane < nl ? lw : w;
    }
    auto match = [&](int P0) {
#pragma unroll
        for (int r = 0; r < NS; r++) match_slot_asm<false>(cnt[r], W[r], P0, R.a[r], 0ull);
    };
    if (nE > IGD_WAVE) {
        int wn = (IGD_WAVE + lane < nE) ? ~a.qw0[f0 + IGD_WAVE + lane - nl] : (int)IGD_NEVER;
        match(w);
        for (int p = IGD_WAVE; p < nE; p += IGD_WAVE) {
            w = wn;
            wn = (p + IGD_WAVE + lane < nE) ? ~a.qw0[f0 + p + IGD_WAVE + lane - nl] : (int)IGD_NEVER;
            match(w);
        }
    } else if (nE > 0) match(w);
    if (nLater != 0) {                                   // (:510-511: a later-tile query does not count the records that start before the tile)
#pragma unroll
        for (int r = 0; r < NS; r++) cnt[r] -= (R.a[r] & 0xFFFFu) == 0xFFFFu ? nLater : 0;
    }
#pragma unroll
    for (int r = 0; r < NS; r++) {
        if (!USE_V) {
            unsigned off;
            asm("v_mad_u32_u16 %0, %1, 4, 0" : "=v"(off) : "v"(R.x[r]));
            atomicAdd((unsigned int *)((char *)hits + off), (unsigned)cnt[r]);
        } else atomicAdd((unsigned int *)hits + R.x[r], (unsigned)cnt[r]);
    }
}

// the lean build's unit, by how many of its slots can hold a record
template <bool USE_V, bool CNT32, bool LDS_HITS, int FEW>
__device__ __forceinline__ void s_compute_lean(const DbView &db, const SortArgs &a, const SRegs &L, int kk, int lane, Raw2 &R,
                                               u64 *hits, unsigned short *sl, unsigned int *hist, unsigned short *sb, bool rankOK,
                                               unsigned *spent, unsigned budget)
{
#if IGD_LEAN_SMALL && IGD_ASM_MATCH && IGD_EXP == 0 && !IGD_EXP_NOMATCH
    if (CNT32 && LDS_HITS && FEW == 0) {
        const int un = R.n;
        if (un == 0) return;
        if (un <= IGD_WAVE) { s_compute_small<USE_V, 1>(db, a, L, kk, lane, R, hits); return; }
        if (un <= 3 * IGD_WAVE) { s_compute_small<USE_V, 3>(db, a, L, kk, lane, R, hits); return; }
    }
#endif
    s_compute<USE_V, CNT32, false, LDS_HITS, FEW>(db, a, L, kk, lane, R, hits, sl, hist, sb, rankOK, nullptr, spent, budget);
}

// CNT32: the workgroup's private counters are 32-bit (LDS_HITS only; the host checks that no counter of the launch can
// reach 2^32); BIG: more than 2^30 records (see s_issue).
// The full (rank method) build wants ~82 VGPRs and ~100 SGPRs: cut to the 64 / 80 of 8 waves per SIMD it spilled 13 + 34 of
// them; at 6 waves per SIMD (two workgroups of 768) nothing spills -- 1.25e7 queries: 234 -> 210 us, and the pairwise
// path of this build runs the headline batch in 72 instead of 94 us.  The lean build is as fast at either.
#ifndef IGD_WG_RANK
#define IGD_WG_RANK 768         // threads per workgroup / waves per SIMD of the full (rank method) build
#define IGD_WPE_RANK 6
#endif
#ifndef IGD_XCD_REMAP
#define IGD_XCD_REMAP 0         // 1: an XCD (blockIdx & 7) takes a contiguous eighth of every round of units
#endif
#ifndef IGD_LEAN_SKIP
#define IGD_LEAN_SKIP 1         // the lean build steps through the visited units only when a round has many others
#endif
#define IGD_ROUND 62            // units whose descriptors a wave of the lean build reads at once, one per lane; the last two lanes stay empty (s_issue, PLAIN)
#ifndef IGD_WG_LEAN
#define IGD_WG_LEAN IGD_WG      // ... and of the lean build
#define IGD_WPE_LEAN IGD_WPE
#endif
template <bool USE_V, bool LDS_HITS, bool CNT32, bool BIG, bool RANK, int FEW = 0>
// (waves per SIMD pinned from both sides: with only the lower bound the compiler budgets the scalar registers for 10 waves --
// 80 -- although the vector registers already hold the kernel at 8, and spills a dozen of them)
__global__ __launch_bounds__(RANK ? IGD_WG_RANK : IGD_WG_LEAN)
__attribute__((amdgpu_waves_per_eu(RANK ? IGD_WPE_RANK : IGD_WPE_LEAN, RANK ? IGD_WPE_RANK : IGD_WPE_LEAN))) void igd_scan_sorted(SortK K)
{
    constexpr int WGT = RANK ? IGD_WG_RANK : IGD_WG_LEAN;
    const DbView &db = K.db;
    const SortArgs &a = K.a;
    bool rankOK;
    {
        const int32_t *ctl = KARG(a.ctl);
        if (__builtin_amdgcn_readfirstlane(ctl[CTL_UNSORTED]) == a.epoch) return;   // not ordered: the bucket path's batch
        rankOK = __builtin_amdgcn_readfirstlane(ctl[CTL_NOTSTART]) != a.epoch;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    u64 *hits;
    unsigned short *sl;
    {
        const int nFiles = KARG(db.nFiles);
        const size_t hitBytes = LDS_HITS ? (((size_t)nFiles * (CNT32 ? 4 : 8) + 15) & ~(size_t)15) : 0;
        hits = LDS_HITS ? (u64 *)smem : KARG(a.out);
        sl = (unsigned short *)(smem + hitBytes + (size_t)wid * (size_t)KARG(a.wldsBytes));
        if (LDS_HITS) {
            if (CNT32) for (int f = threadIdx.x; f < nFiles; f += WGT) ((unsigned int *)hits)[f] = 0u;
            else for (int f = threadIdx.x; f < nFiles; f += WGT) hits[f] = 0;
        }
    }
    unsigned int *hist = (unsigned int *)(sl + IGD_WLDS_S);
    unsigned short *sb = (unsigned short *)(hist + IGD_WLDS_H);
    if (RANK) {
        for (int k = lane; k < IGD_WLDS_S; k += IGD_WAVE) sl[k] = 65535;
        for (int k = lane; k < IGD_WLDS_H; k += IGD_WAVE) hist[k] = 0u;
    }
    if (LDS_HITS) __syncthreads();
    const int wavesPerWG = WGT / IGD_WAVE;
#if IGD_XCD_REMAP
    const int lblk = (gridDim.x & 7) == 0 ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
#else
    const int lblk = (int)blockIdx.x;
#endif
    const int gwave = lblk * wavesPerWG + wid;
    const int nwaves = gridDim.x * wavesPerWG;
    Raw2 A, B;
    unsigned spent = 0u;                                 // CNT32: what this wave's units may have added to any one LDS counter
    const unsigned budget = 0xFFFFFFFFu / (unsigned)(WGT / IGD_WAVE);
    // Issue slots go to the OLDEST wave of a SIMD first: left alone, the eight waves of a SIMD finish their equal
    // shares one after the other (the first in 63 % of the last one's time, measured) and the SIMD runs ever emptier
    // towards the end.  Every wave therefore lowers its own priority as it gets through its share -- a wave that is
    // behind outranks one that is ahead -- and they finish together.
    const int myUnits = (db.nUnits - gwave + nwaves - 1) / nwaves;
    const int quarter = (myUnits + 3) >> 2;
    int prioAt = quarter, prioLevel = 3, done = 0;
#if IGD_OPT_PRIO
    __builtin_amdgcn_s_setprio(3);
#endif

    for (int ub = gwave; ub < db.nUnits; ub += nwaves * IGD_ROUND) {
        SRegs L;
        L.offLo = L.offHi = L.n = L.jf = L.f0 = L.c0 = L.la = L.ln = 0;
#pragma unroll
        for (int r = 0; r < IGD_SLOTS; r++) L.w[r] = 0;
        {
            const long long mi = (long long)ub + (long long)lane * nwaves;
            if (lane < IGD_ROUND && mi < db.nUnits) {
                const Unit *units = KARG(db.units);
                const int32_t *firstQ = KARG(a.firstQ), *spill = KARG(a.spill);
                const UnitRegs u = load_unit_regs(units + mi);
                L.offLo = u.offLo; L.offHi = u.offHi; L.jf = u.jf;
                L.n = u.n;
#pragma unroll
                for (int r = 0; r < IGD_SLOTS; r++) L.w[r] = u.w[r];
                if (u.n > 0) {
                    const int lj = u.jf >> 4;
                    const int lb = lj < IGD_SHORT_TILES - 1 ? lj : IGD_SHORT_TILES - 1;
                    L.f0 = firstQ[u.tile];
                    L.c0 = firstQ[u.tile + 1] - L.f0;
                    // some query reaches this tile as a later tile: where the entries of the queries [fl, f0) lie
                    if (spill[u.tile] == a.epoch) later_range<true>(a, u.tile, lb, firstQ[u.tile - lb], L.f0, L.la, L.ln);
                    // a tile with very many first-tile queries is shared out over all waves (heavy_sorted_body); its own
                    // waves keep the later-tile candidates.  Every unit of the tile takes the same decision from the same
                    // count, its first unit lists it; the list holds IGD_HEAVYS_MAX tiles -- more than a batch can have.
                    const bool heavy = L.c0 > (RANK ? IGD_HEAVY_FIRST : IGD_LEAN_FIRST);
                    if (heavy) {
                        if ((u.jf & 1) && !a.noList) KARG(a.heavyS)[atomicAdd(&KARG(a.ctlw)[CTL_NHEAVYS + (a.epoch & 1)], 1)] = u.tile;
                        L.c0 = 0;
                    }
                    // the lean build keeps nothing but 32-bit LDS counters: a far unit -- whose later-tile candidates nobody
                    // has counted -- is left, whole, to far_units_body in the batch's last launch (the full build bounds what
                    // every unit can add and sends the unit that would overflow a counter to the global hits[] itself)
                    // (... and so does the full build when the candidates span IGD_FAR_WIDE blocks of later[] or more -- a tile
                    // behind one with 10^4 .. 10^6 queries: there the unit is shared out over many waves)
                    if (IGD_LN_FAR(L.ln) && (!RANK || (!BIG && (L.f0 >> a.lbShift) - (L.la >> a.lbShift) >= IGD_FAR_WIDE))) {
                        if (!a.noList) KARG(a.farList)[atomicAdd(&KARG(a.ctlw)[CTL_NFAR + (a.epoch & 1)], 1)] = (int)mi | (heavy ? (int)0x80000000 : 0);
                        L.c0 = 0; L.ln = 0;
                    }
                    if ((L.c0 | L.ln) == 0) L.n = 0;     // nobody asks about this unit: no loads (s_issue), nothing to compare
                    if (!BIG) {                          // (offHi is free: < 2^30 records)
                        const int nlL = IGD_LN_FAR(L.ln) ? 0 : IGD_LN_A(L.ln) + IGD_LN_B(L.ln);
                        const int b0L = L.c0 < IGD_WAVE - nlL ? L.c0 : IGD_WAVE - nlL;
                        L.offHi = (L.f0 + b0L) * 4;
                    }
                }
            }
        }
        int cntU = (int)(((long long)db.nUnits - ub + nwaves - 1) / nwaves);
        if (cntU > IGD_ROUND) cntU = IGD_ROUND;
        // (the lean build: when more than a quarter of the round's units hold no record or are asked about by nobody -- a database
        // with empty tiles -- it steps through the others only, like the full build; its own loop below takes every unit in turn
        // with nothing to find out per unit, which is what a round of visited units wants)
#define IGD_UNIT(kk_, R_) do { if (RANK) s_compute<USE_V, CNT32, RANK, LDS_HITS, FEW>(db, a, L, kk_, lane, R_, hits, sl, hist, sb, rankOK, nullptr, &spent, budget); \
                               else s_compute_lean<USE_V, CNT32, LDS_HITS, FEW>(db, a, L, kk_, lane, R_, hits, sl, hist, sb, rankOK, &spent, budget); } while (0)
        const unsigned long long mVis = __ballot((L.c0 | L.ln) != 0 && L.n > 0);
        if (RANK || (IGD_LEAN_SKIP && __popcll(mVis) * 4 < cntU * 3)) {
            // The full build also serves batches that visit a fraction of the units (one GPU's slab of config 4: one unit
            // in eight): the wave steps through the units somebody asks about only -- an unvisited one still cost its
            // dozen zero-size loads, which queue up behind everybody's real ones.
            unsigned long long m = mVis;
            const int visited = __popcll(m);
            int qd = (visited + 3) >> 2, at = qd, level = 3, nd = 0;
#if IGD_OPT_PRIO
            __builtin_amdgcn_s_setprio(3);
#endif
            int ka = -1, kb = -1;
            if (m) { ka = __builtin_ctzll(m); m &= m - 1; }
            if (m) { kb = __builtin_ctzll(m); m &= m - 1; }
            s_issue<USE_V, BIG>(db, a, L, ka < 0 ? 0 : ka, ka >= 0, lane, A);
            while (ka >= 0) {
                s_issue<USE_V, BIG>(db, a, L, kb < 0 ? 0 : kb, kb >= 0, lane, B);
                s_compute<USE_V, CNT32, RANK, LDS_HITS, FEW>(db, a, L, ka, lane, A, hits, sl, hist, sb, rankOK, nullptr, &spent, budget);
                ka = -1;
                if (m) { ka = __builtin_ctzll(m); m &= m - 1; }
                s_issue<USE_V, BIG>(db, a, L, ka < 0 ? 0 : ka, ka >= 0, lane, A);
                if (kb >= 0) s_compute<USE_V, CNT32, RANK, LDS_HITS, FEW>(db, a, L, kb, lane, B, hits, sl, hist, sb, rankOK, nullptr, &spent, budget);
                kb = -1;
                if (m) { kb = __builtin_ctzll(m); m &= m - 1; }
#if IGD_OPT_PRIO
                nd += 2;
                if (nd >= at) {
                    at += qd;
                    level--;
                    if (level == 2) __builtin_amdgcn_s_setprio(2);
                    else if (level == 1) __builtin_amdgcn_s_setprio(1);
                    else __builtin_amdgcn_s_setprio(0);
                }
#endif
            }
            continue;
        }
        // (three units in flight per wave -- the next unit's loads issued before the current one is compared -- need 74 registers,
        // i.e. 6 waves per SIMD: measured twice, 68.3 against 64.9 us on the final build; the code is gone)
        s_issue<USE_V, BIG, !RANK>(db, a, L, 0, true, lane, A);
        for (int kk = 0; kk < cntU; kk += 2) {
            s_issue<USE_V, BIG, !RANK>(db, a, L, kk + 1, kk + 1 < cntU, lane, B);
            IGD_UNIT(kk, A);
            s_issue<USE_V, BIG, !RANK>(db, a, L, kk + 2, kk + 2 < cntU, lane, A);
            if (kk + 1 < cntU) IGD_UNIT(kk + 1, B);
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
    }
    if (LDS_HITS) {
        __syncthreads();
        const int nFiles = KARG(db.nFiles);
        u64 *row = KARG(a.out) + (size_t)blockIdx.x * nFiles;
        if (CNT32) {                                     // 32-bit rows: half the bytes written here and read back by k_reduce_slabs
            unsigned int *row32 = (unsigned int *)KARG(a.out) + (size_t)blockIdx.x * nFiles;
            for (int f = threadIdx.x; f < nFiles; f += WGT) row32[f] = ((unsigned int *)hits)[f];
        } else for (int f = threadIdx.x; f < nFiles; f += WGT) row[f] = hits[f];
    }
}

#undef IGD_UNIT

// heavy_sorted_body: the merge join's skew valve.  A tile with more than IGD_HEAVY_FIRST (lean build: IGD_LEAN_FIRST)
// first-tile queries -- 10^6 ordered queries inside ONE tile would keep one wave busy for 9 ms -- is listed by
// igd_scan_sorted and left out there; here every (unit of the tile, slice of IGD_HEAVY_SLICE queries) is one work item,
// dealt round-robin to all waves of the hosting launch (the batch's last kernel) -- the rank method is a sum over
// queries, so slices simply add up -- and added to hits[] and the batch total with global atomics.  `wsm`: this wave's
// LDS area for the rank method.  Must sit in a kernel whose FIRST argument is the batch's SortK (KARG).
template <bool USE_V, bool BIG>
__device__ __forceinline__ void heavy_sorted_body(const SortK &K, u64 *__restrict__ d_hits, u64 *__restrict__ d_total,
                                                  unsigned char *wsm, int gwave, int nwaves, int lane, int ctlv, igd_lds_u64 *lhits = nullptr)
{
    const DbView &db = K.db;
    const SortArgs &a = K.a;
    if (__builtin_amdgcn_readlane(ctlv, CTL_UNSORTED) == a.epoch) return;
    int nH = __builtin_amdgcn_readlane(ctlv, CTL_NHEAVYS + (a.epoch & 1));
    if (nH == 0) return;
    if (nH > IGD_HEAVYS_MAX) nH = IGD_HEAVYS_MAX;        // (cannot happen: see IGD_HEAVYS_MAX)
    unsigned short *sl = (unsigned short *)wsm;
    unsigned int *hist = (unsigned int *)(sl + IGD_WLDS_S);
    unsigned short *sb = (unsigned short *)(hist + IGD_WLDS_H);
    for (int k = lane; k < IGD_WLDS_S; k += IGD_WAVE) sl[k] = 65535;
    for (int k = lane; k < IGD_WLDS_H; k += IGD_WAVE) hist[k] = 0u;
    const bool rankOK = __builtin_amdgcn_readlane(ctlv, CTL_NOTSTART) != a.epoch;
    // (the listed tiles' ranges are looked up 64 at a time, one tile per lane: the lean build may list thousands)
    int lf0 = 0, lc0 = 0, lu0 = 0, lnu = 0;
    deal_items(nH, gwave, nwaves, lane,
        [&](int h) {
            lf0 = lc0 = lu0 = lnu = 0;
            if (h < 0) return 0;
            const int tl = a.heavyS[h];
            lf0 = a.firstQ[tl]; lc0 = a.firstQ[tl + 1] - lf0;
            lu0 = db.tileUnit0[tl]; lnu = db.tileUnit0[tl + 1] - lu0;
            return lnu * ((lc0 + IGD_HEAVY_SLICE - 1) / IGD_HEAVY_SLICE);
        },
        [&](int hh, int it) {
            const int f0 = __builtin_amdgcn_readlane(lf0, hh), c0 = __builtin_amdgcn_readlane(lc0, hh);
            const int u0 = __builtin_amdgcn_readlane(lu0, hh), nu = __builtin_amdgcn_readlane(lnu, hh);
            const int u = u0 + it % nu, sc = it / nu;
            const UnitRegs ur = load_unit_regs(db.units + u);                     // the same unit in every lane
            SRegs L;
            L.offLo = ur.offLo; L.offHi = ur.offHi; L.n = ur.n; L.jf = ur.jf;
#pragma unroll
            for (int r = 0; r < IGD_SLOTS; r++) L.w[r] = ur.w[r];
            L.f0 = f0 + sc * IGD_HEAVY_SLICE;
            L.c0 = c0 - sc * IGD_HEAVY_SLICE < IGD_HEAVY_SLICE ? c0 - sc * IGD_HEAVY_SLICE : IGD_HEAVY_SLICE;
            L.la = 0; L.ln = 0;
            Raw2 A;
            s_issue<USE_V, BIG>(db, a, L, 0, true, lane, A);
            s_compute<USE_V, false, true, true>(db, a, L, 0, lane, A, d_hits, sl, hist, sb, rankOK, d_total, nullptr, 0u, lhits);
        });
}

// far_units_body: the units the lean build of igd_scan_sorted listed (far: more later-tile candidates than come with a
// unit's records -- tiles behind very dense ones; that build has neither the walk over the blocks nor 64-bit counters in
// its registers).  One wave per listed unit: its later-tile candidates (far_later) and, unless its tile went to
// heavy_sorted_body, its first-tile queries, added to hits[] and the batch total with global atomics.
template <bool USE_V, bool BIG>
__device__ __forceinline__ void far_units_body(const SortK &K, u64 *__restrict__ d_hits, u64 *__restrict__ d_total,
                                               unsigned char *wsm, int gwave, int nwaves, int lane, int ctlv, igd_lds_u64 *lhits = nullptr)
{
    const DbView &db = K.db;
    const SortArgs &a = K.a;
    if (__builtin_amdgcn_readlane(ctlv, CTL_UNSORTED) == a.epoch) return;
    const int nF = __builtin_amdgcn_readlane(ctlv, CTL_NFAR + (a.epoch & 1));
    if (nF == 0) return;
    unsigned short *sl = (unsigned short *)wsm;
    unsigned int *hist = (unsigned int *)(sl + IGD_WLDS_S);
    unsigned short *sb = (unsigned short *)(hist + IGD_WLDS_H);
    for (int k = lane; k < IGD_WLDS_S; k += IGD_WAVE) sl[k] = 65535;
    for (int k = lane; k < IGD_WLDS_H; k += IGD_WAVE) hist[k] = 0u;
    const bool rankOK = __builtin_amdgcn_readlane(ctlv, CTL_NOTSTART) != a.epoch;
    // A listed unit behind a very dense tile has the later-tile words of up to 10^6 queries to go through -- hundreds of
    // blocks of later[], 2.8 ms for one wave.  Every unit is therefore shared out in nS slices of its blocks (the rank
    // method is a sum over queries: slices add up); slice 0 also takes the unit's first-tile queries.  nS shrinks as the
    // list grows (a slice beyond a unit's few blocks costs its wave three dependent loads to find that out).
    if (nF > db.nUnits) return;                          // (cannot happen: a unit is listed once)
    int nS = nwaves / nF;
    nS = nS < 1 ? 1 : (nS > IGD_FAR_SLICES ? IGD_FAR_SLICES : nS);
    const int sh = a.lbShift;
    for (long long item = gwave; item < (long long)nF * nS; item += nwaves) {
        const int i = (int)(item / nS), sl_ = (int)(item % nS);
        const int ent = __builtin_amdgcn_readfirstlane(a.farList[i]);
        const UnitRegs ur = load_unit_regs(db.units + (ent & 0x7fffffff));        // the same unit in every lane
        SRegs L;
        L.offLo = ur.offLo; L.offHi = ur.offHi; L.n = ur.n; L.jf = ur.jf;
#pragma unroll
        for (int r = 0; r < IGD_SLOTS; r++) L.w[r] = ur.w[r];
        const int lj = ur.jf >> 4;
        const int lb = lj < IGD_SHORT_TILES - 1 ? lj : IGD_SHORT_TILES - 1;
        L.f0 = a.firstQ[ur.tile];
        L.c0 = (ent < 0 || sl_ > 0) ? 0 : a.firstQ[ur.tile + 1] - L.f0;
        later_range<false>(a, ur.tile, lb, a.firstQ[ur.tile - lb], L.f0, L.la, L.ln);
        if (nS > 1) {
            if (sl_ == 0) L.ln = 0;                      // slice 0: the first-tile queries (it needs the true f0) ...
            else if (L.ln != 0) {
                // ... slices 1 .. nS-1: blocks b0 .. b1 of the unit's bA .. bB, described the way far_later reads a range -- nA
                // entries from la on, whole blocks between, the first nB entries of the block that holds "query f0"
                const int bA = L.la >> sh, bB = L.f0 >> sh;
                const int per = (bB - bA + nS - 1) / (nS - 1);   // = ceil((bB - bA + 1) / (nS - 1))
                const int b0 = bA + (sl_ - 1) * per;
                int b1 = b0 + per - 1;
                b1 = b1 > bB ? bB : b1;
                if (b0 > bB) continue;
                if (b0 != bA || b1 != bB) {
                    int nA = IGD_LN_A(L.ln), nB = IGD_LN_B(L.ln);
                    if (b0 != bA) { L.la = b0 << sh; nA = b0 == bB ? nB : a.laterHdr[b0].x; }
                    if (b1 != bB) { L.f0 = b1 << sh; nB = b1 == b0 ? 0 : a.laterHdr[b1].x; }   // (c0 = 0 here: f0 only names the last block)
                    L.ln = (L.ln & ~0x3FFFFFF) | nA | (nB << 13);
                }
            }
        }
        if ((L.c0 | L.ln) == 0) continue;
        Raw2 A;
        s_issue<USE_V, BIG>(db, a, L, 0, true, lane, A);
        s_compute<USE_V, false, true, true>(db, a, L, 0, lane, A, d_hits, sl, hist, sb, rankOK, d_total, nullptr, 0u, lhits);
    }
}
