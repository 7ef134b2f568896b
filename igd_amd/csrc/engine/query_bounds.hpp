// engine/query_bounds.hpp -- part of igd_hip.hip (included there once; not a stand-alone header).
// tile arithmetic, k_query_bounds (queries grouped by tile, fix list, coverage differences), k_count_pairs
// ------------------------------------------------------------------------------------------
// coordinate -> tile index with C semantics (truncation toward zero), src/igd_search.c:459
// x / 2^sh the way C divides (towards zero)
__device__ __forceinline__ int tile_shift(int x, int sh)
{
    return (int)((unsigned)(x + ((x >> 31) & ((1 << sh) - 1)))) >> sh;
}
__device__ __forceinline__ int tile_of(const DbView &db, int x)
{
    if (db.shift >= 0) {
        unsigned m = x < 0 ? 0u - (unsigned)x : (unsigned)x;
        int t = (int)(m >> db.shift);
        return x < 0 ? -t : t;
    }
    return x / db.nbp;
}

// Re-tiled copy (DbView::vshift >= 0): does the FILE's tiling let this query count anything?  nest: rule NEST of the call.
__device__ __forceinline__ bool real_gate(const DbView &db, int c, int qs, bool nest)
{
    const int n1 = tile_shift(qs, db.vshift);            // C division by the file's tile width (:459)
    if (n1 < 0 || n1 >= db.rNTile[c]) return false;      // (:462; n1 < 0: out of bounds in the reference)
    if (!nest) return true;
    const unsigned g = (unsigned)(db.rBase[c] + n1);
    return ((db.rEmpty[g >> 5] >> (g & 31)) & 1u) == 0u; // (:468)
}

// Tile span of one query = the prologue of every reference kernel (src/igd_search.c:455-467):
// n1=qs/nbp, n2=(qe-1)/nbp (C division), n1>mTile -> nothing, n2 clamped, and for rule NEST an
// empty first tile ends the query (:468).  Returns false when the query visits nothing.
// (the contig's two table entries -- its tiles, the number of its first one -- come from the caller: k_split_local keeps the tables in LDS)
__device__ __forceinline__ bool query_span_at(const DbView &db, int c, int ctgTiles, int ctgFirst, int qs, int qe, int rule,
                                              int &gt0, int &ntl)
{
    if (db.vshift >= 0) {                     // a re-tiled copy: the file's tiles decide whether the query counts at all ...
        if (!real_gate(db, c, qs, (rule >> 8) & 1)) return false;
        if (qs < 0) qs = 0;                   // ... and a start before the contig (above -nbp of the file) lies in tile 0
        rule &= 0xff;                         // (the copy's own tiles are visited under rule FLAT)
    }
    int n1 = tile_of(db, qs);
    int n2 = tile_of(db, (int)((unsigned)qe - 1u));
    int mT = ctgTiles - 1;
    if (n1 < 0 || n1 > mT) return false;      // n1<0: out-of-bounds read in the reference
    if (n2 > mT) n2 = mT;
    gt0 = ctgFirst + n1;
    if (rule == IGD_HIP_RULE_NEST && db.tileCnt[gt0] == 0) return false;
    ntl = n2 > n1 ? n2 - n1 + 1 : 1;
    return true;
}
__device__ __forceinline__ bool query_span(const DbView &db, int c, int qs, int qe, int rule,
                                           int &gt0, int &ntl)
{
    if (c < 0 || c >= db.nCtg) return false;
    return query_span_at(db, c, db.ctgNTile[c], db.ctgBase[c], qs, qe, rule, gt0, ntl);
}

// Control words shared by the kernels of one batch (int32 ctl[16]):
//   ctl[1] = epoch of the last batch whose queries were NOT ordered by tile
//   ctl[2] = epoch of a batch that broke a caller's IGD_HIP_FLAG_SORTED promise since the last igd_hip_sync
//            (written by k_query_bounds, cleared by igd_hip_sync: no broken batch goes unreported)
//   ctl[4 + (epoch & 1)] = entries of the bucket path's exact-walk list (k_count_pairs)
//   ctl[6 + (epoch & 1)] = entries of the merge-join path's exact-walk list (k_query_bounds)
//   ctl[8 + (epoch & 1)] = gap-fill budget spent by k_query_bounds (units of 256 tiles)
#define CTL_UNSORTED 1
#define CTL_BROKEN 2
#define CTL_NOTSTART 3   // epoch of the last batch whose queries were ordered by tile but NOT by start inside a tile
#define CTL_NHEAVY 10    // + (epoch & 1): tiles of the batch listed for heavy_bucket_body (bucket path)
#define IGD_HEAVY_PAIRS 2048   // a tile with more (query, tile) pairs than this is shared out in slices of that many
#define IGD_HEAVY_MAX 4096     // listed heavy tiles per batch (a further one stays with its own wave)
#define CTL_NHEAVYS 12   // + (epoch & 1): tiles listed for heavy_sorted_body (merge join)
#define CTL_NFAR 14      // + (epoch & 1): units the lean build of igd_scan_sorted leaves to far_units_body
#define IGD_HEAVY_FIRST 8192   // merge join: a tile with more first-tile queries than this is shared out in slices of IGD_HEAVY_SLICE
// One slice = one wave's work in the batch's last launch: its batches of 64 queries one after the other, each a memory round
// trip with nothing else on the CU to hide it.  Slices of 4096 (until round 5) left 10^6 queries inside ONE tile to 244 waves
// of 64 batches each -- 130 us on an otherwise idle chip; slices of 512 are 1953 waves of 8 batches (round 6).
#ifndef IGD_HEAVY_SLICE
#define IGD_HEAVY_SLICE 512
#endif
#ifndef IGD_FAR_SLICES
#define IGD_FAR_SLICES 1024    // far_units_body: at most this many slices per listed unit
#endif
#define IGD_FAR_WIDE 8         // full build: a unit whose later-tile candidates span this many blocks of later[] goes to far_units_body
#define IGD_LEAN_FIRST 512     // the lean (pairwise-only) build of igd_scan_sorted hands denser tiles to heavy_sorted_body
// The merge join's list can never overflow: a batch has <= IGD_MAX_BATCH queries and a listed tile holds more than
// IGD_LEAN_FIRST (full build: IGD_HEAVY_FIRST) of them as first-tile queries, each query in exactly one tile.
#define IGD_HEAVYS_MAX ((int)(IGD_MAX_BATCH / IGD_LEAN_FIRST))
static_assert(IGD_LEAN_FIRST <= IGD_HEAVY_FIRST, "the list of heavy_sorted_body is sized for the lean build's threshold");
#define CTL_NLONG 4
#define CTL_NFIX 6
#define CTL_BUDGET 8
// Exact-walk list entries (int2: query index, kind).  The scan kernels handle the common case
// only; what they leave out is listed by the grouping kernels and done by k_exact_walk:
#define WALK_FIRST 1    // tile n1 only: first-tile query with qe <= tile start (compact image cannot express it)
#define WALK_ALL 2      // bucket path: the first and the last tile of a long query (the tiles between: coverage, see IGD_COV_*)
#define WALK_LAST 3     // merge join: tile n2 only of a long query (n1 .. n1+3 by the scan kernel, the tiles between: coverage)
// Long queries (more than IGD_SHORT_TILES tiles) in the merge join.  Tiles n1+1 .. n1+3 are reached by the query's
// later[] entry like any other query's, the LAST tile n2 is walked exactly (WALK_LAST) -- and the tiles between, which the
// query covers from end to end, by COVERAGE: every record that STARTS in such a tile (and passes the value filter) is
// an overlap, whatever the query's ends are, so all that matters per tile is HOW MANY long queries cover it.
// k_query_bounds adds +1 / -1 at the ends of each query's covered range to a difference array (two atomics per long
// query, however long); the batch's last launch turns it into counts by a running sum and adds count x (records
// starting in the tile) to hits[] (coverage_body): one atomic per record instead of one per (query, record) -- the
// walk that did this before took 123 ms for 10^6 queries of 100-200 kb.  Two sets of arrays (batch parity): the last
// launch of batch k+1 zeroes what batch k used.
#define IGD_COV_SHIFT 10          // coarse level of the difference array: sums over 1024 tiles
#define CTL_COV 16         // + set * 2 + parity (set 0: merge join, 1: bucket path): the epoch whose long queries wrote the set
#define IGD_COV_LEN(nT_) ((size_t)(nT_) + 2 + ((size_t)(nT_) >> IGD_COV_SHIFT) + 2)   // one set: diff[nT + 2], coarse[(nT >> IGD_COV_SHIFT) + 2]                // + (epoch & 1): epoch of the batch that put something into the parity's difference arrays
#define CTL_PILED 20       // epoch of the last batch in which k_split_local saw a piled-up coarse bucket (k_split_fine_b has work)
#define IGD_CTL_WORDS 32

// Sorted path, step 1.  key(i) = global tile id of query i's FIRST tile, clamped into the
// tile range of its contig (unknown contigs go to the ends), so a batch ordered by
// (contig, start) has non-decreasing keys.  firstQ[t] = first i with key(i) >= t, for
// t = 0..nT (firstQ[nT] = nq).  A decreasing key marks the batch unsorted (ctl[1] = epoch).
__device__ __forceinline__ int tile_key(const DbView &db, int c, int qs)
{
    if (c < 0) return 0;
    if (c >= db.nCtg) return db.nT - 1;
    int n1 = tile_of(db, qs);
    int mT = db.ctgNTile[c] - 1;
    n1 = n1 < 0 ? 0 : (n1 > mT ? mT : n1);
    return db.ctgBase[c] + n1;
}

// inclusive prefix sum over the 64 lanes (DPP: row_shr 1,2,4,8, then row_bcast 15 and 31)
__device__ __forceinline__ int wave_inclusive_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return v;
}

// The compact query word of k_pack_units' image (defined here because k_query_bounds writes it):
//     (65536 - qe') | qs' << 16   with   qe' = min(qe - T, W) + 1,   qs' = first ? max(qs - T + 1, 1) : 1
// A record word matches when both of its 16-bit halves are >= the query's.
__device__ __forceinline__ int query_word(int qs, int qe, bool first, int T, int W)
{
    int qe2 = qe - T;
    qe2 = (qe2 < W ? qe2 : W) + 1;                       // s' < qe2   <=> start < qe
    int qs2 = first ? qs - T + 1 : 1;                    // e' >= qs2  <=> end > qs   (e' >= 1 always)
    if (qs2 < 1) qs2 = 1;
    return (int)((unsigned)(65536 - qe2) | ((unsigned)qs2 << 16));
}
#define IGD_NEVER 0xFFFFFFFFu     // a query word no record word can match (e' <= W <= 32768 < 65535)
#define QB_CTG 1024               // contigs whose tile tables k_query_bounds keeps in LDS
#define QB_COVW 128               // tiles in a wave's LDS window of coverage differences

// What k_query_bounds leaves per query for the merge join (sorted path):
//   compact image (packed != 0):
//     qw0[i] = ~(query word for the query's FIRST tile), or ~IGD_NEVER = 0 when the query does not take part
//              there (unknown contig, first tile out of range, rule NEST with an empty first tile, or the one case
//              the image cannot express, listed as WALK_FIRST).  Stored inverted so that a bounds-checked
//              buffer load past a tile's last query (which returns 0) reads as "never matches".
//     later[]: the queries that also cover LATER tiles (6 % of the benchmark's), compacted per "later block" (the
//              WGT * VEC consecutive queries of one workgroup; region [B << lbShift, ...) of the array), in query
//              order, one word each:  min(qe - T0, 4W) [bits 0..17] | min(span, 3) << 18 | (first global tile & 3) << 20
//              -- all a later tile needs (there, qs' = 1 and qe' = min(qeRel - k W, W) + 1; k = (tile - first tile)
//              follows from 2 bits).  An entry is never 0.
//     lpos[t]  = the number of entries of ITS block that come from queries before firstQ[t]: with firstQ[] itself that
//              makes the later-tile candidates of a tile -- the entries of the queries [firstQ[t - 3], firstQ[t]) -- one
//              run of words (two when the range crosses a block boundary) the scan reads without any search.
//     laterHdr[B] = (entries of block B, last tile any of them covers).
//     spill[t] = epoch for every tile t that some query covers as a later tile (k = 1..3): most units have
//              none and never look at the queries of the tiles before theirs.
//   exact arrays (packed == 0):
//     qw0[i] = (global number of the first tile) << 4 | min(n2 - n1, 15), -1 when it visits nothing.
// VEC queries per thread (4: the three query arrays are read, and the word arrays written, as dwordx4 -- a quarter of
// the memory instructions and four independent chains per thread; 1: arrays that are not 16-byte aligned).
// WGT threads per workgroup = WGT * VEC queries per later block: 1024 x 4 for the large batches, whose tiles have so many
// queries that the candidate range of a tile (the queries of three tiles) would span several smaller blocks.
// FAST: the usual case, decided by the host -- compact image, power-of-two tile size, contig tables that fit the LDS arrays --
// compiled without the other cases' branches (a flat load picking between LDS and global tables, a division), and with a
// short path for the waves all of whose queries lie in ONE contig, in range and in order (every wave of a large sorted
// batch but a few): keys and words from the wave's scalar contig base, nothing looked up or clamped per query.
// One long query covers tiles ta .. tb-1 (global tile numbers) from end to end: +1 / -1 in the batch's difference arrays
// (fine, and per block of 2^IGD_COV_SHIFT tiles), to be summed up by coverage_body in the batch's last launch.
__device__ __forceinline__ void cover_tiles(const DbView &db, int32_t *__restrict__ ctl, int set, int epoch, int ta, int tb)
{
    int32_t *diff = db.cov + (size_t)(set * 2 + (epoch & 1)) * IGD_COV_LEN(db.nT), *coarse = diff + db.nT + 2;
    atomicAdd(&diff[ta], 1); atomicAdd(&diff[tb], -1);
    if ((ta >> IGD_COV_SHIFT) != (tb >> IGD_COV_SHIFT)) { atomicAdd(&coarse[ta >> IGD_COV_SHIFT], 1); atomicAdd(&coarse[tb >> IGD_COV_SHIFT], -1); }
    ctl[CTL_COV + set * 2 + (epoch & 1)] = epoch;
}

// RUNS: a position-sorted batch given as contig RUNS -- `ichr` then points at runStart[nCtg + 1] (queries [runStart[c],
// runStart[c + 1]) lie on contig c, runStart[0] = 0, runStart[nCtg] = nq) instead of one contig number per query: 4 of the
// 12 bytes per query are not read (igd_hip_search_runs_dev).  The table is staged in LDS and checked (a table that is not
// monotone or does not cover [0, nq) is a broken order promise); a wave finds its contig with one bisection and is on the
// short path unless it straddles a run boundary.
// BONLY: the bounds alone -- firstQ[] and the order check -- for the DIRECT step (scan_direct.hpp), whose scan kernel reads the
// queries itself: no ends are read, no word is computed or stored, nothing is listed.  Its keys keep the queries of contig
// numbers outside the database OUT of every tile's range (-1: before tile 0, nT: behind the last tile), where the ordinary
// step clamps them into the first / last tile and masks their words.
template <int VEC, bool FAST, int WGT, bool RUNS = false, bool BONLY = false>
__global__ __launch_bounds__(WGT, WGT == 256 ? 7 : (FAST ? 8 : 4)) void k_query_bounds(DbView db, const int32_t *__restrict__ ichr,
                                                      const int32_t *__restrict__ qs,
                                                      const int32_t *__restrict__ qe, int nq, int rule,
                                                      int packed_, int32_t *__restrict__ firstQ, int32_t *__restrict__ lpos,
                                                      int2 *__restrict__ fix, int32_t *__restrict__ ctl, int epoch,
                                                      u64 *__restrict__ zeroHits, u64 *__restrict__ zeroTotal,
                                                      int32_t *__restrict__ qw0, int32_t *__restrict__ later,
                                                      int32_t *__restrict__ spill, int2 *__restrict__ laterHdr, int promised)
{
    constexpr int NW = WGT / IGD_WAVE;
    const bool vnest = (rule >> 8) & 1;                   // a re-tiled copy (DbView::vshift): rule NEST of the call, applied to the FILE's tiles
    rule &= 0xff;
    // The thread's queries (and the one before them) first: their loads are in flight while the tables below are staged
    // and the batch's state is looked up (a workgroup that then leaves at once has read 12 KiB for nothing).
    const int i0 = (int)(blockIdx.x * WGT + threadIdx.x) * VEC;
    int qc[VEC], qs_[VEC], qe_[VEC];
    if (VEC == 4) {
        int4 c4 = make_int4(0, 0, 0, 0), s4 = c4, e4 = c4;
        if (i0 + 3 < nq) {
            if (!RUNS) c4 = *(const int4 *)(ichr + i0);
            s4 = *(const int4 *)(qs + i0);
            if (!BONLY) e4 = *(const int4 *)(qe + i0);
        } else {
            if (i0 < nq) { if (!RUNS) c4.x = ichr[i0]; s4.x = qs[i0]; if (!BONLY) e4.x = qe[i0]; }
            if (i0 + 1 < nq) { if (!RUNS) c4.y = ichr[i0 + 1]; s4.y = qs[i0 + 1]; if (!BONLY) e4.y = qe[i0 + 1]; }
            if (i0 + 2 < nq) { if (!RUNS) c4.z = ichr[i0 + 2]; s4.z = qs[i0 + 2]; if (!BONLY) e4.z = qe[i0 + 2]; }
        }
        qc[0] = c4.x; qc[1 % VEC] = c4.y; qc[2 % VEC] = c4.z; qc[3 % VEC] = c4.w;
        qs_[0] = s4.x; qs_[1 % VEC] = s4.y; qs_[2 % VEC] = s4.z; qs_[3 % VEC] = s4.w;
        qe_[0] = e4.x; qe_[1 % VEC] = e4.y; qe_[2 % VEC] = e4.z; qe_[3 % VEC] = e4.w;
    } else if (i0 < nq) { qc[0] = RUNS ? 0 : ichr[i0]; qs_[0] = qs[i0]; qe_[0] = BONLY ? 0 : qe[i0]; }
    int pc = -1, ps = INT_MIN;
    if (i0 > 0 && i0 < nq) { if (!RUNS) pc = ichr[i0 - 1]; ps = qs[i0 - 1]; }
    // the batch's first and last query (head and tail of firstQ[], at the end of the kernel): asked for HERE -- four scalar
    // loads -- so that the kernel's last step is not two more dependent round trips in every wave
    int edgeC0 = 0, edgeS0 = 0, edgeC1 = 0, edgeS1 = 0;
    if (nq > 0) { if (!RUNS) { edgeC0 = ichr[0]; edgeC1 = ichr[nq - 1]; } edgeS0 = qs[0]; edgeS1 = qs[nq - 1]; }
    // the two per-contig tables every query looks up: from LDS (one latency instead of a dependent global gather)
    __shared__ int32_t sBase[QB_CTG], sNTile[QB_CTG];
    __shared__ int sCnt[NW], sFixCnt[NW], sFixBase, sFixAny;
    const bool ldsTab = FAST || db.nCtg <= QB_CTG;
    const int packed = FAST ? 1 : packed_;
    // Has any wave found the batch unordered already?  ONE device-scope load per workgroup (an L1-cached one would keep
    // returning the stale line): a load per wave -- 10^5 requests for the one address at 1.25e7 queries -- queued up at
    // its memory channel for as long as the rest of the kernel takes.
    __shared__ int sSeen;
    int seen = 0;
    if (threadIdx.x == 0) seen = __hip_atomic_load(&ctl[CTL_UNSORTED], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (looked at further down)
    if (threadIdx.x < NW) { sCnt[threadIdx.x] = 0; sFixCnt[threadIdx.x] = 0; }   // (a wave that leaves early counts as one without entries)
    if (threadIdx.x == NW) { sSeen = 0; sFixAny = 0; }                     // (defined also when wave 0 is the one that leaves before it stores the flag)
    if (ldsTab)
        for (int c = threadIdx.x; c < db.nCtg; c += WGT) { sBase[c] = db.ctgBase[c]; sNTile[c] = db.ctgNTile[c]; }
    __shared__ int32_t sCovW[NW * QB_COVW];               // per wave: coverage differences of its long queries (see step 3; cleared by the wave that uses it)
    __shared__ int32_t sRun[RUNS ? 2 * QB_CTG : 1];       // runStart[0..nCtg], padded with INT_MAX to a power of two
    __shared__ int sK[2];                                 // RUNS: contig of the batch's first and last query
    // What every batch owes the NEXT one and its caller -- hits[] cleared (IGD_HIP_FLAG_ZERO_FIRST), the other parity's list
    // counters at zero -- is done before any way out of the kernel (a malformed run table below leaves at once: the batch
    // after it would have appended to the stale fix list of two batches ago).
    const int t = blockIdx.x * WGT + threadIdx.x;
    if (zeroHits) for (int f = t; f < db.nFiles; f += gridDim.x * WGT) zeroHits[f] = 0;   // IGD_HIP_FLAG_ZERO_FIRST
    if (zeroTotal && t == 0) *zeroTotal = 0;
    if (t == 0) {                                           // next batch's list counters
        ctl[CTL_NLONG + ((epoch + 1) & 1)] = 0;
        ctl[CTL_NFIX + ((epoch + 1) & 1)] = 0;
        ctl[CTL_BUDGET + ((epoch + 1) & 1)] = 0;
        ctl[CTL_NHEAVY + ((epoch + 1) & 1)] = 0;
        ctl[CTL_NHEAVYS + ((epoch + 1) & 1)] = 0;
        ctl[CTL_NFAR + ((epoch + 1) & 1)] = 0;
    }
    int runLevels = 0;
    if (RUNS) {
        int badRuns = 0;
        while ((1 << runLevels) < db.nCtg + 1) runLevels++;
        for (int c = threadIdx.x; c < (1 << runLevels); c += WGT) {
            const int r0 = c <= db.nCtg ? ichr[c] : INT_MAX;
            sRun[c] = r0;
            if (c < db.nCtg) badRuns |= r0 > ichr[c + 1] ? 1 : 0;
            if (c == 0) badRuns |= (r0 != 0 || ichr[db.nCtg] != nq) ? 1 : 0;
        }
        if (__syncthreads_or(badRuns)) {                  // not a run table of this batch: the order promise is broken
            if (threadIdx.x == 0) { ctl[CTL_UNSORTED] = epoch; if (promised) ctl[CTL_BROKEN] = epoch; }
            return;
        }
    } else
    __syncthreads();
    // contig of query i: the number of run starts 1..nCtg that are <= i (an empty run shares its start with the next one)
    auto contig_of = [&](int i) -> int {
        int pos = 0;                                      // entries sRun[1..] taken so far
        for (int S = 1 << runLevels >> 1; S > 0; S >>= 1) pos += sRun[pos + S] <= i ? S : 0;
        return pos;
    };
    if (RUNS) {
        const int wv_ = (int)(threadIdx.x >> 6);
        if (wv_ == 0 && nq > 0) { const int c0_ = contig_of(0); if (threadIdx.x == 0) sK[0] = c0_; }
        if (wv_ == NW - 1 && nq > 0) { const int c1_ = contig_of(nq - 1); if ((threadIdx.x & 63) == 0) sK[1] = c1_; }
        // the thread's first query.  Up to 63 contigs: the wave's first query finds its run with ONE look at the table, a run
        // start per lane and a ballot, and the lanes inside that run -- all of them unless the wave straddles a run boundary
        // -- are done (the five dependent LDS reads of a bisection per thread made this build slower than the one that loads
        // a contig number per query: 10.9 against 9.5 us at 10^6 queries)
        int cl;
        if (db.nCtg < IGD_WAVE) {
            const int ln_ = (int)(threadIdx.x & 63);
            const int iw = __builtin_amdgcn_readfirstlane(i0);
            const int tv = ln_ <= db.nCtg ? sRun[ln_] : INT_MAX;
            const int cu = __popcll(__ballot(ln_ >= 1 && tv <= iw));       // (sRun[nCtg] = nq > iw whenever iw < nq)
            const int lo_ = __builtin_amdgcn_readlane(tv, cu < db.nCtg ? cu : db.nCtg), hi_ = __builtin_amdgcn_readlane(tv, cu < db.nCtg ? cu + 1 : db.nCtg);
            cl = i0 >= nq ? db.nCtg : ((i0 >= lo_ && i0 < hi_) ? cu : contig_of(i0));
        } else cl = i0 < nq ? contig_of(i0) : db.nCtg;
#pragma unroll
        for (int v = 0; v < VEC; v++) {
            int c = cl;
            while (c < db.nCtg && i0 + v >= sRun[c + 1]) c++;          // (only at a run boundary)
            qc[v] = c;
        }
        if (i0 > 0 && i0 < nq) pc = i0 - 1 >= sRun[cl] ? cl : contig_of(i0 - 1);
    }
#define QB_BASE(c) (FAST ? sBase[c] : (ldsTab ? sBase[c] : db.ctgBase[c]))
#define QB_NTILE(c) (FAST ? sNTile[c] : (ldsTab ? sNTile[c] : db.ctgNTile[c]))
#define QB_TILE(x) (FAST ? tile_shift(x, db.shift) : tile_of(db, x))
    const int lane = threadIdx.x & 63;
    // Once any wave has found the batch unordered nothing this kernel produces is going to be read
    // (the merge join is off, the bucket path keeps its own lists): later workgroups stop here (an unordered batch
    // worked through to the end, gap filling included, took 50 instead of 5 us).
    int w0v[VEC], w1v[VEC];
    int key[VEC], lo[VEC];
    int pend[VEC];                                          // what the scan leaves to the exact walk: list entry of query i0 + v (0: none)
#pragma unroll
    for (int v = 0; v < VEC; v++) pend[v] = 0;
    bool quick = false;
    if (FAST && VEC == 4 && db.vshift < 0) {
        // ---- the short path: the wave's 256 queries and the one before them lie in one contig, inside its tiles, with
        // non-negative starts in non-decreasing order, and none is inverted over its tile's start or longer than four tiles
        const int W = db.nbp, sh = db.shift;
        const int cu = __builtin_amdgcn_readfirstlane(qc[0]);
        int a_[VEC], d_[VEC];
        // (bitwise on purpose: one straight run of compares, no branch per term)
        int ok = (i0 + 3 < nq) & (i0 > 0) & (pc == cu) & (ps >= 0);
        int prev = ps;
#pragma unroll
        for (int v = 0; v < VEC; v++) {
            a_[v] = qs_[v] & (W - 1);
            d_[v] = qe_[v] - (qs_[v] - a_[v]);              // qe - T0
            if (BONLY) ok &= (qc[v] == cu) & (qs_[v] >= prev);
            else
            ok &= (qc[v] == cu) & (qs_[v] >= prev) & ((unsigned)(d_[v] - 1) < (unsigned)(4 * W));   // 0 < qe - T0 <= 4W
            prev = qs_[v];
        }
        if (__builtin_amdgcn_readfirstlane((unsigned)cu < (unsigned)db.nCtg ? 1 : 0)) {
            const int cm = __builtin_amdgcn_readfirstlane(sNTile[cu]) - 1, cb = __builtin_amdgcn_readfirstlane(sBase[cu]);
            ok &= (qs_[VEC - 1] >> sh) <= cm;               // (starts are ordered: the last one's tile bounds them all)
            if (__ballot(ok != 0) == ~0ull) {
                quick = true;
                int pk = cb + (ps >> sh);
#pragma unroll
                for (int v = 0; v < VEC; v++) {
                    const int n1 = qs_[v] >> sh;
                    key[v] = cb + n1;
                    lo[v] = pk + 1;
                    pk = key[v];
                    w1v[v] = 0;
                    if (BONLY) continue;
                    // ~query_word(): low half qe' - 1 = min(qe - T0, W), high half 65535 - qs' = 65534 - (qs - T0)
                    w0v[v] = (d_[v] < W ? d_[v] : W) | ((65534 - a_[v]) << 16);
                    if (d_[v] > W) {                        // reaches beyond its first tile -- unless that is the contig's last
                        int n2 = (qe_[v] - 1) >> sh;
                        if (n2 > cm) n2 = cm;
                        const int sp = n2 - n1;             // 0..3 (d <= 4W)
                        if (sp > 0) {
                            w1v[v] = d_[v] | (sp << 18) | ((key[v] & 3) << 20);
                            spill[key[v] + 1] = epoch;
                            if (sp > 1) { spill[key[v] + 2] = epoch; if (sp > 2) spill[key[v] + 3] = epoch; }
                        }
                    }
                }
            }
        }
    }
    if (!quick) {
    // predecessor of the thread's first query
    int prevKey = -1;
    if (i0 > 0 && i0 < nq) {
        if (FAST) {                                         // tile_key from the staged tables
            if (pc < 0) prevKey = BONLY ? -1 : 0;
            else if (pc >= db.nCtg) prevKey = BONLY ? db.nT : db.nT - 1;
            else {
                const int n1 = tile_shift(ps, db.shift), mT = sNTile[pc] - 1;     // (negative: clamped to tile 0 either way)
                prevKey = sBase[pc] + (n1 < 0 ? 0 : (n1 > mT ? mT : n1));
            }
        } else { prevKey = tile_key(db, pc, ps); if (BONLY) prevKey = pc < 0 ? -1 : (pc >= db.nCtg ? db.nT : prevKey); }
    }
    // 1. keys and order of the thread's queries (query i0 + v fills firstQ[lo[v]..key[v]] = i0 + v)
    int cBase[VEC], cMT[VEC];
    bool unordered = false, notStart = false, brokenStart = false;
    int prevC = pc;
#pragma unroll
    for (int v = 0; v < VEC; v++) {
        const int i = i0 + v;
        lo[v] = 0; key[v] = -1; cBase[v] = 0; cMT[v] = -1;  // (cMT = -1: no tile of this query is in range)
        w0v[v] = packed ? 0 : -1; w1v[v] = 0;
        if (i < nq) {
            const int c = qc[v], s0 = qs_[v];
            const bool cOk = c >= 0 && c < db.nCtg;
            const int cb = cOk ? QB_BASE(c) : 0, cm = cOk ? QB_NTILE(c) - 1 : 0;
            const int n1r = QB_TILE((db.vshift >= 0 && s0 < 0) ? 0 : s0);
            // key(i): global number of the first tile, clamped into the contig (tile_key)
            const int n1c = n1r < 0 ? 0 : (n1r > cm ? cm : n1r);
            const int k = c < 0 ? (BONLY ? -1 : 0) : (c >= db.nCtg ? (BONLY ? db.nT : db.nT - 1) : cb + n1c);
            unordered |= k < prevKey;
            notStart |= k == prevKey && s0 < ps;
            // (the promise is about (contig, start): contig numbers outside the database share the key of the first / last
            // tile here -- their starts are not compared with those of the contig whose tile that is)
            brokenStart |= k == prevKey && c == prevC && s0 < ps;
            prevC = c;
            lo[v] = i == 0 ? k + 1 : prevKey + 1;           // the tiles up to the first query's key: filled by the whole grid (below)
            key[v] = k;
            cBase[v] = cb; cMT[v] = cOk ? cm : -1;
            prevKey = k; ps = s0;
        }
    }
    // 2. One lane per wave reports (hundreds of thousands of stores to ONE address would queue up for tens of
    // microseconds), and a wave that has seen disorder leaves: nothing it would still produce is going to be read.
    {
        // IGD_HIP_FLAG_SORTED promises the order (contig, START): starts that decrease inside a tile break it like anything else,
        // and the batch adds nothing -- whichever step it would have taken (the DIRECT step's rank method needs that order in
        // every tile, its heavy-tile slices in the last launch included: found HERE, before anything is counted).  Without the
        // promise such a batch is a legal merge-join batch: the pairwise compares hold, the rank method is switched off.
        const unsigned long long bu = __ballot(unordered || (promised && brokenStart)), bs = __ballot(notStart);
        if (bs && lane == __builtin_ctzll(bs)) ctl[CTL_NOTSTART] = epoch;   // ordered by tile but not by start inside a tile:
                                                                            // the merge join still holds, the rank method does not
        if (bu) {
            if (lane == __builtin_ctzll(bu)) {
                ctl[CTL_UNSORTED] = epoch;
                if (promised) ctl[CTL_BROKEN] = epoch;      // sticky until the next igd_hip_sync (any promised batch since)
            }
            return;
        }
    }
    // 3. the words the scan reads
    int covA[VEC], covB[VEC];                               // what it leaves to the coverage arrays (covA < 0: nothing)
#pragma unroll
    for (int v = 0; v < VEC; v++) { covA[v] = -1; covB[v] = -1; }
    if (!BONLY)
#pragma unroll
    for (int v = 0; v < VEC; v++) {
        const int i = i0 + v;
        const int s0 = qs_[v], n1 = QB_TILE((db.vshift >= 0 && s0 < 0) ? 0 : s0);   // (re-tiled copy: a start above -nbp of the FILE lies in tile 0)
        if (i < nq && n1 >= 0 && n1 <= cMT[v] && (db.vshift < 0 || real_gate(db, qc[v], s0, vnest))) {
            const int e0 = qe_[v];
            int n2 = QB_TILE((int)((unsigned)e0 - 1u));
            if (n2 > cMT[v]) n2 = cMT[v];
            const int span = n2 > n1 ? n2 - n1 : 0;
            const int g0 = cBase[v] + n1;
            const int T0 = (int)((unsigned)n1 * (unsigned)db.nbp);
            // what the scan kernel leaves to k_exact_walk (the walk applies the visiting rule itself)
            if (n2 - n1 >= IGD_SHORT_TILES) {
                // a long query: its last tile is walked exactly, the tiles n1+4 .. n2-1 are covered from end to end (IGD_COV_*)
                pend[v] = WALK_LAST | (qc[v] << 4);     // (the walk finds the contig here)
                if (n2 - n1 > IGD_SHORT_TILES && !(rule == IGD_HIP_RULE_NEST && db.tileCnt[g0] == 0)) {   // (rule NEST: an empty first tile ends the query)
                    covA[v] = g0 + IGD_SHORT_TILES; covB[v] = g0 + (n2 - n1);    // covered from end to end: tiles covA .. covB - 1
                }
            }
            const bool needExact = packed && e0 <= T0;
            if (needExact) pend[v] = WALK_FIRST | (qc[v] << 4);
            if (!packed) w0v[v] = (g0 << 4) | (span < 15 ? span : 15);
            else {
                // rule NEST (an empty first tile ends the query, :468) needs no look-up here: the first tile's own
                // units have records by definition, and for the later tiles the scan knows from the unit's flags
                // which of the tiles before it are empty
                if (!needExact) w0v[v] = ~query_word(s0, e0, true, T0, db.nbp);
                if (span > 0) {
                    const int sp = span < IGD_SHORT_TILES - 1 ? span : IGD_SHORT_TILES - 1;
                    int rel = e0 - T0;                      // > W here, since the query reaches the next tile
                    if (rel > 4 * db.nbp) rel = 4 * db.nbp;
                    w1v[v] = rel | (sp << 18) | ((g0 & 3) << 20);
                    for (int kk = 1; kk <= sp; kk++) spill[g0 + kk] = epoch;
                }
            }
        }
    }
    // The coverage differences of the wave's long queries.  A position-sorted batch puts the +1 / -1 of neighbouring queries on
    // the same few entries of the difference array: one global atomic per query end made 7 x 10^5 requests to the memory side
    // for 10^6 queries of 100-200 kbp (this kernel: 250 us; 10 without long queries).  They are summed in a window of the
    // wave's own LDS first -- QB_COVW tiles from the first tile its queries cover -- and every entry of the window that is
    // not zero goes out as one atomic, neighbouring entries in one request; an end beyond the window (a sparse batch, a very
    // long query) takes the direct way.  The coarse level (a query crossing a block of 1024 tiles) stays direct: it is rare.
    {
        unsigned long long any = 0;
#pragma unroll
        for (int v = 0; v < VEC; v++) any |= __ballot(covA[v] >= 0);
        if (any) {
            int32_t *win = sCovW + (threadIdx.x >> 6) * QB_COVW;
            for (int k = lane; k < QB_COVW; k += IGD_WAVE) win[k] = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            int first = INT_MAX;
#pragma unroll
            for (int v = 0; v < VEC; v++) if (covA[v] >= 0 && covA[v] < first) first = covA[v];
            for (int o = 32; o > 0; o >>= 1) { const int x = __shfl_xor(first, o); first = x < first ? x : first; }
            int32_t *diff = db.cov + (size_t)(0 * 2 + (epoch & 1)) * IGD_COV_LEN(db.nT), *coarse = diff + db.nT + 2;
#pragma unroll
            for (int v = 0; v < VEC; v++) {
                if (covA[v] < 0) continue;
                const int ta = covA[v], tb = covB[v];
                if ((unsigned)(ta - first) < (unsigned)QB_COVW) atomicAdd(&win[ta - first], 1); else atomicAdd(&diff[ta], 1);
                if ((unsigned)(tb - first) < (unsigned)QB_COVW) atomicAdd(&win[tb - first], -1); else atomicAdd(&diff[tb], -1);
                if ((ta >> IGD_COV_SHIFT) != (tb >> IGD_COV_SHIFT)) { atomicAdd(&coarse[ta >> IGD_COV_SHIFT], 1); atomicAdd(&coarse[tb >> IGD_COV_SHIFT], -1); }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int k = lane; k < QB_COVW; k += IGD_WAVE) {
                const int d = win[k];
                if (d != 0 && first + k <= db.nT + 1) atomicAdd(&diff[first + k], d);
            }
            if (lane == 0) ctl[CTL_COV + 0 * 2 + (epoch & 1)] = epoch;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    }
#undef QB_BASE
#undef QB_NTILE
#undef QB_TILE
    // 4. the workgroup's later-tile words are compacted in query order into its block of later[]: position of each
    // query's (possible) entry = entries of the queries before it in the block.  (A wave that left above is not waited
    // for by the barrier -- and nothing of an unordered batch's block is read.)
    int pos[VEC];
    int total = 0;
    const bool blockLive = !BONLY && packed && (long long)blockIdx.x * (WGT * VEC) < nq;   // (workgroups past the queries only help filling firstQ[])
#pragma unroll
    for (int v = 0; v < VEC; v++) pos[v] = 0;
    int c = 0;
#pragma unroll
    for (int v = 0; v < VEC; v++) c += w1v[v] != 0 ? 1 : 0;
    const int inc = wave_inclusive_sum(c);
    if (lane == 63) sCnt[threadIdx.x >> 6] = inc;
    // the queries listed for the exact walk: counted per wave here, appended per WORKGROUP below
    int myFix = 0;
    if (!quick) {                                           // (a wave on the short path lists nothing)
#pragma unroll
        for (int v = 0; v < VEC; v++) myFix += __popcll(__ballot(pend[v] != 0));
        if (lane == 0 && myFix) { sFixCnt[threadIdx.x >> 6] = myFix; sFixAny = 1; }
    }
    // Once any wave has found the batch unordered nothing this kernel produces is going to be read (the merge join is
    // off, the bucket path keeps its own lists): workgroups that see the mark stop here, before they store anything (an
    // unordered batch worked through to the end, gap filling included, took 50 instead of 5 us).
    if (threadIdx.x == 0) sSeen = seen;
    __syncthreads();
    const bool marked = sSeen == epoch;
    if (marked) return;
    {
        // ONE returning atomic per workgroup for the list of the exact walk.  Requests for one address are served one after
        // the other by its memory channel, ~12 ns each: one per long query -- and still one per wave and pass -- made this
        // kernel take 190-250 us for 10^6 queries of which a quarter or all are long (10 us without).
        if (sFixAny) {                                      // (the same answer in every wave of the workgroup; no: the usual batch)
        const int fm = lane < NW ? sFixCnt[lane] : 0;
        const int fr = wave_inclusive_sum(fm);
        const int ftotal = __builtin_amdgcn_readlane(fr, NW - 1);
        {
            const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
            const int before = wv > 0 ? __builtin_amdgcn_readlane(fr, wv - 1) : 0;
            // (asked for by the first wave that has entries: it is certainly still here -- wave 0 may have left the kernel
            // when it saw the batch out of order)
            if (lane == 0 && before == 0 && myFix > 0) sFixBase = atomicAdd(&ctl[CTL_NFIX + (epoch & 1)], ftotal);
            __syncthreads();
            int at = sFixBase + before;
#pragma unroll
            for (int v = 0; v < VEC; v++) {
                const unsigned long long m = __ballot(pend[v] != 0);
                if (pend[v] != 0) fix[at + __popcll(m & ((1ull << lane) - 1ull))] = make_int2(i0 + v, pend[v]);
                at += __popcll(m);
            }
        }
        }
    }
    if (blockLive && !(IGD_EXP & 8192)) {
        // entries of the waves before this one / of the whole block: one LDS read per lane and a wave scan (the numbers
        // are the same for all lanes of a wave)
        const int mine = lane < NW ? sCnt[lane] : 0;
        const int run = wave_inclusive_sum(mine);
        const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        total = __builtin_amdgcn_readlane(run, NW - 1);
        int off = inc - c + (wv > 0 ? __builtin_amdgcn_readlane(run, wv - 1) : 0);
        int32_t *reg = later + (size_t)blockIdx.x * (WGT * VEC);
#pragma unroll
        for (int v = 0; v < VEC; v++) {
            pos[v] = off;
            if (w1v[v] != 0) reg[off++] = w1v[v];
        }
        if (threadIdx.x == 0) laterHdr[blockIdx.x] = make_int2(total, 0);
    }
    // 5. firstQ (+ lpos): short gaps by the owner, long gaps by the whole wave.  In an ordered batch the gaps add up
    // to at most nT entries; an unordered one would make them add up to nq * nT / 2.  Long gaps are
    // therefore charged to a budget (ctl[CTL_BUDGET + parity], zeroed by the previous batch) and
    // the batch is declared unsorted -- which it then certainly is -- once the budget is spent.
    // A wave that finds the batch already marked fills nothing: firstQ[] is not going to be used.  (One that saw
    // disorder itself has left above.)
    if (!marked && !(IGD_EXP & 0x40000)) {
#pragma unroll
        for (int v = 0; v < VEC; v++) {
            const int i = i0 + v, l1 = lo[v], h1 = key[v], p1 = pos[v];
            const bool some = h1 >= l1;
            if (quick && __ballot(some) == 0) continue;     // (a dense batch: most queries share their tile with the one before)
            const bool big = h1 - l1 >= 8;
            if (!big) for (int tt = l1; tt <= h1; tt++) { firstQ[tt] = i; if (!BONLY && !(IGD_EXP & 0x800000)) lpos[tt] = p1; }
            unsigned long long m = __ballot(big);
            if (m == 0) continue;
            // the budget is charged once for all long gaps of the wave's 64 queries (a returning atomic each made a small
            // batch, whose every gap is long, wait 64 times in a row)
            int charge = big && h1 - l1 >= 256 ? (h1 - l1) >> 8 : 0;
            for (int o = 32; o > 0; o >>= 1) charge += __shfl_xor(charge, o);
            bool over = false;
            if (charge) {
                int spent = 0;
                if (lane == 0) spent = atomicAdd(&ctl[CTL_BUDGET + (epoch & 1)], charge);
                spent = __builtin_amdgcn_readfirstlane(spent);
                if (spent + charge > (db.nT >> 8) + 16) {
                    over = true;
                    if (lane == 0) {
                        ctl[CTL_UNSORTED] = epoch;
                        if (promised) ctl[CTL_BROKEN] = epoch;
                    }
                }
            }
            while (m) {
                const int src = __builtin_ctzll(m);
                m &= m - 1;
                const int l2 = __builtin_amdgcn_readlane(l1, src), h2 = __builtin_amdgcn_readlane(h1, src);
                const int v2 = __builtin_amdgcn_readlane(i, src), p2 = __builtin_amdgcn_readlane(p1, src);
                if (over && h2 - l2 >= 256) continue;
                for (int tt = l2 + lane; tt <= h2; tt += IGD_WAVE) { firstQ[tt] = v2; if (!BONLY) lpos[tt] = p2; }
            }
        }
    }
    if (BONLY) { }
    else if (VEC == 4) {
        if (i0 + 3 < nq) *(int4 *)(qw0 + i0) = make_int4(w0v[0], w0v[1 % VEC], w0v[2 % VEC], w0v[3 % VEC]);
        else {
#pragma unroll
            for (int v = 0; v < VEC; v++)
                if (i0 + v < nq) qw0[i0 + v] = w0v[v];
        }
    } else if (i0 < nq) qw0[i0] = w0v[0];
    // head and tail of firstQ[] -- the tiles up to the first query's key and after the last one's, together all the
    // tiles a batch does not reach (7/8 of them for one GPU's slab of an 8-GPU job) -- are filled by the whole grid:
    // left to the first / last query's own wave they took longer than everything else in this kernel.  lpos[] of the
    // first three tiles after the last query's (all that a query can still reach) = the entries of the last block, if the
    // queries end inside it: written by that block's own workgroup.
    if (nq > 0 && !(IGD_EXP & 0x80000)) {
        // (tile_key over the staged tables when there are any: no look-up in global memory on the way out)
        auto edge_key = [&](int c, int q) -> int {
            if (BONLY && c < 0) return -1;
            if (BONLY && c >= db.nCtg) return db.nT;
            if (!ldsTab) return tile_key(db, c, q);
            if (c < 0) return 0;
            if (c >= db.nCtg) return db.nT - 1;
            int n1 = tile_of(db, q);
            const int mT = sNTile[c] - 1;
            n1 = n1 < 0 ? 0 : (n1 > mT ? mT : n1);
            return sBase[c] + n1;
        };
        const int k0 = edge_key(RUNS ? sK[0] : edgeC0, edgeS0), kl = edge_key(RUNS ? sK[1] : edgeC1, edgeS1);
        const int nth = gridDim.x * WGT;
        for (int tt = t; tt <= k0; tt += nth) { firstQ[tt] = 0; if (!BONLY) lpos[tt] = 0; }
        for (int tt = kl + 1 + t; tt <= db.nT; tt += nth) firstQ[tt] = nq;
        if (!BONLY && (int)blockIdx.x == (nq - 1) / (WGT * VEC) && (int)threadIdx.x < IGD_SHORT_TILES - 1 && kl + 1 + (int)threadIdx.x <= db.nT)
            lpos[kl + 1 + threadIdx.x] = (nq % (WGT * VEC)) != 0 ? total : 0;
    }
}

// Bucket path (any query order).  `gate`: 0 = always run; otherwise run only when
// ctl[CTL_UNSORTED] == gate, i.e. when k_query_bounds found this batch unsorted.
// Queries the bucket path does not turn into pairs: long ones, and (compact image) first-tile
// queries with qe <= tile start.  Returns the exact-walk kind or -1.
__device__ __forceinline__ int walk_kind(const DbView &db, int qs, int qe, int ntl, int packed)
{
    if (db.vshift >= 0 && qs < 0) qs = 0;                 // (re-tiled copy: see query_span)
    if (ntl > IGD_SHORT_TILES) return WALK_ALL;
    if (packed && qe <= (int)((unsigned)tile_of(db, qs) * (unsigned)db.nbp)) return WALK_FIRST;
    return -1;
}

// step 1: per-tile pair counts (+ the exact-walk list)
__global__ void k_count_pairs(DbView db, const int32_t *__restrict__ ichr,
                              const int32_t *__restrict__ qs, const int32_t *__restrict__ qe,
                              int nq, int rule, int packed, int32_t *__restrict__ pairCnt,
                              int2 *__restrict__ longList, int32_t *__restrict__ ctl,
                              int gate, int epoch, u64 *__restrict__ zeroHits, u64 *__restrict__ zeroTotal)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (zeroHits && i < db.nFiles) zeroHits[i] = 0;        // IGD_HIP_FLAG_ZERO_FIRST (bucket-only mode)
    if (zeroTotal && i == 0) *zeroTotal = 0;
    if (gate == 0 && i == 0) {
        ctl[CTL_NLONG + ((epoch + 1) & 1)] = 0;
        ctl[CTL_NFIX + ((epoch + 1) & 1)] = 0;
        ctl[CTL_BUDGET + ((epoch + 1) & 1)] = 0;
        ctl[CTL_NHEAVY + ((epoch + 1) & 1)] = 0;
        ctl[CTL_NHEAVYS + ((epoch + 1) & 1)] = 0;
        ctl[CTL_NFAR + ((epoch + 1) & 1)] = 0;
    }
    if (gate != 0 && ctl[CTL_UNSORTED] != gate) return;
    if (i >= nq) return;
    int gt0, ntl;
    if (!query_span(db, ichr[i], qs[i], qe[i], rule, gt0, ntl)) return;
    const int kind = walk_kind(db, qs[i], qe[i], ntl, packed);
    if (kind >= 0) {
        longList[atomicAdd(&ctl[CTL_NLONG + (epoch & 1)], 1)] = make_int2(i, kind);
        if (kind == WALK_ALL) cover_tiles(db, ctl, 1, epoch, gt0 + 1, gt0 + ntl - 1);   // first and last tile by the walk, the rest covered
        return;
    }
    for (int k = 0; k < ntl; k++)
        if (db.tileCnt[gt0 + k] > 0) atomicAdd(&pairCnt[gt0 + k], 1);
}
