// igd_hip.hip -- MI355X (gfx950 / CDNA4) overlap-search engine behind include/igd_hip.h.
//
// Replaces, for a whole batch of queries at once, the reference's per-query kernels
//   get_overlaps    /root/reference/src/igd_search.c:454-534   (rule NEST)
//   get_overlaps_v  /root/reference/src/igd_search.c:623-694   (rule FLAT, value>=v)
//   get_overlaps0   /root/reference/src/igd_search.c:30-112    (12-byte records)
//   get_overlaps_f1 /root/reference/src/igd_search.c:537-620   (enumeration)
// and the hits[] accumulator (:925, :491, :524, :654, :684).
//
// Design (DESIGN.md has the long form).  The reference walks queries and, per query, seeks
// a tile, bisects it for the last start<qe and scans backwards testing end>qs.  Here the
// loop is turned inside out so that BOTH sides stream:
//   1. group the queries by tile.  Two ways, chosen ON THE DEVICE per batch:
//      a. queries already ordered by (contig, start) -- what a position-sorted BED gives:
//         k_query_bounds finds, in one pass and without atomics, the first query of every
//         tile (firstQ[]), verifies the order and leaves per query its READY-MADE compare word for
//         its first tile (qw0) and, for the 6 % that reach into later tiles, one more word, compacted
//         per block of 1024 queries (later[]); the scan kernel then reads the words of tile t's
//         queries, and of those of the up-to-3 tiles before it that reach it, with the tile's
//         records (a merge join: both sides stream, nothing is computed per candidate);
//      b. any other order: every (query, visited tile) pair is grouped by tile id without global
//         atomics (k_split_local -> k_split_fine: LDS counting in two levels);
//         these kernels return at once when (a) holds.
//      The NEST/FLAT visiting rule and the "first tile" notion live entirely in this step.
//   2. scan:   igd_scan_sorted (a) / igd_scan_tiles (b) -- one wavefront owns one <=320-record chunk
//      ("unit") of one tile at a time.  It loads the unit's records once, coalesced, into 5 register
//      slots (record r*64+lane) -- by default from a compact 6-byte tile-relative image
//      (k_pack_units) -- keeps two units in flight, and runs through the tile's queries 64 at a
//      time across the lanes.  Each slot has a summary word (component-wise max of its 64 record
//      words, kept in the unit's descriptor); the queries that pass against it -- one vector compare
//      for all 64 -- are broadcast with v_readlane one at a time, and per query and slot the test
//            lob <= start < qe  &&  end > qs  [&& value >= v]
//      (lob = tile start for a non-first tile: the reference's tS prefix skip, :510-511; the
//      upper bound start<qe is what its bisection computes, :479-487) is one packed 16-bit max,
//      one compare and one add-with-carry into a per-record hit count; the counts go into a
//      per-workgroup LDS copy of hits[] (privatised counters).  Tiles with >= 32 queries are counted
//      by ranks instead (two bisections per query/record: see "rank" at igd_scan_sorted).
//   3. flush/reduce: each workgroup stores its LDS counters to its own slab row with plain
//      coalesced stores; k_reduce_slabs sums the rows into the caller's int64 hits[].  The same
//      launch walks, on the exact arrays, the few queries the scan leaves out (more than
//      IGD_SHORT_TILES tiles long, or needing exact starts: see k_pack_units), and shares out over
//      all its waves the tiles that were listed as too heavy for the one wave that owns them
//      (heavy_bucket_body / heavy_sorted_body: the skew valves).
//   `-f` (igd_enum_queries: query-major, streamed out in chunks) and `-m` (igd_hitmap_tiles) are separate kernels on the exact arrays;
//   Seqpare `-s` (igd_hip_seqpare) = the `-f` kernel emitting similarities + radix sorts into the
//   greedy order (igd_sortscan.hpp) + a wave-per-group matching kernel (k_seq_greedy).
// No MFMA anywhere: this is integer compare + count, bound by HBM / VALU issue, not by math.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <vector>
#include <thread>
#include <mutex>
#include <time.h>
#include <unistd.h>

#include "igd_hip.h"
#include "igd_sortscan.hpp"

#define IGD_WAVE 64
#ifndef IGD_SLOTS
#define IGD_SLOTS 5                          // register slots per array per lane
#endif
#define IGD_CHUNK (IGD_WAVE * IGD_SLOTS)     // records per work chunk (320)
#define IGD_SHORT_TILES 4                    // queries spanning more tiles take the long path
#ifndef IGD_WPE
#define IGD_WPE 8                            // scan kernel: waves per SIMD the register budget is cut for
#endif
#ifndef IGD_WG
#define IGD_WG 1024                          // threads per scan workgroup (16 waves; 2 workgroups per CU)
#endif
#define IGD_MAX_BATCH (1ll << 24)            // queries per device batch
#define IGD_SCAN_ITEMS 16                    // elements per thread in the tile scan
#define IGD_SCAN_BLOCK 256
#define IGD_SCAN_TILE (IGD_SCAN_ITEMS * IGD_SCAN_BLOCK)
#ifndef IGD_TAIL_WGS
#define IGD_TAIL_WGS 256                     // workgroups (IGD_TAIL_WG threads) of the batch's last launch: one per CU -- with the ~100 registers of the
                                             // long queries' four-deep walks a CU holds one anyway, and a second round of workgroups cost every batch 1 us
#endif
#ifndef IGD_TAIL_WG
#define IGD_TAIL_WG 1024                     // ... in workgroups of 16 waves: what the long queries' work counts in a workgroup's LDS
                                             // leaves it as one global atomic per dataset, and 2048 workgroups of 4 waves made 3.9 x 10^6 of those
#endif
#ifndef IGD_REDUCE_GROUPS
#define IGD_REDUCE_GROUPS 64                 // (32-bit slab rows: 6.2 us with 128 groups, 5.4 with 64, 6.4 with 32, 10.1 with 16)
#endif
#define IGD_PIPE_EVENTS 16                   // profiling: launches whose whole pipeline (not just the scan kernel) is timed
#define IGD_WINDOW_FILES 10240
#define IGD_MAX_WINDOWS 16                   // passes over the files of a database with more files than LDS counters
#define IGD_LDS_HITS_MAX_BYTES (120 * 1024)   // + 37 KiB of rank-method areas (igd_scan_sorted) stays below 160 KiB

typedef unsigned long long u64;

// tuning / experiment knobs (defaults are the shipped configuration)
#ifndef IGD_BUFFER_LOADS
#define IGD_BUFFER_LOADS 1    // compact image read with bounds-checked buffer loads (descriptor per unit)
#endif
#ifndef IGD_EXP_NOMATCH
#define IGD_EXP_NOMATCH 0     // measurement only: load everything, compare nothing (wrong results)
#endif

#ifndef IGD_EXP
#define IGD_EXP 0      // measurement-only builds (bits 1..512 and 8192 give WRONG counts): 1 no LDS flush, 2 no per-query compares,
                       // 4 no compares at all, 8 no later-tile queries, 64/128/256 rank method without term B / the searches of
                       // term A / prefix sums, 512 later-tile queries found but not searched, 8192 k_query_bounds without the
                       // compaction of the later-tile words; 32 time stamps per wave (tools/stamps.py), 1024 section timers of the
                       // rank method (printed by igd_hip_close); 0x10000 / 0x20000 the last launch without heavy_sorted_body /
                       // far_units_body (WRONG counts)
#endif
#ifndef IGD_NT_AUX
#define IGD_NT_AUX 0   // cache policy of igd_scan_sorted's record loads (measured: 2 = nt is 6 % slower -- consecutive batches find part of the image in the Infinity Cache)
#endif
#ifndef IGD_OPT_PRIO
#define IGD_OPT_PRIO 1 // igd_scan_sorted: waves lower their issue priority as they get through their share
#endif
#if IGD_EXP & 1024
// diagnostic build: the waves' time (s_memtime ticks) in the sections of the rank method, summed over all launches
__device__ u64 d_sect[8];
#define SECT(i) do { const u64 t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) atomicAdd(&hist[321 + (i)], (unsigned)(t_ - tsec)); tsec = t_; } while (0)   /* per wave, in spare words of its LDS histogram */
#else
#define SECT(i) do { } while (0)
#endif
#if IGD_EXP & 32
static u64 *g_stamps = nullptr;     // diagnostic build: s_memtime stamps of the last igd_scan_sorted launch
static int g_stampWaves = 0;
#endif

// ------------------------------------------------------------------------------------------
// error plumbing
static thread_local char g_err[512] = "";
static void set_err(const char *what, hipError_t e, const char *file, int line)
{
    snprintf(g_err, sizeof g_err, "%s: %s (%s:%d)", what, hipGetErrorString(e), file, line);
}
#define HIPCHK(call)                                                          \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) {                                               \
            set_err(#call, e_, __FILE__, __LINE__);                           \
            return IGD_HIP_ERR_DEVICE;                                        \
        }                                                                     \
    } while (0)

extern "C" const char *igd_hip_last_error(void) { return g_err; }
extern "C" void igd_hip_set_error_(const char *msg) { snprintf(g_err, sizeof g_err, "%s", msg); }   // igd_create.hip

extern "C" int igd_hip_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        set_err("hipGetDeviceCount", e, __FILE__, __LINE__);
        return 0;
    }
    return n;
}

// Queries per engine call of the HOST-buffer entry points (igd_hip_search*, igd_hip_enumerate*, igd_hip_seqpare*) and the
// step of the command line tool's loops over longer query files.  IGD_MAX_BATCH in production; the TEST-ONLY variable
// IGD_HIP_MAX_BATCH (read once per process) lowers it so that every multi-batch seam -- the loop of igd_hip_search_ex, the
// `-f` and `-s` loops of igd_cli_abi.c -- is crossed by ordinary small fixtures (tests/test_gpu_batches.py).
static int64_t max_batch(void)
{
    static const int64_t m = []() -> int64_t {
        const char *e = getenv("IGD_HIP_MAX_BATCH");
        const long long x = e && *e ? atoll(e) : 0;
        return x >= 1 && x < IGD_MAX_BATCH ? (int64_t)x : (int64_t)IGD_MAX_BATCH;
    }();
    return m;
}
extern "C" int64_t igd_hip_max_batch(void) { return max_batch(); }

// What this library was compiled as: bits 0..23 = IGD_EXP, bit 24 = IGD_EXP_NOMATCH.  A build with any bit of
// IGD_HIP_BUILD_WRONG_COUNTS gives WRONG counts on purpose (measurement of kernel sections): igd_hip_open refuses to
// work in such a build unless IGD_HIP_ALLOW_EXP_BUILD=1 says the caller knows (tools/valu_ab.sh does).
#define IGD_EXP_WRONG_BITS (1 | 2 | 4 | 8 | 64 | 128 | 256 | 512 | 8192 | 0x10000 | 0x20000)
extern "C" unsigned igd_hip_build_flags(void) { return ((unsigned)IGD_EXP & 0xffffffu) | (IGD_EXP_NOMATCH ? 1u << 24 : 0u); }
extern "C" unsigned igd_hip_build_wrong_counts(void) { return (((unsigned)IGD_EXP) & (unsigned)IGD_EXP_WRONG_BITS) | (IGD_EXP_NOMATCH ? 1u << 24 : 0u); }

// ------------------------------------------------------------------------------------------
// device view of one database (passed to kernels by value)
// One unit of scan work: a chunk of <= IGD_CHUNK records of one tile (48 bytes, three dwordx4 loads).
// Every tile has at least one unit; an empty tile gets a placeholder with n == 0 so that the
// long queries that START in it still have an owner in the sorted path.
struct __attribute__((aligned(16))) Unit {
    int64_t off;      // index of the unit's first record in the SoA arrays
    int32_t tile;     // global tile id
    int32_t n;        // records in this unit
    int32_t jf;       // (j << 4) | flags; j = tile index inside its contig;
                      // flag bit 0: first unit of its tile; bit k (1..3): tile j-k of the contig is EMPTY
    uint32_t W[6];    // compact image: summary word of each 64-record slot (k_pack_units): the component-wise
                      // maximum of the slot's record words = (65535 - smallest s') | largest e' << 16
    int32_t pre;      // compact image: the unit's records that start BEFORE its tile (s' = 0) -- the first `pre` of it, the tile
                      // being ordered by start; a tile covered from end to end counts the others without looking at them
};
#define UNIT_J(u) ((u).jf >> 4)
#define UNIT_FLAGS(u) ((u).jf & 15)

struct SpTuple { int32_t t, s, e; };     // split path: one (tile, qs, qe) pair on its way to its tile: 12 bytes, one dwordx3 access

struct DbView {
    int32_t nbp, shift, nCtg, nT, nUnits, nFiles;
    const Unit *units;
    const int32_t *start, *end, *idx, *value;   // SoA over all records, file order (exact)
    // compact tile-relative image of the same records (6 bytes each), see k_pack_units:
    const uint32_t *pse;                        // s' | e' << 16
    const uint16_t *px;                         // idx
    const uint32_t *pxv;                        // idx | value << 16 (only when every value fits 16 bits)
    const int64_t *tileOff;                     // [nT+1] record offset of each tile
    const int32_t *tileCnt;                     // [nT]
    const int32_t *tileBd;                      // [nT] tile start coordinate j*nbp (INT_MIN for j==0)
    const int32_t *ctgBase;                     // [nCtg] global tile id of the contig's tile 0
    const int32_t *ctgNTile;                    // [nCtg]
    const int32_t *tileUnit0;                   // [nT+1] number of each tile's first unit
    int32_t *cov;                               // workspace: 4 sets of coverage difference arrays (IGD_COV_*)
    // A database whose file is bucketed with another tile width than the image likes (-b 11..13, 16..19) is searched over a
    // RE-TILED copy: the same records in tiles of 2^14 bp (igd_hip_db::inner).  Which records a query counts does not depend on
    // the tile width -- every record that overlaps it, once -- with two exceptions that are properties of the FILE's tiles: a
    // query whose first tile lies outside the contig's tiles counts nothing (:462), and under rule NEST neither does one whose
    // first tile is empty (:468).  vshift >= 0 marks such a copy: log2 of the file's tile width; rNTile / rBase / rEmpty describe
    // the file's tiles (per contig: their number and the number of the first one; one bit per tile: it holds no record).
    int32_t vshift;
    const int32_t *rNTile, *rBase;
    const uint32_t *rEmpty;
    int32_t fileLo;                             // scan kernels, WIN builds: the pass counts the files [fileLo, fileLo + nFiles) (see igd_hip_db::winN)
};

struct igd_hip_db {
    int device;
    int32_t nbp, gType, nCtg, nFiles, nT;
    int64_t nRec;
    DbView v;
    // owned device memory of the image
    int32_t *d_start, *d_end, *d_idx, *d_value;
    uint32_t *d_pse;
    uint16_t *d_px;
    uint32_t *d_pxv;
    bool packed, packedV;         // compact image usable (nbp<=32768, nFiles<=65536) / values fit int16
    int64_t *d_tileOff;
    int32_t *d_tileCnt, *d_tileBd, *d_ctgBase, *d_ctgNTile, *d_tileUnit0;
    int32_t *d_heavy;             // [IGD_HEAVY_MAX] tiles of the batch with more pairs than a wave should take alone
    int32_t *d_far;               // [nUnits + 1] units the lean build of igd_scan_sorted leaves to far_units_body
    Unit *d_units;
    int32_t nUnits;
    int64_t resident;
    int32_t *d_firstQ, *d_pairN;  // [nT+1] first query of each tile (sorted path); pair counts copy
    int32_t *d_lpos;              // [nT+1] lpos[]: entries of its later block before query firstQ[t] (k_query_bounds)
    int32_t *d_cov;               // coverage of long queries (IGD_COV_*): 4 sets (path x batch parity) of { diff[nT + 2], coarse[(nT >> IGD_COV_SHIFT) + 2] }
    bool covStale;                // a batch returned an error after its first kernel: clear d_cov before the next one
    igd_hip_db *inner;            // the re-tiled copy the counting searches run on (file bucketed with -b 11..13 / 16..19), else null
    uint32_t *d_rEmpty;           // ... and its bitmap of the FILE's empty tiles
    int vnest;                    // (set on a re-tiled copy by the call that forwards to it: rule NEST of that call)
    int32_t *d_runIchr;           // igd_hip_search_runs_dev: contig numbers written out for the batches the RUNS build does not take
    int64_t runCap;
    int32_t *d_qw;                // [wsQueries] per-query word of the merge join (k_query_bounds: qw0)
    int32_t *d_later;               // [wsQueries + 1088] later[]: later-tile words, compacted per later block (compact image only)
    int32_t *d_spill;             // [nT+1] epoch stamps: a query covers the tile as a later tile
    int ldsSorted;                // dynamic LDS of igd_scan_sorted: counters + the waves' rank-method areas
    int32_t maxTileCnt;           // records of the fullest tile
    int sbCap;                    // igd_scan_sorted, rank method: query starts of one tile a wave keeps in LDS
    int32_t *d_laterHdr;         // laterHdr[]: int2 per later block (entries, last tile covered as a later tile)
    int lbShift;                  // log2(queries per later block) of the batch in flight
    int lastMode, lastPacked;     // of the last batch (igd_hip_last_scan_kernel)
    int forceRank;                // IGD_HIP_RANK at open (tests): 0 lean build, 1 full build, -1 the engine decides
    bool bigImage;                // the compact image is addressed with per-unit 64-bit bases (>= 2^30 tile records; IGD_HIP_BIG=1 at open: tests)
    bool qbVec1, timing;          // IGD_HIP_QB_VEC1 (A/B), IGD_TIMING at open: no getenv on the per-batch path
    uint32_t *d_spTable;          // split path: [nWG][nCoarse] offset | count << 16
    SpTuple *d_spT;               // regions: the pairs of each k_split_local workgroup, grouped by coarse bucket
    uint32_t *d_spSub;            // piled-up buckets: [nCoarse][SPF_S][2^spShift] pairs per (bucket, share, tile), then bucketBase[nCoarse], bucketLong[nCoarse]
    int spShift, spCoarse;        // coarse bucket = tile >> spShift; spShift < 0: split path not applicable
    char *arena;                  // one hipMalloc holds the whole resident image (carved by dalloc)
    size_t arenaSize, arenaUsed;
    int32_t epoch;                // batch counter: device-side flags are compared against it
    int32_t promised;             // != 0: a batch was launched under IGD_HIP_FLAG_SORTED since the last igd_hip_sync
    // per-batch workspace
    int32_t *d_pairCnt, *d_pairPos, *d_blockSums;
    void *d_pairs;                // int2[cap*K] (or int4 for the enumerate path)
    int2 *d_long, *d_fix;         // exact-walk lists: bucket path / merge-join path
    int32_t *d_ctl;               // control words (CTL_*)
    int64_t wsQueries;            // capacity in queries (merge-join arrays)
    int64_t wsBucket;             // capacity in queries of the bucket-path structures
    int pairBytes;
    u64 *d_slab;
    int grid, ldsBytes;
    bool ldsHits;
    int winN, nWin;               // files per window / windows (1: the files fit the LDS counters, or there are too many for windows)
    // `-f` streaming workspace (created by the first enumeration, kept)
    int64_t *d_qcount, *d_qoff, *d_enumBsum;
    int64_t enumQCap, enumChunkCap;          // capacity in queries / overlaps per chunk buffer
    igd_hip_hit *d_enumOut[2], *h_enumPin[2];
    bool enumPinned;
    hipStream_t copyStream;
    hipEvent_t evFill[2], evCopy[2];
    // host-API staging
    int32_t *d_qc, *d_qs, *d_qe;
    int64_t qcap;
    int64_t *d_hits, *d_total;
    hipStream_t stream;
    // profiling
    std::vector<hipEvent_t> ev;   // 4 per launch: pipeline start, scan start, scan stop, pipeline stop
    int evMax, evUsed;
    hipEvent_t evStart, evStop;   // set while a timed launch of a promised-sorted batch is being enqueued: the scan kernel's own dispatch is
                                  // bracketed by them (hipExtLaunchKernel), not two event packets around it
    int evEvery, evSeen;          // every evEvery-th launch is timed (igd_hip_profile_sampling)
    bool evOn;
};

// ------------------------------------------------------------------------------------------
// upload: AoS (file order) -> SoA.  One record per lane: a 16-byte gdata_t is one dwordx4
// load (src/igd_base.h:41-46: idx,start,end,value); 12-byte gdata0_t three dword loads.
__global__ void k_aos_to_soa16(const int4 *__restrict__ aos, int64_t n, int32_t *__restrict__ start,
                               int32_t *__restrict__ end, int32_t *__restrict__ idx,
                               int32_t *__restrict__ value)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        int4 r = aos[i];
        idx[i] = r.x; start[i] = r.y; end[i] = r.z; value[i] = r.w;
    }
}
__global__ void k_aos_to_soa12(const int32_t *__restrict__ aos, int64_t n, int32_t *__restrict__ start,
                               int32_t *__restrict__ end, int32_t *__restrict__ idx)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        idx[i] = aos[3 * i]; start[i] = aos[3 * i + 1]; end[i] = aos[3 * i + 2];
    }
}

// hits[] is indexed by idx without a bounds check in the reference (src/igd_search.c:491); on
// the GPU an out-of-range idx would corrupt LDS, so the image is validated once at open.
__global__ void k_idx_range(const int32_t *__restrict__ idx, int64_t n, int32_t nFiles, int32_t *__restrict__ bad)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int b = 0;
    for (; i < n; i += stride) b |= (idx[i] < 0) | (idx[i] >= nFiles);
    if (b) atomicOr(bad, 1);
}

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
__device__ __forceinline__ bool query_span(const DbView &db, int c, int qs, int qe, int rule,
                                           int &gt0, int &ntl)
{
    if (c < 0 || c >= db.nCtg) return false;
    if (db.vshift >= 0) {                     // a re-tiled copy: the file's tiles decide whether the query counts at all ...
        if (!real_gate(db, c, qs, (rule >> 8) & 1)) return false;
        if (qs < 0) qs = 0;                   // ... and a start before the contig (above -nbp of the file) lies in tile 0
        rule &= 0xff;                         // (the copy's own tiles are visited under rule FLAT)
    }
    int n1 = tile_of(db, qs);
    int n2 = tile_of(db, (int)((unsigned)qe - 1u));
    int mT = db.ctgNTile[c] - 1;
    if (n1 < 0 || n1 > mT) return false;      // n1<0: out-of-bounds read in the reference
    if (n2 > mT) n2 = mT;
    gt0 = db.ctgBase[c] + n1;
    if (rule == IGD_HIP_RULE_NEST && db.tileCnt[gt0] == 0) return false;
    ntl = n2 > n1 ? n2 - n1 + 1 : 1;
    return true;
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
#define IGD_HEAVY_FIRST 8192   // merge join: a tile with more first-tile queries than this is shared out in slices of 4096
#define IGD_HEAVY_SLICE 4096
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
template <int VEC, bool FAST, int WGT, bool RUNS = false>
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
            s4 = *(const int4 *)(qs + i0); e4 = *(const int4 *)(qe + i0);
        } else {
            if (i0 < nq) { if (!RUNS) c4.x = ichr[i0]; s4.x = qs[i0]; e4.x = qe[i0]; }
            if (i0 + 1 < nq) { if (!RUNS) c4.y = ichr[i0 + 1]; s4.y = qs[i0 + 1]; e4.y = qe[i0 + 1]; }
            if (i0 + 2 < nq) { if (!RUNS) c4.z = ichr[i0 + 2]; s4.z = qs[i0 + 2]; e4.z = qe[i0 + 2]; }
        }
        qc[0] = c4.x; qc[1 % VEC] = c4.y; qc[2 % VEC] = c4.z; qc[3 % VEC] = c4.w;
        qs_[0] = s4.x; qs_[1 % VEC] = s4.y; qs_[2 % VEC] = s4.z; qs_[3 % VEC] = s4.w;
        qe_[0] = e4.x; qe_[1 % VEC] = e4.y; qe_[2 % VEC] = e4.z; qe_[3 % VEC] = e4.w;
    } else if (i0 < nq) { qc[0] = RUNS ? 0 : ichr[i0]; qs_[0] = qs[i0]; qe_[0] = qe[i0]; }
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
                    // ~query_word(): low half qe' - 1 = min(qe - T0, W), high half 65535 - qs' = 65534 - (qs - T0)
                    w0v[v] = (d_[v] < W ? d_[v] : W) | ((65534 - a_[v]) << 16);
                    w1v[v] = 0;
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
            if (pc < 0) prevKey = 0;
            else if (pc >= db.nCtg) prevKey = db.nT - 1;
            else {
                const int n1 = tile_shift(ps, db.shift), mT = sNTile[pc] - 1;     // (negative: clamped to tile 0 either way)
                prevKey = sBase[pc] + (n1 < 0 ? 0 : (n1 > mT ? mT : n1));
            }
        } else prevKey = tile_key(db, pc, ps);
    }
    // 1. keys and order of the thread's queries (query i0 + v fills firstQ[lo[v]..key[v]] = i0 + v)
    int cBase[VEC], cMT[VEC];
    bool unordered = false, notStart = false;
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
            const int k = c < 0 ? 0 : (c >= db.nCtg ? db.nT - 1 : cb + n1c);
            unordered |= k < prevKey;
            notStart |= k == prevKey && s0 < ps;
            lo[v] = i == 0 ? k + 1 : prevKey + 1;           // the tiles up to the first query's key: filled by the whole grid (below)
            key[v] = k;
            cBase[v] = cb; cMT[v] = cOk ? cm : -1;
            prevKey = k; ps = s0;
        }
    }
    // 2. One lane per wave reports (hundreds of thousands of stores to ONE address would queue up for tens of
    // microseconds), and a wave that has seen disorder leaves: nothing it would still produce is going to be read.
    {
        const unsigned long long bu = __ballot(unordered), bs = __ballot(notStart);
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
    const bool blockLive = packed && (long long)blockIdx.x * (WGT * VEC) < nq;   // (workgroups past the queries only help filling firstQ[])
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
    if (!marked) {
#pragma unroll
        for (int v = 0; v < VEC; v++) {
            const int i = i0 + v, l1 = lo[v], h1 = key[v], p1 = pos[v];
            const bool some = h1 >= l1;
            if (quick && __ballot(some) == 0) continue;     // (a dense batch: most queries share their tile with the one before)
            const bool big = h1 - l1 >= 8;
            if (!big) for (int tt = l1; tt <= h1; tt++) { firstQ[tt] = i; lpos[tt] = p1; }
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
                for (int tt = l2 + lane; tt <= h2; tt += IGD_WAVE) { firstQ[tt] = v2; lpos[tt] = p2; }
            }
        }
    }
    if (VEC == 4) {
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
    if (nq > 0) {
        // (tile_key over the staged tables when there are any: no look-up in global memory on the way out)
        auto edge_key = [&](int c, int q) -> int {
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
        for (int tt = t; tt <= k0; tt += nth) { firstQ[tt] = 0; lpos[tt] = 0; }
        for (int tt = kl + 1 + t; tt <= db.nT; tt += nth) firstQ[tt] = nq;
        if ((int)blockIdx.x == (nq - 1) / (WGT * VEC) && (int)threadIdx.x < IGD_SHORT_TILES - 1 && kl + 1 + (int)threadIdx.x <= db.nT)
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

// ------------------------------------------------------------------------------------------
// Bucket path without global atomics ("split"): 10^6 random atomicAdds on the per-tile counters cost
// ~40 us each way on this part (device-scope atomics are resolved beyond the XCD-private L2s), so the
// (query, tile) pairs are grouped in two levels with LDS atomics only:
//   k_split_local   a workgroup takes 1024 queries, counts their pairs per COARSE bucket (tile >> shift,
//                   <= 1024 buckets) in LDS, and writes them, grouped by bucket, into its own region
//                   + one table row (offset | count << 16 per bucket);
//   k_split_fine    one workgroup per bucket collects the bucket's segments from all regions, counts per
//                   tile in LDS (the bucket spans 2^shift tiles), writes pairN/pairPos of its tiles and
//                   the pairs, tile by tile, into `pairs`.
// Output = exactly what count/scan/scatter leave (pairN, pairPos = END of each tile's range, pairs).
#ifndef SP_WG
#define SP_WG 1024
#endif
#ifndef SP_PER
#define SP_PER 4
#endif
#define SP_Q (SP_WG * SP_PER) // queries per workgroup of k_split_local: more queries = longer segments per (workgroup, bucket)
#ifndef SPF_WG
#define SPF_WG 256           // threads of k_split_fine
#endif
#define SP_CAP (SP_Q * IGD_SHORT_TILES)
#define SP_LONG (SP_Q / 4)     // pairs of one k_split_local workgroup in one coarse bucket from which the bucket counts as piled up
#define SP_MAXC 1024

__global__ __launch_bounds__(SP_WG) void k_split_local(DbView db, const int32_t *__restrict__ ichr,
                                                       const int32_t *__restrict__ qs, const int32_t *__restrict__ qe,
                                                       int nq, int rule, int packed, int shift, int nCoarse,
                                                       uint32_t *__restrict__ table, SpTuple *__restrict__ reg,
                                                       int2 *__restrict__ longList, int32_t *__restrict__ ctl, int gate,
                                                       int epoch, u64 *__restrict__ zeroHits, u64 *__restrict__ zeroTotal,
                                                       int32_t *__restrict__ bucketLong)
{
    {
        const int gi = blockIdx.x * SP_WG + threadIdx.x;
        if (zeroHits) for (int f = gi; f < db.nFiles; f += gridDim.x * SP_WG) zeroHits[f] = 0;   // IGD_HIP_FLAG_ZERO_FIRST
        if (zeroTotal && gi == 0) *zeroTotal = 0;
        if (gate == 0 && gi == 0) {
            ctl[CTL_NLONG + ((epoch + 1) & 1)] = 0;
            ctl[CTL_NFIX + ((epoch + 1) & 1)] = 0;
            ctl[CTL_BUDGET + ((epoch + 1) & 1)] = 0;
        ctl[CTL_NHEAVY + ((epoch + 1) & 1)] = 0;
        ctl[CTL_NHEAVYS + ((epoch + 1) & 1)] = 0;
        ctl[CTL_NFAR + ((epoch + 1) & 1)] = 0;
        }
    }
    if (gate != 0 && __builtin_amdgcn_readfirstlane(ctl[CTL_UNSORTED]) != gate) return;
    __shared__ uint32_t hist[SP_MAXC], cur[SP_MAXC], wsum[SP_WG / IGD_WAVE];
    for (int b = threadIdx.x; b < nCoarse; b += SP_WG) hist[b] = 0;
    __syncthreads();
    int gt0[SP_PER], ntl[SP_PER], s_[SP_PER], e_[SP_PER];
#pragma unroll
    for (int k = 0; k < SP_PER; k++) {
        const int i = blockIdx.x * SP_Q + k * SP_WG + threadIdx.x;
        ntl[k] = 0; gt0[k] = 0; s_[k] = 0; e_[k] = 0;
        if (i < nq) {
            const int s = qs[i], e = qe[i];
            int g, n;
            if (query_span(db, ichr[i], s, e, rule, g, n)) {
                const int kind = walk_kind(db, s, e, n, packed);
                if (kind >= 0) {
                    longList[atomicAdd(&ctl[CTL_NLONG + (epoch & 1)], 1)] = make_int2(i, kind);
                    if (kind == WALK_ALL) cover_tiles(db, ctl, 1, epoch, g + 1, g + n - 1);   // first and last tile by the walk, the rest covered
                } else {
                    int live = 0;                         // bit j: tile g+j is not empty
                    for (int j = 0; j < n; j++) live |= (db.tileCnt[g + j] > 0) << j;
                    gt0[k] = g; ntl[k] = live; s_[k] = s; e_[k] = e;
                    for (int j = 0; j < n; j++)
                        if ((live >> j) & 1) atomicAdd(&hist[(g + j) >> shift], 1u);
                }
            }
        }
    }
    __syncthreads();
    {   // exclusive prefix over the buckets: thread t owns buckets 4t .. 4t+3 (SP_MAXC = 4 * SP_WG)
        const int b0 = threadIdx.x * (SP_MAXC / SP_WG);
        uint32_t c[SP_MAXC / SP_WG], sum = 0;
#pragma unroll
        for (int k = 0; k < SP_MAXC / SP_WG; k++) { c[k] = b0 + k < nCoarse ? hist[b0 + k] : 0u; sum += c[k]; }
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        uint32_t x = sum;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = (uint32_t)__shfl_up((int)x, o);
            if (lane >= o) x += y;
        }
        if (lane == 63) wsum[w] = x;
        __syncthreads();
        uint32_t run = x - sum;
        for (int k = 0; k < w; k++) run += wsum[k];
#pragma unroll
        for (int k = 0; k < SP_MAXC / SP_WG; k++) {
            if (b0 + k < nCoarse) {
                table[(size_t)blockIdx.x * nCoarse + b0 + k] = run | (c[k] << 16);
                if (c[k] >= SP_LONG && bucketLong) { bucketLong[b0 + k] = epoch; ctl[CTL_PILED] = epoch; }   // a quarter of this workgroup's queries in ONE bucket: a piled-up batch
                cur[b0 + k] = run;
            }
            run += c[k];
        }
    }
    __syncthreads();
    const size_t rb = (size_t)blockIdx.x * SP_CAP;
#pragma unroll
    for (int k = 0; k < SP_PER; k++)
        for (int live = ntl[k], j = 0; live; live >>= 1, j++)
            if (live & 1) {
                const int t = gt0[k] + j;
                const uint32_t pos = atomicAdd(&cur[t >> shift], 1u);
                SpTuple tu; tu.t = t; tu.s = s_[k]; tu.e = e_[k];
                reg[rb + pos] = tu;
            }
}

#define SP_ROWS 4     // table rows a thread keeps in flight
__device__ __forceinline__ void split_fine_whole(int b, int nT, int shift, int nCoarse, int nWG, const uint32_t *__restrict__ table,
                                                      const SpTuple *__restrict__ reg,
                                                      int32_t *__restrict__ pairN, int32_t *__restrict__ pairPos,
                                                      int2 *__restrict__ pairs, const int32_t *__restrict__ ctl, int gate,
                                                      int32_t *__restrict__ ctlw, int epoch, int32_t *__restrict__ heavy)
{
    extern __shared__ uint32_t sp_lds[];
    const int F = 1 << shift;
    uint32_t *cnt = sp_lds, *start = sp_lds + F;
    __shared__ uint32_t wsum[SPF_WG / IGD_WAVE], baseSh;
    const int t0 = b << shift;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int f = threadIdx.x; f < F; f += SPF_WG) cnt[f] = 0;
    {   // first pair of this bucket = pairs of all earlier buckets = the sum over the table's rows of each row's own
        // exclusive prefix at this column (the low half of the entries this workgroup reads anyway): no kernel of column sums
        uint32_t x = 0;
        for (int w = threadIdx.x; w < nWG; w += SPF_WG) x += table[(size_t)w * nCoarse + b] & 0xFFFFu;
        for (int o = 32; o > 0; o >>= 1) x += (uint32_t)__shfl_down((int)x, o);
        if (lane == 0) wsum[wv] = x;
    }
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t t = 0; for (int k = 0; k < SPF_WG / IGD_WAVE; k++) t += wsum[k]; baseSh = t; }
    // Every (workgroup of k_split_local, this bucket) segment of pairs is walked: short ones by the thread that looked
    // them up, long ones (>= 64 pairs: queries that come in sorted runs put a workgroup's 4096 queries into one or two
    // buckets, and ONE thread walked them all -- 5 x the time of scattered queries) by the whole wave.
    auto walk = [&](auto fn) {
        for (int wb = 0; wb < nWG; wb += SPF_WG * SP_ROWS) {
            const int w0 = wb + (int)threadIdx.x;
            uint32_t e[SP_ROWS];
#pragma unroll
            for (int r = 0; r < SP_ROWS; r++) { const int w = w0 + r * SPF_WG; e[r] = w < nWG ? table[(size_t)w * nCoarse + b] : 0u; }
#pragma unroll
            for (int r = 0; r < SP_ROWS; r++) {
                const unsigned at = (unsigned)(w0 + r * SPF_WG) * (unsigned)SP_CAP + (e[r] & 0xFFFFu);   // (< 2^24 queries x 4 pairs: fits 32 bits)
                const int c = (int)(e[r] >> 16);
                const bool longSeg = c >= IGD_WAVE;
                if (!longSeg) for (int j = 0; j < c; j++) fn(reg[(size_t)at + j]);
                unsigned long long m = __ballot(longSeg);
                while (m) {
                    const int src = __builtin_ctzll(m);
                    m &= m - 1;
                    const unsigned at2 = (unsigned)__builtin_amdgcn_readlane((int)at, src);
                    const int c2 = __builtin_amdgcn_readlane(c, src);
                    // four tuples in flight per lane (one workgroup owns the bucket: 10^6 queries piled up in a few tiles are
                    // ALL its pairs, and a load waited for per tuple made that 2.6 ms)
                    int j = lane;
                    for (; j + 3 * IGD_WAVE < c2; j += 4 * IGD_WAVE) {
                        const SpTuple a0 = reg[(size_t)at2 + j], a1 = reg[(size_t)at2 + j + IGD_WAVE];
                        const SpTuple a2 = reg[(size_t)at2 + j + 2 * IGD_WAVE], a3 = reg[(size_t)at2 + j + 3 * IGD_WAVE];
                        fn(a0); fn(a1); fn(a2); fn(a3);
                    }
                    for (; j < c2; j += IGD_WAVE) fn(reg[(size_t)at2 + j]);
                }
            }
        }
    };
    walk([&](const SpTuple &tu) { atomicAdd(&cnt[tu.t - t0], 1u); });
    __syncthreads();
    {   // exclusive prefix over the bucket's tiles: thread t owns F/SPF_WG consecutive tiles
        const int per0 = F >= SPF_WG ? F / SPF_WG : 1, f0 = threadIdx.x * per0;
        const int per = f0 < F ? per0 : 0;               // (more threads than tiles: the rest own none)
        uint32_t sum = 0;
        for (int k = 0; k < per; k++) sum += cnt[f0 + k];
        uint32_t x = sum;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = (uint32_t)__shfl_up((int)x, o);
            if (lane >= o) x += y;
        }
        __syncthreads();
        if (lane == 63) wsum[wv] = x;
        __syncthreads();
        uint32_t run = baseSh + x - sum;
        for (int k = 0; k < wv; k++) run += wsum[k];
        for (int k = 0; k < per; k++) {
            const uint32_t c = cnt[f0 + k];
            start[f0 + k] = run;
            if (t0 + f0 + k < nT) {
                pairPos[t0 + f0 + k] = (int32_t)(run + c);
                int32_t pn = (int32_t)c;
                if (heavy && c > IGD_HEAVY_PAIRS) {       // too many pairs for one wave: heavy_bucket_body shares the tile out
                    const int at = atomicAdd(&ctlw[CTL_NHEAVY + (epoch & 1)], 1);
                    if (at < IGD_HEAVY_MAX) { heavy[at] = t0 + f0 + k; pn = -pn; }   // negative: "not yours" for igd_scan_tiles
                }
                pairN[t0 + f0 + k] = pn;
            }
            run += c;
        }
    }
    __syncthreads();
    walk([&](const SpTuple &tu) {
        const uint32_t pos = atomicAdd(&start[tu.t - t0], 1u);
        pairs[pos] = make_int2(tu.s, tu.e);
    });
}

__global__ __launch_bounds__(SPF_WG) void k_split_fine(int nT, int shift, int nCoarse, int nWG, const uint32_t *__restrict__ table,
                                                      const SpTuple *__restrict__ reg,
                                                      int32_t *__restrict__ pairN, int32_t *__restrict__ pairPos,
                                                      int2 *__restrict__ pairs, const int32_t *__restrict__ ctl, int gate,
                                                      int32_t *__restrict__ ctlw, int epoch, int32_t *__restrict__ heavy)
{
    if (gate != 0 && __builtin_amdgcn_readfirstlane(ctl[CTL_UNSORTED]) != gate) return;
    split_fine_whole((int)blockIdx.x, nT, shift, nCoarse, nWG, table, reg, pairN, pairPos, pairs, ctl, gate, ctlw, epoch, heavy);
}

// Several workgroups per coarse bucket for the buckets of a PILED-UP batch (round 4).  One workgroup per bucket walks ALL pairs
// of the bucket twice: an unordered batch piled up in a few tiles puts 10^6 pairs into one bucket, and that one workgroup
// streams 12 MB of tuples at what a single CU keeps in flight -- 1.7 ms where the whole batch otherwise takes 0.13.  A bucket
// in which k_split_local has seen a long segment (bucketLong[b] == epoch) is shared by SPF_S workgroups of SPF_W waves, each
// wave taking every (SPF_S * SPF_W)-th segment, and the step that needs all of them -- where a tile's pairs start -- sits
// between two kernels: _a counts (bucket, share, tile), _b adds the shares up, places its own and scatters.  Every other
// bucket is grouped by its first workgroup in _a exactly as before (split_fine_whole) and costs the other seven one load.
#ifndef SPF_S
#define SPF_S 8                // workgroups per piled-up coarse bucket ...
#endif
#define SPF_W (SPF_WG / IGD_WAVE)   // ... of this many waves each
template <typename FN>
__device__ __forceinline__ void split_walk_share(int nWG, int nCoarse, int b, int part, int lane, const uint32_t *__restrict__ table,
                                                 const SpTuple *__restrict__ reg, FN fn)
{
    constexpr int P = SPF_S * SPF_W;
    for (int wb = part; wb < nWG; wb += P * IGD_WAVE) {
        const int w = wb + lane * P;
        const uint32_t e = w < nWG ? table[(size_t)w * nCoarse + b] : 0u;
        const unsigned at = (unsigned)w * (unsigned)SP_CAP + (e & 0xFFFFu);
        const int c = (int)(e >> 16);
        const bool longSeg = c >= IGD_WAVE;
        if (!longSeg) for (int j = 0; j < c; j++) fn(reg[(size_t)at + j]);
        unsigned long long m = __ballot(longSeg);
        while (m) {
            const int src = __builtin_ctzll(m);
            m &= m - 1;
            const unsigned at2 = (unsigned)__builtin_amdgcn_readlane((int)at, src);
            const int c2 = __builtin_amdgcn_readlane(c, src);
            int j = lane;
            for (; j + 3 * IGD_WAVE < c2; j += 4 * IGD_WAVE) {
                const SpTuple a0 = reg[(size_t)at2 + j], a1 = reg[(size_t)at2 + j + IGD_WAVE];
                const SpTuple a2 = reg[(size_t)at2 + j + 2 * IGD_WAVE], a3 = reg[(size_t)at2 + j + 3 * IGD_WAVE];
                fn(a0); fn(a1); fn(a2); fn(a3);
            }
            for (; j < c2; j += IGD_WAVE) fn(reg[(size_t)at2 + j]);
        }
    }
}

__global__ __launch_bounds__(SPF_WG) void k_split_fine_a(int nT, int shift, int nCoarse, int nWG, const uint32_t *__restrict__ table,
                                                        const SpTuple *__restrict__ reg, uint32_t *__restrict__ sub,
                                                        uint32_t *__restrict__ bucketBase, const int32_t *__restrict__ bucketLong,
                                                        int32_t *__restrict__ pairN, int32_t *__restrict__ pairPos,
                                                        int2 *__restrict__ pairs, const int32_t *__restrict__ ctl, int gate,
                                                        int32_t *__restrict__ ctlw, int epoch, int32_t *__restrict__ heavy)
{
    if (gate != 0 && __builtin_amdgcn_readfirstlane(ctl[CTL_UNSORTED]) != gate) return;
    const int b = blockIdx.x % nCoarse, share = blockIdx.x / nCoarse;   // (workgroups go round-robin to the 8 XCDs: share = blockIdx & 7 put every first share -- all the work of an ordinary batch -- on ONE of them: 121 instead of 25 us)
    if (__builtin_amdgcn_readfirstlane(bucketLong[b]) != epoch) {      // the usual bucket: one workgroup, one kernel
        if (share == 0) split_fine_whole(b, nT, shift, nCoarse, nWG, table, reg, pairN, pairPos, pairs, ctl, gate, ctlw, epoch, heavy);
        return;
    }
    extern __shared__ uint32_t sp_lds[];
    __shared__ uint32_t wsumA[SPF_W];
    const int F = 1 << shift;
    uint32_t *cnt = sp_lds;
    const int t0 = b << shift, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int f = threadIdx.x; f < F; f += SPF_WG) cnt[f] = 0;
    if (share == 0) {           // pairs of all earlier buckets: the sum over the table's rows of each row's own exclusive prefix at this column
        uint32_t x = 0;
        for (int w = threadIdx.x; w < nWG; w += SPF_WG) x += table[(size_t)w * nCoarse + b] & 0xFFFFu;
        for (int o = 32; o > 0; o >>= 1) x += (uint32_t)__shfl_xor((int)x, o);
        if (lane == 0) wsumA[wv] = x;
    }
    __syncthreads();
    if (share == 0 && threadIdx.x == 0) { uint32_t t = 0; for (int k = 0; k < SPF_W; k++) t += wsumA[k]; bucketBase[b] = t; }
    split_walk_share(nWG, nCoarse, b, share * SPF_W + wv, lane, table, reg, [&](const SpTuple &tu) { atomicAdd(&cnt[tu.t - t0], 1u); });
    __syncthreads();
    uint32_t *mine = sub + (((size_t)b * SPF_S + share) << shift);
    for (int f = threadIdx.x; f < F; f += SPF_WG) mine[f] = cnt[f];
}

__global__ __launch_bounds__(SPF_WG) void k_split_fine_b(int nT, int shift, int nCoarse, int nWG, const uint32_t *__restrict__ table,
                                                        const SpTuple *__restrict__ reg, const uint32_t *__restrict__ sub,
                                                        const uint32_t *__restrict__ bucketBase, const int32_t *__restrict__ bucketLong,
                                                        int32_t *__restrict__ pairN, int32_t *__restrict__ pairPos,
                                                        int2 *__restrict__ pairs, const int32_t *__restrict__ ctl, int gate,
                                                        int32_t *__restrict__ ctlw, int epoch, int32_t *__restrict__ heavy)
{
    if (gate != 0 && __builtin_amdgcn_readfirstlane(ctl[CTL_UNSORTED]) != gate) return;
    if (__builtin_amdgcn_readfirstlane(ctl[CTL_PILED]) != epoch) return;  // no piled-up bucket in this batch (the usual case: an empty launch of a few hundred workgroups)
    extern __shared__ uint32_t sp_lds[];
    __shared__ uint32_t wsumB[SPF_W], carry;
    const int F = 1 << shift;
    uint32_t *start = sp_lds;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int item = blockIdx.x; item < nCoarse * SPF_S; item += gridDim.x) {
    const int b = item % nCoarse, share = item / nCoarse;
    if (__builtin_amdgcn_readfirstlane(bucketLong[b]) != epoch) continue;  // done by k_split_fine_a
    const int t0 = b << shift;
    __syncthreads();                                     // (the previous item's LDS is done with)
    if (threadIdx.x == 0) carry = bucketBase[b];
    __syncthreads();
    // per tile: all shares' pairs (where the next tile starts) and those of the shares before this one (where this one's go)
    for (int f0 = 0; f0 < F; f0 += SPF_WG) {
        const int f = f0 + (int)threadIdx.x;
        uint32_t all = 0, before = 0;
        if (f < F) {
            const uint32_t *col = sub + (((size_t)b * SPF_S) << shift) + f;
#pragma unroll
            for (int k = 0; k < SPF_S; k++) { const uint32_t c = col[(size_t)k << shift]; all += c; before += k < share ? c : 0u; }
        }
        uint32_t x = all;                                 // inclusive prefix over the round's tiles: inside the wave, then over the waves
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)x, o); if (lane >= o) x += y; }
        if (lane == 63) wsumB[wv] = x;
        __syncthreads();
        uint32_t first = carry + x - all;
        for (int k = 0; k < wv; k++) first += wsumB[k];
        if (f < F) {
            start[f] = first + before;
            if (share == 0 && t0 + f < nT) {
                pairPos[t0 + f] = (int32_t)(first + all);
                int32_t pn = (int32_t)all;
                if (heavy && all > IGD_HEAVY_PAIRS) {     // too many pairs for one wave: heavy_bucket_body shares the tile out
                    const int at = atomicAdd(&ctlw[CTL_NHEAVY + (epoch & 1)], 1);
                    if (at < IGD_HEAVY_MAX) { heavy[at] = t0 + f; pn = -pn; }
                }
                pairN[t0 + f] = pn;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) { uint32_t t = carry; for (int k = 0; k < SPF_W; k++) t += wsumB[k]; carry = t; }
        __syncthreads();
    }
    split_walk_share(nWG, nCoarse, b, share * SPF_W + wv, lane, table, reg, [&](const SpTuple &tu) {
        const uint32_t pos = atomicAdd(&start[tu.t - t0], 1u);
        pairs[pos] = make_int2(tu.s, tu.e);
    });
    }
}

// step 2: exclusive scan of pairCnt -> pairPos (two kernels, no inter-block protocol).  The
// apply kernel also moves the counts to pairN and leaves pairCnt zeroed for the next batch.
__global__ __launch_bounds__(IGD_SCAN_BLOCK) void k_scan_block_sums(const int32_t *__restrict__ in,
                                                                    int n, int32_t *__restrict__ blockSums,
                                                                    const int32_t *__restrict__ ctl, int gate)
{
    if (gate != 0 && ctl[CTL_UNSORTED] != gate) return;
    __shared__ int32_t red[IGD_SCAN_BLOCK / IGD_WAVE];
    int base = blockIdx.x * IGD_SCAN_TILE + threadIdx.x * IGD_SCAN_ITEMS;
    int s = 0;
#pragma unroll
    for (int k = 0; k < IGD_SCAN_ITEMS; k++)
        if (base + k < n) s += in[base + k];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < IGD_SCAN_BLOCK / IGD_WAVE; w++) t += red[w];
        blockSums[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(IGD_SCAN_BLOCK) void k_scan_apply(int32_t *__restrict__ in, int n,
                                                               const int32_t *__restrict__ blockSums,
                                                               int32_t *__restrict__ out,
                                                               int32_t *__restrict__ copy,
                                                               const int32_t *__restrict__ ctl, int gate)
{
    if (gate != 0 && ctl[CTL_UNSORTED] != gate) return;
    __shared__ int32_t red[IGD_SCAN_BLOCK / IGD_WAVE];
    __shared__ int32_t wsum[IGD_SCAN_BLOCK / IGD_WAVE];
    // prefix of the earlier blocks' sums (every block recomputes it; a few hundred values)
    int pre = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += IGD_SCAN_BLOCK) pre += blockSums[b];
    for (int o = 32; o > 0; o >>= 1) pre += __shfl_down(pre, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = pre;
    // local items
    int base = blockIdx.x * IGD_SCAN_TILE + threadIdx.x * IGD_SCAN_ITEMS;
    int v[IGD_SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int k = 0; k < IGD_SCAN_ITEMS; k++) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        s += v[k];
    }
    // inclusive scan of s across the wave
    int inc = s;
    int lane = threadIdx.x & 63;
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    int blockPre = 0;
    for (int w = 0; w < IGD_SCAN_BLOCK / IGD_WAVE; w++) blockPre += red[w];
    int wavePre = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) wavePre += wsum[w];
    int run = blockPre + wavePre + inc - s;
#pragma unroll
    for (int k = 0; k < IGD_SCAN_ITEMS; k++) {
        if (base + k < n) {
            out[base + k] = run;
            if (copy) { copy[base + k] = v[k]; in[base + k] = 0; }
        }
        run += v[k];
    }
}

// step 3: scatter (qs,qe) of every pair to its tile's slot range.  After this kernel
// pairPos[t] is the END of tile t's range.
__global__ void k_scatter_pairs(DbView db, const int32_t *__restrict__ ichr,
                                const int32_t *__restrict__ qs, const int32_t *__restrict__ qe,
                                int nq, int rule, int packed, int32_t *__restrict__ pairPos,
                                void *__restrict__ pairs, const int32_t *__restrict__ ctl, int gate)
{
    if (gate != 0 && ctl[CTL_UNSORTED] != gate) return;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    int gt0, ntl;
    int s = qs[i], e = qe[i];
    if (!query_span(db, ichr[i], s, e, rule, gt0, ntl)) return;
    if (walk_kind(db, s, e, ntl, packed) >= 0) return;
    for (int k = 0; k < ntl; k++) {
        if (db.tileCnt[gt0 + k] > 0) {
            int p = atomicAdd(&pairPos[gt0 + k], 1);
            ((int2 *)pairs)[p] = make_int2(s, e);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Compact image.  Everything the COUNT of a (query, tile) pair depends on is relative to the
// tile: with T = tile start, W = tile width, a record of tile j is reduced to
//     s' = start < T ? 0 : start - T + 1        in [0, W]    (0: "starts before this tile")
//     e' = min(end - T, W)                       in [1, W]    (W: "reaches the tile's end")
// and a query visiting the tile to
//     qe' = min(qe - T, W) + 1,   qs' = first ? max(qs - T + 1, 1) : 1,   lob' = first ? 0 : 1
// so that   lob' <= s' < qe'  &&  e' >= qs'   <=>   lob <= start < qe  &&  end > qs
// for every query with qe > T (the conditions of SURVEY App. B.3; proof in DESIGN.md).
// Stored word:  (65535 - s') | e' << 16.  With the query word  (65536 - qe') | qs' << 16  the
// test  s' < qe' && e' >= qs'  is "both 16-bit halves >= the query's halves": one v_pk_max_u16
// and one compare.  The remaining condition s' >= lob' only excludes records that start before
// the tile (s' = 0) from queries for which this is not the first tile; since EVERY such query
// matches EVERY such record on the other two conditions, it is applied once per unit as a
// correction (hits -= number of non-first queries) instead of once per record and query.  A
// first-tile query with qe <= T (an inverted query reaching back over the tile start) is the one
// case that needs the exact starts: the grouping kernels list it for k_exact_walk (WALK_FIRST).
// 6 bytes per record (4 + 2; 8 with the 16-bit value) instead of 12 (16).
// Slot summaries: the scan kernel reads a unit as IGD_SLOTS slots of 64 consecutive records.  The
// component-wise maximum of a slot's words, (65535 - min s') | max e' << 16, passes the query
// test exactly when SOME word in the slot COULD pass it, so a query whose word fails against the
// summary skips the slot.  The low half is read from the slot's first record (the tile is sorted by
// start); the whole summary word of every slot is kept in the unit descriptor (Unit::W), so the scan
// kernel can prune a unit's queries before -- or without -- waiting for the unit's records.
__global__ __launch_bounds__(256) void k_pack_units(DbView db, Unit *__restrict__ unitsOut, uint32_t *__restrict__ pse,
                                                    uint16_t *__restrict__ px, uint32_t *__restrict__ pv,
                                                    int32_t *__restrict__ flag /* bit 0: a value needs > 16 bits; bit 1: malformed tile */)
{
    const int lane = threadIdx.x & 63;
    const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nw = (gridDim.x * blockDim.x) >> 6;
    int wide = 0;
    for (int ui = gw; ui < db.nUnits; ui += nw) {
        const Unit u = unitsOut[ui];
        const int T = (int)((unsigned)UNIT_J(u) * (unsigned)db.nbp);
        unsigned mx[6] = {0u, 0u, 0u, 0u, 0u, 0u};        // summary words of the unit's slots
        int npre = 0;
        for (int i0 = 0; i0 < u.n; i0 += IGD_WAVE) {
            const int i = i0 + lane;
            unsigned edv = 0u, spv = 65535u;
            if (i < u.n) {
            const int64_t r = u.off + i;
            const int st = db.start[r], en = db.end[r];
            const unsigned sp = st < T ? 0u : (unsigned)(st - T) + 1u;
            long long ed = (long long)en - T;
            if (ed > db.nbp) ed = db.nbp;
            if (ed < 1) ed = 1;
            edv = (unsigned)ed;
            spv = sp;
            pse[r] = (65535u - sp) | ((unsigned)ed << 16);
            px[r] = (uint16_t)db.idx[r];
            if (pv) {
                const int v = db.value[r];
                wide |= (v < -32768) | (v > 32767);
                pv[r] = (uint32_t)(uint16_t)db.idx[r] | ((uint32_t)(uint16_t)(int16_t)v << 16);
            }
            // a record that does not belong to its tile (malformed file): keep the exact path
            if (!(st < T + db.nbp && en > T)) wide |= 2;
            }
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned y = (unsigned)__shfl_xor((int)edv, o);
                edv = y > edv ? y : edv;
            }
            npre += __popcll(__ballot(spv == 0u));
            const unsigned s0 = (unsigned)__shfl((int)spv, 0);      // the tile is sorted by start: the slot's first record has its smallest s'
            if (i0 / IGD_WAVE < 6) mx[i0 / IGD_WAVE] = (65535u - s0) | (edv << 16);
        }
        if (lane == 0) {
            for (int r = 0; r < 6; r++) unitsOut[ui].W[r] = mx[r];
            unitsOut[ui].pre = npre;
        }
    }
    if (wide) atomicOr(flag, wide);
}

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

// Compact image: the queries of `live` (one per lane, word P0) against the unit, slot by slot.  A
// query is compared with the records of a slot only if its word passes against the slot's summary
// W[r] -- the same packed test, done for 64 queries at once; on the benchmark that leaves 1.8 of
// 4.9 slots per (query, unit).  cnt[r] += hit; no exec masking, no LDS.
__device__ __forceinline__ void match_slots(const Raw &R, int (&cnt)[IGD_SLOTS], const uint32_t (&W)[IGD_SLOTS], int P0,
                                            unsigned long long live)
{
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
    u64 *stamps;                 // IGD_EXP & 32 (diagnostic build): 4 s_memtime stamps per wave
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
template <bool USE_V, bool BIG>
__device__ __forceinline__ void s_issue(const DbView &db, const SortArgs &a, const SRegs &L, int kk, bool valid, int lane, Raw2 &R)
{
    const int kq = kk & 63;
    int c0 = __builtin_amdgcn_readlane(L.c0, kq), ln = __builtin_amdgcn_readlane(L.ln, kq);
    if (!valid) { c0 = 0; ln = 0; }
    const int f0 = __builtin_amdgcn_readlane(L.f0, kq);
    const int n = (c0 | ln) ? __builtin_amdgcn_readlane(L.n, kq) : 0;
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
    const int nA = IGD_LN_FAR(ln) ? 0 : IGD_LN_A(ln), nB = IGD_LN_FAR(ln) ? 0 : IGD_LN_B(ln);
    const int nl = nA + nB;
    const int b0 = c0 < IGD_WAVE - nl ? c0 : IGD_WAVE - nl;
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void *)a.qw0, 0, (f0 + b0) * 4, 0x00020000);
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
__device__ __forceinline__ void match_words(const Raw2 &R, int (&cnt)[IGD_SLOTS], const uint32_t (&W)[IGD_SLOTS], int P0)
{
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
                                          u64 *found = nullptr, unsigned *spent = nullptr, unsigned budget = 0u)
{
    const int c0 = R.c0, ln = R.ln;
    if ((c0 | ln) == 0) return;                          // nobody asks about this unit
#if IGD_EXP & 1024
    const u64 t_unit = __builtin_amdgcn_s_memtime();
#endif
    const int un = R.n;
    if (un == 0) return;                                 // placeholder of an empty tile
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
    const int nl = (IGD_LN_FAR(ln) || (IGD_EXP & 8) != 0) ? 0 : IGD_LN_A(ln) + IGD_LN_B(ln);
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
        for (int p = 0; p < nE; p += IGD_WAVE) {
            // the next 64 words are on their way while these are compared (a dense tile is a chain of such batches)
            const int wn = (p + IGD_WAVE + lane < nE) ? ~a.qw0[f0 + p + IGD_WAVE + lane - nl] : (int)IGD_NEVER;
            match_words(R, cnt, W, w);
            w = wn;
        }
        if (far)
            far_later<RANK>(a, __builtin_amdgcn_readlane(L.la, kk), ln, f0, lane, [&](int e) {
                bool covers;
                const int lw = later_word(db.nbp, e, g2, deadk, true, covers);
                nLater += __popcll(__ballot(covers));
                match_words(R, cnt, W, lw);
            });
        // records that start before the tile (s' = 0, low half 65535) were matched by every "later tile"
        // query, none of which may count them (the reference's tS skip, :510-511)
        if (nLater != 0) {
#pragma unroll
            for (int r = 0; r < IGD_SLOTS; r++) cnt[r] -= (R.a[r] & 0xFFFFu) == 0xFFFFu ? nLater : 0;
        }
    } else {
        // ---- rank ---------------------------------------------------------------------------------
#if IGD_EXP & 1024
        u64 tsec = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        SECT(5);                                         // waiting for the unit's records
#endif
#define IGD_TILE_START ((int)((unsigned)(__builtin_amdgcn_readlane(L.jf, kk) >> 4) * (unsigned)db.nbp))   /* only the seldom-taken branches need it */
#pragma unroll
        for (int r = 0; r < IGD_SLOTS; r++)
            sl[r * IGD_WAVE + lane] = (unsigned short)(65535u - (R.a[r] & 0xFFFFu));   // lanes past the unit: 65535
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const bool inLds = c0 < a.sbCap;           // the tile's query starts fit the wave's LDS array (a power of two)
        int nFirst = 0;
        SECT(0);
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
        SECT(1);
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
        SECT(2);
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
        SECT(3);
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
        SECT(4);
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
            if ((!RANK && !few) || cnt[r]) atomicAdd((unsigned int *)hits + R.x[r], (unsigned)cnt[r]);
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
#if IGD_EXP & 1024
    if (lane == 0) atomicAdd(&hist[321 + 6], (unsigned)(__builtin_amdgcn_s_memtime() - t_unit));   // all of the unit's compare phase
#endif
    if (found) {                                         // skew valve: the batch total is kept by the caller of this unit
        int t = 0;
#pragma unroll
        for (int r = 0; r < IGD_SLOTS; r++) t += cnt[r];
        for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o);
        if (lane == 0 && t) atomicAdd(found, (u64)(unsigned)t);
    }
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
#ifndef IGD_LEAN_DEPTH
#define IGD_LEAN_DEPTH 2        // units in flight per wave in the lean build
#endif
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
#if IGD_EXP & 1024
    const u64 t_kernel = __builtin_amdgcn_s_memtime();
#endif
#if IGD_EXP & 32
    const u64 t_start = __builtin_amdgcn_s_memtime();
    u64 t_desc = 0, t_first = 0;
#endif
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

    for (int ub = gwave; ub < db.nUnits; ub += nwaves * IGD_WAVE) {
        SRegs L;
        L.offLo = L.offHi = L.n = L.jf = L.f0 = L.c0 = L.la = L.ln = 0;
#pragma unroll
        for (int r = 0; r < IGD_SLOTS; r++) L.w[r] = 0;
        {
            const long long mi = (long long)ub + (long long)lane * nwaves;
            if (mi < db.nUnits) {
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
                }
            }
        }
        int cntU = (int)(((long long)db.nUnits - ub + nwaves - 1) / nwaves);
        if (cntU > IGD_WAVE) cntU = IGD_WAVE;
#if IGD_EXP & 32
        if (ub == gwave) { asm volatile("s_waitcnt vmcnt(0)" ::"v"(L.f0), "v"(L.c0), "v"(L.ln)); t_desc = __builtin_amdgcn_s_memtime(); }
#endif
        if (RANK) {
            // The full build also serves batches that visit a fraction of the units (one GPU's slab of config 4: one unit
            // in eight): the wave steps through the units somebody asks about only -- an unvisited one still cost its
            // dozen zero-size loads, which queue up behind everybody's real ones.
            unsigned long long m = __ballot((L.c0 | L.ln) != 0 && L.n > 0);
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
#if IGD_LEAN_DEPTH == 3
        // three units in flight per wave (the lean build at 6 waves per SIMD has the registers for a third buffer):
        // the kernel is bound by memory latency x units in flight, 24 x 3 per CU instead of 32 x 2
        {
            Raw2 C;
            s_issue<USE_V, BIG>(db, a, L, 0, true, lane, A);
            s_issue<USE_V, BIG>(db, a, L, 1, 1 < cntU, lane, B);
            for (int kk = 0; kk < cntU; kk += 3) {
                s_issue<USE_V, BIG>(db, a, L, kk + 2, kk + 2 < cntU, lane, C);
                s_compute<USE_V, CNT32, RANK, LDS_HITS, FEW>(db, a, L, kk, lane, A, hits, sl, hist, sb, rankOK, nullptr, &spent, budget);
                s_issue<USE_V, BIG>(db, a, L, kk + 3, kk + 3 < cntU, lane, A);
                if (kk + 1 < cntU) s_compute<USE_V, CNT32, RANK, LDS_HITS, FEW>(db, a, L, kk + 1, lane, B, hits, sl, hist, sb, rankOK, nullptr, &spent, budget);
                s_issue<USE_V, BIG>(db, a, L, kk + 4, kk + 4 < cntU, lane, B);
                if (kk + 2 < cntU) s_compute<USE_V, CNT32, RANK, LDS_HITS, FEW>(db, a, L, kk + 2, lane, C, hits, sl, hist, sb, rankOK, nullptr, &spent, budget);
#if IGD_OPT_PRIO
                done += 3;
                if (done >= prioAt) {
                    prioAt += quarter;
                    prioLevel--;
                    if (prioLevel == 2) __builtin_amdgcn_s_setprio(2);
                    else if (prioLevel == 1) __builtin_amdgcn_s_setprio(1);
                    else __builtin_amdgcn_s_setprio(0);
                }
#endif
            }
            continue;
        }
#endif
        s_issue<USE_V, BIG>(db, a, L, 0, true, lane, A);
        for (int kk = 0; kk < cntU; kk += 2) {
            s_issue<USE_V, BIG>(db, a, L, kk + 1, kk + 1 < cntU, lane, B);
#if IGD_EXP & 32
            if (ub == gwave && kk == 0) { asm volatile("s_waitcnt vmcnt(0)" ::"v"(A.a[0]), "v"(A.x[0])); t_first = __builtin_amdgcn_s_memtime(); }
#endif
            s_compute<USE_V, CNT32, RANK, LDS_HITS, FEW>(db, a, L, kk, lane, A, hits, sl, hist, sb, rankOK, nullptr, &spent, budget);
            s_issue<USE_V, BIG>(db, a, L, kk + 2, kk + 2 < cntU, lane, A);
            if (kk + 1 < cntU) s_compute<USE_V, CNT32, RANK, LDS_HITS, FEW>(db, a, L, kk + 1, lane, B, hits, sl, hist, sb, rankOK, nullptr, &spent, budget);
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
#if IGD_EXP & 32
    const u64 t_loop = __builtin_amdgcn_s_memtime();
#endif
#if IGD_EXP & 1024
    if (RANK && lane < 5) atomicAdd(&d_sect[lane], (u64)hist[321 + lane]);
    if (RANK && lane == 6) atomicAdd(&d_sect[6], (u64)hist[321 + 5]);
    if (RANK && lane == 7) atomicAdd(&d_sect[7], (u64)hist[321 + 6]);
    if (RANK && lane == 5) atomicAdd(&d_sect[5], __builtin_amdgcn_s_memtime() - t_kernel);
#endif
    if (LDS_HITS) {
        __syncthreads();
        const int nFiles = KARG(db.nFiles);
        u64 *row = KARG(a.out) + (size_t)blockIdx.x * nFiles;
        if (CNT32) {                                     // 32-bit rows: half the bytes written here and read back by k_reduce_slabs
            unsigned int *row32 = (unsigned int *)KARG(a.out) + (size_t)blockIdx.x * nFiles;
            for (int f = threadIdx.x; f < nFiles; f += WGT) row32[f] = ((unsigned int *)hits)[f];
        } else for (int f = threadIdx.x; f < nFiles; f += WGT) row[f] = hits[f];
    }
#if IGD_EXP & 32
    if (KARG(a.stamps) && lane == 0) {
        u64 *o = KARG(a.stamps) + (size_t)gwave * 5;
        o[0] = t_start; o[1] = t_desc; o[2] = t_first; o[3] = t_loop; o[4] = __builtin_amdgcn_s_memtime();
    }
#endif
}

// heavy_sorted_body: the merge join's skew valve.  A tile with more than IGD_HEAVY_FIRST (lean build: IGD_LEAN_FIRST)
// first-tile queries -- 10^6 ordered queries inside ONE tile would keep one wave busy for 9 ms -- is listed by
// igd_scan_sorted and left out there; here every (unit of the tile, slice of IGD_HEAVY_SLICE queries) is one work item,
// dealt round-robin to all waves of the hosting launch (the batch's last kernel) -- the rank method is a sum over
// queries, so slices simply add up -- and added to hits[] and the batch total with global atomics.  `wsm`: this wave's
// LDS area for the rank method.  Must sit in a kernel whose FIRST argument is the batch's SortK (KARG).
template <bool USE_V, bool BIG>
__device__ __forceinline__ void heavy_sorted_body(const SortK &K, u64 *__restrict__ d_hits, u64 *__restrict__ d_total,
                                                  unsigned char *wsm, int gwave, int nwaves, int lane, int ctlv)
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
            s_compute<USE_V, false, true>(db, a, L, 0, lane, A, d_hits, sl, hist, sb, rankOK, d_total);
        });
}

// far_units_body: the units the lean build of igd_scan_sorted listed (far: more later-tile candidates than come with a
// unit's records -- tiles behind very dense ones; that build has neither the walk over the blocks nor 64-bit counters in
// its registers).  One wave per listed unit: its later-tile candidates (far_later) and, unless its tile went to
// heavy_sorted_body, its first-tile queries, added to hits[] and the batch total with global atomics.
template <bool USE_V, bool BIG>
__device__ __forceinline__ void far_units_body(const SortK &K, u64 *__restrict__ d_hits, u64 *__restrict__ d_total,
                                               unsigned char *wsm, int gwave, int nwaves, int lane, int ctlv)
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
        s_compute<USE_V, false, true>(db, a, L, 0, lane, A, d_hits, sl, hist, sb, rankOK, d_total);
    }
}

// ------------------------------------------------------------------------------------------
// exact_walk_body: what the scan kernel leaves out (the exact-walk list of the batch's path): a wave walks a listed
// query's tiles (WALK_*: which of them) on the EXACT arrays, 5 slots at a time, and adds into the workgroup's LDS
// counters of the batch's last launch -- the caller's global hits[] where the files do not fit -- and the batch total.
// Rare for the benchmark's queries; a batch of long ones lists every query (its last tile).
template <bool USE_V>
__device__ __forceinline__ void exact_walk_body(const DbView &db, const ScanArgs &a, const int2 *__restrict__ fixList,
                                                const int2 *__restrict__ longList, int gwave, int nwaves, int ctlv, u64 *hist)
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
                const int jf = vkind == WALK_LAST ? vj1 : vn1;
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
                w.live = 0;
#pragma unroll
                for (int r = 0; r < IGD_SLOTS; r++) {
                    const int i = r * IGD_WAVE + lane;
                    const bool on = r * IGD_WAVE < n && (int)(65535u - ((unsigned)__builtin_amdgcn_readlane(vur.w[r], e) & 0xFFFFu)) < w.qe2;   // wave-uniform
                    w.pa[r] = 0xFFFFu; w.px[r] = 0u;     // (s' = 0: counts nothing)
                    if (on) {
                        w.live |= 1 << r;
                        if (i < n) { w.pa[r] = db.pse[off + i]; w.px[r] = USE_V ? db.pxv[off + i] : (uint32_t)db.px[off + i]; }
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
                    if (hit) { if (hist) atomicAdd(&hist[ix], 1ull); else atomicAdd(&a.out[ix], 1ull); }
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
                            if (i < n) { C.pa[r] = db.pse[off + i]; C.px[r] = USE_V ? db.pxv[off + i] : (uint32_t)db.px[off + i]; }
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
                        st[r] = ok ? db.start[toff + i] : INT_MAX;
                        en[r] = ok ? db.end[toff + i] : INT_MIN;
                        ix[r] = ok ? db.idx[toff + i] : 0;
                        if (USE_V) va[r] = ok ? db.value[toff + i] : INT_MIN;
                    }
                    if (__builtin_amdgcn_readfirstlane(st[0]) >= qe) break;    // sorted: nothing further
#pragma unroll
                    for (int r = 0; r < IGD_SLOTS; r++) {
                        bool hit = (st[r] < qe) & (st[r] >= lob) & (en[r] > qs);
                        if (USE_V) hit = hit & (va[r] >= a.v);
                        found += __popcll(__ballot(hit));
                        if (hit) { if (hist) atomicAdd(&hist[ix[r]], 1ull); else atomicAdd(&a.out[ix[r]], 1ull); }
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

#define IGD_COV_CHUNK 16      // units a wave takes at a time (32 left a quarter of the last launch's 8192 waves without a chunk of the benchmark's 190 000 units) (strided over the launch's waves: long queries may all lie in one region)
template <bool USE_V>
__device__ __forceinline__ void coverage_body(const DbView &db, const ScanArgs &a, u64 *__restrict__ d_hits,
                                              u64 *__restrict__ d_total, int gwave, int nwaves, int ctlv, u64 *hist)
{
    const int lane = threadIdx.x & 63;
    const bool uns = __builtin_amdgcn_readlane(ctlv, CTL_UNSORTED) == a.epoch;
    if (a.mode == 1 && uns) return;                      // broken promise: the batch adds nothing
    const int set = (a.mode == 1 || (a.mode == 0 && !uns)) ? 0 : 1, par = a.epoch & 1;   // merge join / bucket path
    if (__builtin_amdgcn_readlane(ctlv, CTL_COV + set * 2 + par) != a.epoch) return;     // no long query in this batch
    const int32_t *diff = db.cov + (size_t)(set * 2 + par) * IGD_COV_LEN(db.nT), *coarse = diff + db.nT + 2;
    u64 found = 0;
    for (int u0 = gwave * IGD_COV_CHUNK; u0 < db.nUnits; u0 += nwaves * IGD_COV_CHUNK) {
        const int cnt = db.nUnits - u0 < IGD_COV_CHUNK ? db.nUnits - u0 : IGD_COV_CHUNK;
        // one unit per lane: its tile, and the number of long queries that cover that tile from end to end = (coarse +
        // fine prefix at the chunk's first tile) + the differences of the tiles since -- no chain of loads from unit to unit
        UnitRegs ur = load_unit_regs(db.units + u0 + (lane < cnt ? lane : 0));
        const int tile = ur.tile, tile0 = __builtin_amdgcn_readfirstlane(tile);
        const int bd = db.tileBd[tile];                  // (a covered tile is never the first of its contig)
        int p0;
        {
            const int blk = tile0 >> IGD_COV_SHIFT;
            int sum = 0;
            for (int c = lane; c < blk; c += IGD_WAVE) sum += coarse[c];
            for (int t = (blk << IGD_COV_SHIFT) + lane; t <= tile0; t += IGD_WAVE) sum += diff[t];
            for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
            p0 = sum;
        }
        const int before = __shfl_up(tile, 1);
        const int gap = (lane == 0 || lane >= cnt) ? 0 : tile - before;   // tiles since the unit before (0: same tile; > 1: empty tiles between)
        int d = 0;
        if (gap <= 4) {
#pragma unroll
            for (int k = 0; k < 4; k++) d += k < gap ? diff[tile - k] : 0;
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
        struct Cov { int st[IGD_SLOTS], ix[IGD_SLOTS], va[USE_V ? IGD_SLOTS : 1]; int c, lob; };
        // (compact image: the records that start in the tile are the unit's records from number `pre` on -- Unit::pre -- so
        // only their dataset numbers are read: 2 bytes a record, 4 with the value, where the exact arrays cost 8 and 12)
        const bool cimg = USE_V ? a.packedWalk == 2 : a.packedWalk != 0;
        auto issue = [&](int e, Cov &w) {
            w.c = __builtin_amdgcn_readlane(cv, e);
            w.lob = __builtin_amdgcn_readlane(bd, e);
            const int n = __builtin_amdgcn_readlane(ur.n, e);
            const int64_t off = (int64_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(ur.offHi, e) << 32) |
                                          (unsigned)__builtin_amdgcn_readlane(ur.offLo, e));
            if (cimg) {
                const int pre = __builtin_amdgcn_readlane(ur.pre, e);
#pragma unroll
                for (int r = 0; r < IGD_SLOTS; r++) {
                    const int i = r * IGD_WAVE + lane;
                    const bool in = i >= pre && i < n;
                    w.st[r] = in ? INT_MAX : INT_MIN;            // (>= lob / < lob)
                    w.ix[r] = 0;
                    if (USE_V) w.va[r] = INT_MIN;
                    if (in && (r + 1) * IGD_WAVE > pre && r * IGD_WAVE < n) {
                        if (USE_V) { const uint32_t x = db.pxv[off + i]; w.ix[r] = (int)(x & 0xFFFFu); w.va[r] = (int)x >> 16; }
                        else w.ix[r] = (int)db.px[off + i];
                    }
                }
                return;
            }
#pragma unroll
            for (int r = 0; r < IGD_SLOTS; r++) {
                const int i = r * IGD_WAVE + lane;
                w.st[r] = i < n ? db.start[off + i] : INT_MIN;
                w.ix[r] = i < n ? db.idx[off + i] : 0;
                if (USE_V) w.va[r] = i < n ? db.value[off + i] : INT_MIN;
            }
        };
        auto count = [&](const Cov &w) {
#pragma unroll
            for (int r = 0; r < IGD_SLOTS; r++) {
                bool in = w.st[r] >= w.lob;              // the copy of the record that counts (:510-511)
                if (USE_V) in = in && w.va[r] >= a.v;
                found += (u64)__popcll(__ballot(in)) * (u64)(unsigned)w.c;
                if (in) { if (hist) atomicAdd(&hist[w.ix[r]], (u64)(unsigned)w.c); else atomicAdd(&d_hits[w.ix[r]], (u64)(unsigned)w.c); }
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
    u64 *hist = nullptr;
    if (K.a.tailHistOff >= 0) {
        const bool uns = __builtin_amdgcn_readlane(ctlv, CTL_UNSORTED) == wa.epoch;
        const bool sortedPath = wa.mode == 1 || (wa.mode == 0 && !uns);
        const int nList = __builtin_amdgcn_readlane(ctlv, (sortedPath ? CTL_NFIX : CTL_NLONG) + (wa.epoch & 1));
        const bool cov = __builtin_amdgcn_readlane(ctlv, CTL_COV + (sortedPath ? 0 : 2) + (wa.epoch & 1)) == wa.epoch;
        if (!(wa.mode == 1 && uns) && (nList > 0 || cov)) {          // (the same answer in every wave of the launch)
            hist = (u64 *)(smem + K.a.tailHistOff);
            for (int f = threadIdx.x; f < K.db.nFiles; f += blockDim.x) hist[f] = 0;
            __syncthreads();
        }
    }
    // (Tried: the long queries' work on a quarter of the launch's workgroups, to quarter the 3.9 x 10^6 atomics with which 2048
    // workgroups flush 1900 LDS counters each -- slower, 560 -> 874 us for 10^6 queries of 100-200 kbp: the walks want the waves.)
    exact_walk_body<USE_V>(K.db, wa, fixList, longList, gwave, nwaves, ctlv, hist);
    coverage_body<USE_V>(K.db, wa, d_hits, d_total, gwave, nwaves, ctlv, hist);
    if (hist) {
        __syncthreads();
        for (int f = threadIdx.x; f < K.db.nFiles; f += blockDim.x) {
            const u64 c = hist[f];
            if (c) atomicAdd(&d_hits[f], c);
        }
    }
    if (valves & 1) heavy_bucket_body<USE_V>(K.db, wa, heavyB, d_hits, d_total, gwave, nwaves, lane, ctlv);
    if (valves & 2) {
        unsigned char *wsm = smem + (size_t)(threadIdx.x >> 6) * (size_t)K.a.wldsBytes;
        if (valves & 4) heavy_sorted_body<USE_V, true>(K, d_hits, d_total, wsm, gwave, nwaves, lane, ctlv);
        else {
            if (!(IGD_EXP & 0x10000)) heavy_sorted_body<USE_V, false>(K, d_hits, d_total, wsm, gwave, nwaves, lane, ctlv);
            if (!(IGD_EXP & 0x20000)) far_units_body<USE_V, false>(K, d_hits, d_total, wsm, gwave, nwaves, lane, ctlv);   // (the lean build does not exist for BIG images)
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
template <bool USE_V>
__global__ __launch_bounds__(IGD_TAIL_WG) void k_reduce_slabs(SortK K, const u64 *__restrict__ slab, int rows, int nFiles,
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

// ------------------------------------------------------------------------------------------
// `-f` enumeration (get_overlaps_f1/_f0, src/igd_search.c:537-620,114-200).  A wave walks the tiles
// of its query in ascending order and each tile from its LAST 64 records to the first, so that
// record indices come out descending, as the reverse scans at :575-579 and :608-612 emit them.
//   pass COUNT: qcount[q] = overlaps of query q
//   scan      : qoff = exclusive scan of qcount
//   pass FILL : the rank-th overlap of query q goes to out[qoff[q] + rank]
// SEQ (Seqpare, src/igd_search.c:253-352): the record's place is taken by what seq_overlaps stores for it:
// start <- idx_g (index of the record inside its tile), end <- the bits of the float similarity
// sm = st / (qlen + rlen - st), computed in single precision in the reference's order of operations.
__device__ __forceinline__ int seq_similarity_bits(int qs, int qe, int s, int e)
{
    const float qlen = (float)(qe - qs);
    const float st = (float)((qe < e ? qe : e) - (qs > s ? qs : s));
    const float rlen = (float)(e - s);
    return __float_as_int(__fdiv_rn(st, __fsub_rn(__fadd_rn(qlen, rlen), st)));
}

// Query-major: one wave per query, so that a contiguous range of queries is a contiguous range of the
// output -- which is what lets the host side stream the result out in chunks while later chunks are
// still being produced (the path is bound by the 16 bytes per overlap that cross PCIe, not by these
// kernels).  No grouping step at all: with queries in any order a tile's records are simply re-read from
// L2 / HBM (2.2 KB per (query, tile) pair; < 1 ms per 10^6 queries either way).
//   COUNT (FILL = false): qcount[q] = overlaps of query q            (all its tiles)
//   FILL                : out[qoff[q] - base0 + rank] = the overlaps of queries [qa, qb), reference order
template <bool FILL, bool SEQ = false>
__global__ __launch_bounds__(256) void igd_enum_queries(
    DbView db, const int32_t *__restrict__ q_ichr, const int32_t *__restrict__ q_qs,
    const int32_t *__restrict__ q_qe, int qa, int qb, int64_t *__restrict__ qcount,
    const int64_t *__restrict__ qoff, int64_t base0, igd_hip_hit *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int gwave = blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    const u64 above = (lane == 63) ? 0ull : (~0ull << (lane + 1));   // lanes with a higher record index

    for (int q = qa + gwave; q < qb; q += nwaves) {
        const int qs = __builtin_amdgcn_readfirstlane(q_qs[q]);
        const int qe = __builtin_amdgcn_readfirstlane(q_qe[q]);
        const int cc = __builtin_amdgcn_readfirstlane(q_ichr[q]);
        int64_t cnt = 0;
        int gt0, ntl;
        if (query_span(db, cc, qs, qe, IGD_HIP_RULE_NEST, gt0, ntl)) {
            gt0 = __builtin_amdgcn_readfirstlane(gt0);
            ntl = __builtin_amdgcn_readfirstlane(ntl);
            const int64_t base = FILL ? qoff[q] - base0 : 0;
            for (int k = 0; k < ntl; k++) {
                const int t = gt0 + k;
                const int tcnt = __builtin_amdgcn_readfirstlane(db.tileCnt[t]);
                if (tcnt == 0) continue;
                const int lob = (k == 0) ? INT_MIN : __builtin_amdgcn_readfirstlane(db.tileBd[t]);
                const int64_t toff = db.tileOff[t];
                // records from the end of the tile towards the front, 64 at a time
                for (int hi = tcnt; hi > 0; hi -= IGD_WAVE) {
                    const int i = hi - IGD_WAVE + lane;           // lane 63 = highest index of this step
                    const bool ok = i >= 0;
                    const int s = ok ? db.start[toff + i] : INT_MAX;
                    const int e = ok ? db.end[toff + i] : INT_MIN;
                    const bool hit = (s < qe) & (s >= lob) & (e > qs);
                    const u64 m = __ballot(hit);
                    if (FILL && hit) {
                        igd_hip_hit h;
                        h.q = q; h.idx = db.idx[toff + i]; h.start = s; h.end = e;
                        if (SEQ) { h.start = i; h.end = seq_similarity_bits(qs, qe, s, e); }
                        out[base + cnt + __popcll(m & above)] = h;
                    }
                    cnt += __popcll(m);
                    // all starts in this step are below lob => so is everything before it
                    const int smax = __builtin_amdgcn_readlane(s, 63);
                    if (smax < lob) break;
                }
            }
        }
        if (!FILL && lane == 0) qcount[q] = cnt;
    }
}

// exclusive scan of int64 per-query counts -> qoff[0..n] (two kernels, like the tile scan)
__global__ __launch_bounds__(IGD_SCAN_BLOCK) void k_scan64_sums(const int64_t *__restrict__ in, int n,
                                                                int64_t *__restrict__ blockSums)
{
    __shared__ int64_t red[IGD_SCAN_BLOCK / IGD_WAVE];
    const int base = blockIdx.x * IGD_SCAN_TILE + threadIdx.x * IGD_SCAN_ITEMS;
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < IGD_SCAN_ITEMS; k++)
        if (base + k < n) s += in[base + k];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t t = 0;
        for (int w = 0; w < IGD_SCAN_BLOCK / IGD_WAVE; w++) t += red[w];
        blockSums[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(IGD_SCAN_BLOCK) void k_scan64_apply(const int64_t *__restrict__ in, int n,
                                                                 const int64_t *__restrict__ blockSums,
                                                                 int64_t *__restrict__ out /* n+1 */)
{
    __shared__ int64_t red[IGD_SCAN_BLOCK / IGD_WAVE];
    __shared__ int64_t wsum[IGD_SCAN_BLOCK / IGD_WAVE];
    int64_t pre = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += IGD_SCAN_BLOCK) pre += blockSums[b];
    for (int o = 32; o > 0; o >>= 1) pre += __shfl_down(pre, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = pre;
    const int base = blockIdx.x * IGD_SCAN_TILE + threadIdx.x * IGD_SCAN_ITEMS;
    int64_t v[IGD_SCAN_ITEMS];
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < IGD_SCAN_ITEMS; k++) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        s += v[k];
    }
    int64_t inc = s;
    const int lane = threadIdx.x & 63;
    for (int o = 1; o < 64; o <<= 1) {
        int64_t t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    int64_t run = inc - s;
    for (int w = 0; w < IGD_SCAN_BLOCK / IGD_WAVE; w++) run += red[w];
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) run += wsum[w];
#pragma unroll
    for (int k = 0; k < IGD_SCAN_ITEMS; k++) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
        if (base + k == n - 1) out[n] = run;             // the grand total closes the offsets
    }
}

// ------------------------------------------------------------------------------------------
// `-m`: dataset x dataset hit map (getMap src/igd_search.c:772-826, getMap_v :829-886;
// SURVEY 8f row f3).  Tile by tile, every record j is a query against its own tile:
//     hitmap[idx_j][idx_i]++   for every i with  start_i < end_j && end_i > start_j
//                                                && (start_j >= bd || start_i >= bd)   [&& value_j,value_i > v]
// (the last clause is the reference's tS skip, :803-804: two records that both begin before the
// tile were already paired in an earlier tile; its maxE early exit, :791-796/:811, only shortens
// the scan).  One workgroup per tile; each thread owns a record j of a 256-record slice and walks
// the tile's records, staged 256 at a time in LDS (broadcast reads), leaving as soon as the sorted
// starts pass every end of the slice.  Counters are the reference's uint32.
#define IGD_MAP_WG 256
template <bool USE_V>
__global__ __launch_bounds__(IGD_MAP_WG) void igd_hitmap_tiles(DbView db, int v, uint32_t *__restrict__ hitmap,
                                                               u64 *__restrict__ total)
{
    __shared__ int32_t sS[IGD_MAP_WG], sE[IGD_MAP_WG], sX[IGD_MAP_WG], sV[IGD_MAP_WG];
    __shared__ u64 red[IGD_MAP_WG / IGD_WAVE];
    u64 found = 0;
    for (int t = blockIdx.x; t < db.nT; t += gridDim.x) {
        const int cnt = db.tileCnt[t];
        if (cnt == 0) continue;
        const int64_t off = db.tileOff[t];
        const int tb = db.tileBd[t];
        const int bd = tb == INT_MIN ? 0 : tb;                       // getMap uses nbp*n1, also for tile 0
        for (int jb = 0; jb < cnt; jb += IGD_MAP_WG) {
            const int j = jb + (int)threadIdx.x;
            bool act = j < cnt;
            const int qs = act ? db.start[off + j] : 0;
            const int qe = act ? db.end[off + j] : INT_MIN;
            const int jj = act ? db.idx[off + j] : 0;
            if (USE_V && act) act = db.value[off + j] > v;
            const bool prefix = qs < bd;
            uint32_t *row = hitmap + (size_t)jj * (size_t)db.nFiles;
            for (int ib = 0; ib < cnt; ib += IGD_MAP_WG) {
                const int i = ib + (int)threadIdx.x;
                __syncthreads();
                if (i < cnt) {
                    sS[threadIdx.x] = db.start[off + i];
                    sE[threadIdx.x] = db.end[off + i];
                    sX[threadIdx.x] = db.idx[off + i];
                    if (USE_V) sV[threadIdx.x] = db.value[off + i];
                }
                __syncthreads();
                // sorted by start: once the first start of this stage is >= every end of the slice, done
                if (!__syncthreads_or(act && qe > sS[0])) break;
                const int nB = cnt - ib < IGD_MAP_WG ? cnt - ib : IGD_MAP_WG;
                if (act) {
                    for (int k = 0; k < nB; k++) {
                        const int s = sS[k];
                        if (s >= qe) break;                          // this thread's partners end here
                        bool hit = sE[k] > qs && (!prefix || s >= bd);
                        if (USE_V) hit = hit && sV[k] > v;
                        if (hit) { atomicAdd(&row[sX[k]], 1u); found++; }
                    }
                }
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) found += __shfl_down(found, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = found;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 s = 0;
        for (int w = 0; w < IGD_MAP_WG / IGD_WAVE; w++) s += red[w];
        if (s) atomicAdd(total, s);
    }
}

// ------------------------------------------------------------------------------------------
// instrumentation: exact terms of the algorithmic byte model, thread per query.
__device__ __forceinline__ int lower_bound_start(const int32_t *s, int n, int key)
{
    int lo = 0, hi = n;          // first index with s[i] >= key
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (s[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ void k_batch_stats(DbView db, const int32_t *__restrict__ ichr,
                              const int32_t *__restrict__ qs, const int32_t *__restrict__ qe, int nq,
                              int rule, u64 *__restrict__ acc /* queries,pairs,S,B */)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    u64 nqv = 0, pairs = 0, S = 0, B = 0;
    if (i < nq) {
        int gt0, ntl;
        int c = ichr[i];
        // "reached the tile logic": valid contig and n1 in range (before the NEST test)
        if (c >= 0 && c < db.nCtg) {
            int n1 = qs[i] / db.nbp;
            if (n1 >= 0 && n1 <= db.ctgNTile[c] - 1) nqv = 1;
        }
        if (query_span(db, c, qs[i], qe[i], rule, gt0, ntl)) {
            for (int k = 0; k < ntl; k++) {
                int t = gt0 + k;
                int cnt = db.tileCnt[t];
                if (cnt == 0) continue;
                const int32_t *s = db.start + db.tileOff[t];
                if (!(qe[i] > s[0])) continue;
                int hi = lower_bound_start(s, cnt, qe[i]);
                int lo = (k == 0) ? 0 : lower_bound_start(s, cnt, db.tileBd[t]);
                pairs++;
                S += hi > lo ? (u64)(hi - lo) : 0;
                int b = 0;
                while ((1ll << b) < (long long)cnt + 1) b++;
                B += b;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        nqv += __shfl_down(nqv, o); pairs += __shfl_down(pairs, o);
        S += __shfl_down(S, o); B += __shfl_down(B, o);
    }
    if ((threadIdx.x & 63) == 0) {
        if (nqv) atomicAdd(&acc[0], nqv);
        if (pairs) atomicAdd(&acc[1], pairs);
        if (S) atomicAdd(&acc[2], S);
        if (B) atomicAdd(&acc[3], B);
    }
}

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
#if IGD_EXP & 1024
    {
        u64 h[8];
        (void)hipDeviceSynchronize();
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(d_sect), sizeof h) == hipSuccess)
            fprintf(stderr, "[igd sect] stage %.3f  A %.3f  later %.3f  B %.3f  prefix %.3f  of the waves' time in the unit loop (%llu ticks)\n",
                    (double)h[0] / h[5], (double)h[1] / h[5], (double)h[2] / h[5], (double)h[3] / h[5], (double)h[4] / h[5], (unsigned long long)h[5]);
        fprintf(stderr, "[igd sect] waiting for records %.3f, compare phases of all visited units %.3f\n", (double)h[6] / h[5], (double)h[7] / h[5]);
    }
#endif
#if IGD_EXP & 32
    if (g_stamps) {
        std::vector<u64> h((size_t)g_stampWaves * 5);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), g_stamps, h.size() * 8, hipMemcpyDeviceToHost);
        FILE *f = fopen("gpurun_out/stamps.bin", "wb");
        if (f) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
    }
#endif
    if (t_arenaOwner == db) t_arenaOwner = nullptr;
    if (db->inner) { igd_hip_close(db->inner); db->inner = nullptr; }
    (void)hipSetDevice(db->device);
    if (db->d_rEmpty) (void)hipFree(db->d_rEmpty);
    void *ptrs[] = {db->d_start, db->d_end, db->d_idx, db->d_value, db->d_tileOff, db->d_tileCnt,
                    db->d_tileBd, db->d_ctgBase, db->d_ctgNTile, db->d_tileUnit0, db->d_heavy, db->d_far,
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
    int ndev = igd_hip_device_count();
    if (ndev <= 0) {
        if (!g_err[0]) snprintf(g_err, sizeof g_err, "igd_hip_open: no HIP device");
        return IGD_HIP_ERR_DEVICE;
    }
    if (device < 0 || device >= ndev) {
        snprintf(g_err, sizeof g_err, "igd_hip_open: device %d out of range (%d visible)", device, ndev);
        return IGD_HIP_ERR_ARG;
    }
    HIPCHK(hipSetDevice(device));
    OPEN_PHASE("HIP runtime init");
    igd_hip_db *db = new igd_hip_db();   // value-initialised: every field zero
    db->device = device;
    {   // the environment is read here, once: the per-batch entry points look at nothing but the handle
        const char *fr = getenv("IGD_HIP_RANK");
        db->forceRank = fr && *fr ? atoi(fr) : -1;
        const char *fb = getenv("IGD_HIP_BIG");
        db->bigImage = fb && *fb == '1';                 // (|| the record count, once it is known)
        db->qbVec1 = getenv("IGD_HIP_QB_VEC1") != nullptr;
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

    int rc;
    int64_t *acct = &db->resident;
#define TRY(x) do { rc = (x); if (rc != IGD_HIP_OK) { igd_hip_close(db); return rc; } } while (0)
#define TRYHIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_err(#x, e_, __FILE__, __LINE__); igd_hip_close(db); return IGD_HIP_ERR_DEVICE; } } while (0)
    TRYHIP(hipStreamCreateWithFlags(&db->stream, hipStreamNonBlocking));
    OPEN_PHASE("host tables, stream");
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
            int spare = (160 * 1024 / 2 - 512 - hitB) / (IGD_WG_RANK / IGD_WAVE) - IGD_WLDS_BYTES;   // the full build: 2 workgroups per CU
            db->sbCap = 0;                               // a power of two (s_compute pads the array to one)
            for (int c = 64; c <= 2048 && c <= spare / 2; c <<= 1) db->sbCap = c;
            if (db->sbCap == 0) {                        // counters that leave two workgroups per CU nothing: one workgroup, with the arrays
                spare = (160 * 1024 - 512 - hitB) / (IGD_WG_RANK / IGD_WAVE) - IGD_WLDS_BYTES;
                for (int c = 64; c <= 2048 && c <= spare / 2; c <<= 1) db->sbCap = c;
            }
            db->ldsSorted = hitB + (IGD_WG_RANK / IGD_WAVE) * (IGD_WLDS_BYTES + 2 * db->sbCap);
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
        size_t total = 16 * n + (willPack ? 10 * (n + IGD_CHUNK) : 0) + 112 * ((size_t)nT + 2) + (sizeof(Unit) + 4) * (units.size() + 1) + 4 * (IGD_HEAVY_MAX + IGD_HEAVYS_MAX) +
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
    v.tileOff = db->d_tileOff; v.tileCnt = db->d_tileCnt; v.tileBd = db->d_tileBd;
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
        int sh = 8;
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
    k_split_local<<<nWG, SP_WG, 0, st>>>(db->v, d_ichr, d_qs, d_qe, nq, rule, packed, db->spShift, db->spCoarse, db->d_spTable,
                                         db->d_spT, db->d_long, db->d_ctl, gate, db->epoch, zeroHits, zeroTotal, blong);
    if (!shared)
    k_split_fine<<<db->spCoarse, SPF_WG, (size_t)2 * 4 << db->spShift, st>>>(db->nT, db->spShift, db->spCoarse, nWG, db->d_spTable,
                                                                          db->d_spT,
                                                                          db->d_pairN, db->d_pairPos, (int2 *)db->d_pairs,
                                                                          db->d_ctl, gate, db->d_ctl, db->epoch, packed ? db->d_heavy : nullptr);
    else {
        k_split_fine_a<<<db->spCoarse * SPF_S, SPF_WG, (size_t)2 * 4 << db->spShift, st>>>(db->nT, db->spShift, db->spCoarse, nWG, db->d_spTable, db->d_spT,
            db->d_spSub, bases, blong, db->d_pairN, db->d_pairPos, (int2 *)db->d_pairs, db->d_ctl, gate, db->d_ctl, db->epoch, packed ? db->d_heavy : nullptr);
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
    sa.stamps = nullptr;
#if IGD_EXP & 32
    {   // diagnostic build: the LAST launch's stamps are dumped by igd_hip_close (gpurun_out/stamps.bin)
        static u64 *d_st = nullptr;
        if (!d_st) (void)hipMalloc((void **)&d_st, (size_t)db->grid * (IGD_WG / IGD_WAVE) * 5 * 8);
        sa.stamps = d_st;
        g_stamps = d_st; g_stampWaves = db->grid * (IGD_WG / IGD_WAVE);
    }
#endif
    SortK K;
    K.db = db->v; K.a = sa; K.hitsOut = (u64 *)a.hitsOut; K.totalOut = a.total;
    return K;
}

// One launch of the merge join's kernel.  A timed launch (igd_hip_profile_begin) of a promised-sorted batch passes its event
// pair to the launch itself: hipExtLaunchKernel stamps them with the dispatch's own start and end -- the figures a profiler
// reads (rocprofv3 --kernel-trace) -- where two hipEventRecord packets around the kernel also time the two packet gaps
// (3-6 % of a 70 us kernel) and put two more packets between the step's kernels.
template <typename F>
static void launch_sorted(igd_hip_db *db, F kernel, int grid, int block, size_t lds, hipStream_t st, const SortK &K)
{
    if (db->evStart) {
        SortK k = K;
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
    if (runsK && VEC_ == 4 && FAST_)                                                                                                  \
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
        if (wide) { if (fast) QB_LAUNCH(4, true, IGD_QB_WIDE); else QB_LAUNCH(4, false, IGD_QB_WIDE); }
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
    const int valves = (mode != 1 && packed && db->spShift >= 0 ? 1 : 0) | (mode != 2 && packed ? 2 : 0) |
                       (db->bigImage || db->nRec + IGD_CHUNK >= (1ll << 30) ? 4 : 0);
    // (the valve's slices of IGD_HEAVY_SLICE queries are beyond any LDS array of query starts: its waves get none)
    size_t tailLds = (valves & 2) ? (size_t)(db->ldsHits ? IGD_TAIL_WG / IGD_WAVE : 4) * (size_t)IGD_WLDS_BYTES : 0;   // (k_exact_walk: workgroups of 4 waves)
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
            Kt.a.sbCap = 0; Kt.a.wldsBytes = IGD_WLDS_BYTES; Kt.a.tailHistOff = tailHistOff;
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
            flags &= ~IGD_HIP_FLAG_SORTED;
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

// ------------------------------------------------------------------------------------------
// Several devices driven by ONE process (the C host's form of SURVEY.md 8e; the process-per-GPU form is bench.py over
// torch.distributed).  The database is resident on every device of the group, a query set is cut into contiguous slabs,
// one per device, and the path's ONE exchange -- the sum of the nFiles-long hits[] vectors, which the reference keeps as one
// accumulator over all queries and prints once (src/igd_search.c:925,1032-1039) -- is an RCCL all-reduce over xGMI,
// enqueued on every engine's own stream right behind its kernels: ncclAllReduce(d_hits, nFiles, ncclInt64, ncclSum).
// librccl is mapped only when a group is created (it is large, and a one-device `igd search` has no use for it).  If it
// cannot be loaded or the communicators cannot be built, the vectors are added on the host instead (SURVEY.md section 5's
// fallback; igd_hip_group_reduce_kind() says which of the two a group uses, IGD_MULTI_REDUCE=host forces the second).
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <thread>
struct RcclApi {
    void *lib;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*GroupStart)(void);
    ncclResult_t (*GroupEnd)(void);
    const char *(*GetErrorString)(ncclResult_t);
};
static RcclApi *rccl_api(void)
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, []() {
        memset(&api, 0, sizeof api);
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.lib) break;
        }
        if (!api.lib) return;
        api.CommInitAll = (decltype(api.CommInitAll))dlsym(api.lib, "ncclCommInitAll");
        api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
        api.AllReduce = (decltype(api.AllReduce))dlsym(api.lib, "ncclAllReduce");
        api.GroupStart = (decltype(api.GroupStart))dlsym(api.lib, "ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))dlsym(api.lib, "ncclGroupEnd");
        api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
        if (!api.CommInitAll || !api.CommDestroy || !api.AllReduce || !api.GroupStart || !api.GroupEnd) { dlclose(api.lib); api.lib = nullptr; }
    });
    return api.lib ? &api : nullptr;
}

#define IGD_GROUP_MAX 16
struct igd_hip_group {
    int n;
    igd_hip_db *db[IGD_GROUP_MAX];
    ncclComm_t comm[IGD_GROUP_MAX];
    bool rccl;
    char why[256];                // why the host add is used instead of RCCL (empty: RCCL is)
};

extern "C" int igd_hip_group_create(igd_hip_db *const *dbs, int n, igd_hip_group **out)
{
    if (!dbs || !out || n < 1 || n > IGD_GROUP_MAX) { snprintf(g_err, sizeof g_err, "igd_hip_group_create: bad argument"); return IGD_HIP_ERR_ARG; }
    for (int r = 0; r < n; r++)
        if (!dbs[r] || dbs[r]->nFiles != dbs[0]->nFiles) { snprintf(g_err, sizeof g_err, "igd_hip_group_create: the databases differ"); return IGD_HIP_ERR_ARG; }
    igd_hip_group *g = new (std::nothrow) igd_hip_group();
    if (!g) return IGD_HIP_ERR_NOMEM;
    g->n = n; g->rccl = false; g->why[0] = 0;
    int devs[IGD_GROUP_MAX];
    bool distinct = true;
    for (int r = 0; r < n; r++) {
        g->db[r] = dbs[r]; g->comm[r] = nullptr; devs[r] = dbs[r]->device;
        for (int k = 0; k < r; k++) if (devs[k] == devs[r]) distinct = false;
    }
    const char *how = getenv("IGD_MULTI_REDUCE");
    if (how && !strcmp(how, "host")) snprintf(g->why, sizeof g->why, "IGD_MULTI_REDUCE=host");
    else if (!distinct) snprintf(g->why, sizeof g->why, "a device is listed twice: RCCL wants one rank per GPU");
    else if (RcclApi *A = rccl_api()) {
        // RCCL announces its version on STDOUT when the first communicator is built -- stdout is the command line tool's
        // result (the reference's table): file descriptor 1 points at stderr while the communicators are made
        fflush(stdout);
        const int keep = dup(1);
        if (keep >= 0) (void)dup2(2, 1);
        const ncclResult_t e = A->CommInitAll(g->comm, n, devs);
        fflush(stdout);
        if (keep >= 0) { (void)dup2(keep, 1); close(keep); }
        if (e == ncclSuccess) g->rccl = true;
        else {
            snprintf(g->why, sizeof g->why, "ncclCommInitAll: %s", A->GetErrorString ? A->GetErrorString(e) : "failed");
            for (int r = 0; r < n; r++) g->comm[r] = nullptr;
        }
    } else snprintf(g->why, sizeof g->why, "librccl could not be loaded: %s", dlerror() ? dlerror() : "?");
    if (!g->rccl && how && !strcmp(how, "rccl")) {           // the caller insists: no silent host add
        snprintf(g_err, sizeof g_err, "igd_hip_group_create: IGD_MULTI_REDUCE=rccl but %s", g->why);
        delete g;
        return IGD_HIP_ERR_DEVICE;
    }
    *out = g;
    return IGD_HIP_OK;
}

extern "C" void igd_hip_group_destroy(igd_hip_group *g)
{
    if (!g) return;
    if (g->rccl) if (RcclApi *A = rccl_api()) for (int r = 0; r < g->n; r++) if (g->comm[r]) (void)A->CommDestroy(g->comm[r]);
    delete g;
}

extern "C" const char *igd_hip_group_reduce_kind(const igd_hip_group *g) { return !g ? "" : g->rccl ? "rccl" : "host"; }
extern "C" const char *igd_hip_group_reduce_note(const igd_hip_group *g) { return g ? g->why : ""; }

extern "C" int igd_hip_group_search(igd_hip_group *g, const int32_t *ichr, const int32_t *qs, const int32_t *qe, int64_t nq,
                                    int32_t v, int rule, int flags, int64_t *hits, int64_t *total)
{
    if (!g || !hits || nq < 0 || (nq > 0 && (!ichr || !qs || !qe))) { snprintf(g_err, sizeof g_err, "igd_hip_group_search: bad argument"); return IGD_HIP_ERR_ARG; }
    if (total) *total = 0;
    const int n = g->n;
    const int32_t nf = g->db[0]->nFiles;
    if (nq == 0 || nf == 0) return IGD_HIP_OK;
    // contiguous slabs (the rule of igd_amd/dist.py shard_bounds), one host thread per device; an empty slab leaves zeros
    int rcs[IGD_GROUP_MAX];
    char errs[IGD_GROUP_MAX][256];
    std::thread th[IGD_GROUP_MAX];
    const int64_t base = nq / n, rem = nq % n;
    auto work = [&](int r) {
        const int64_t lo = r * base + (r < rem ? r : rem), m = base + (r < rem ? 1 : 0);
        rcs[r] = search_slab_resident(g->db[r], ichr + lo, qs + lo, qe + lo, m, v, rule, flags);
        errs[r][0] = 0;
        if (rcs[r] != IGD_HIP_OK) snprintf(errs[r], sizeof errs[r], "%s", g_err);      // thread-local text
    };
    for (int r = 1; r < n; r++) th[r] = std::thread(work, r);
    work(0);
    int rc = IGD_HIP_OK;
    for (int r = 0; r < n; r++) {
        if (r > 0) th[r].join();
        if (rcs[r] != IGD_HIP_OK && rc == IGD_HIP_OK) { rc = rcs[r]; snprintf(g_err, sizeof g_err, "%s", errs[r]); }
    }
    if (rc != IGD_HIP_OK) return rc;
    std::vector<int64_t> h((size_t)nf);
    int64_t tot = 0;
    if (g->rccl) {
        RcclApi *A = rccl_api();
        // the one exchange of the path: every device ends up with the sum, device 0's copy is returned
        ncclResult_t e = A->GroupStart();
        for (int r = 0; r < n && e == ncclSuccess; r++) {
            e = A->AllReduce(g->db[r]->d_hits, g->db[r]->d_hits, (size_t)nf, ncclInt64, ncclSum, g->comm[r], g->db[r]->stream);
            if (e == ncclSuccess) e = A->AllReduce(g->db[r]->d_total, g->db[r]->d_total, 1, ncclInt64, ncclSum, g->comm[r], g->db[r]->stream);
        }
        const ncclResult_t e2 = A->GroupEnd();
        if (e == ncclSuccess) e = e2;
        if (e != ncclSuccess) { snprintf(g_err, sizeof g_err, "igd_hip_group_search: ncclAllReduce: %s", A->GetErrorString ? A->GetErrorString(e) : "failed"); return IGD_HIP_ERR_DEVICE; }
        for (int r = 0; r < n; r++) { HIPCHK(hipSetDevice(g->db[r]->device)); HIPCHK(hipStreamSynchronize(g->db[r]->stream)); }
        HIPCHK(hipSetDevice(g->db[0]->device));
        HIPCHK(hipMemcpy(h.data(), g->db[0]->d_hits, (size_t)nf * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(&tot, g->db[0]->d_total, 8, hipMemcpyDeviceToHost));
        for (int32_t f = 0; f < nf; f++) hits[f] += h[f];
    } else {
        for (int r = 0; r < n; r++) {
            int64_t t1 = 0;
            HIPCHK(hipSetDevice(g->db[r]->device));
            HIPCHK(hipMemcpy(h.data(), g->db[r]->d_hits, (size_t)nf * 8, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(&t1, g->db[r]->d_total, 8, hipMemcpyDeviceToHost));
            for (int32_t f = 0; f < nf; f++) hits[f] += h[f];
            tot += t1;
        }
    }
    if (total) *total = tot;
    return IGD_HIP_OK;
}

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
static int enumerate_core(igd_hip_db *db, const int32_t *ichr, const int32_t *qs, const int32_t *qe, int64_t nq,
                          int64_t *qoff, bool wantWhole, igd_hip_hit **wholeOut, igd_hip_enum_sink sink, void *ctx, int64_t *total)
{
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
        if (sink && sink(ctx, 0, nq, qoff, nullptr) != 0) {
            snprintf(g_err, sizeof g_err, "igd_hip_enumerate_stream: stopped by the sink");
            return IGD_HIP_ERR_ARG;
        }
        return IGD_HIP_OK;
    }
    int64_t maxq = 0;
    for (int64_t i = 0; i < nq; i++) if (qoff[i + 1] - qoff[i] > maxq) maxq = qoff[i + 1] - qoff[i];
    if (maxq > db->enumChunkCap) {                         // one query larger than a chunk buffer: grow them
        rc = ensure_enum_workspace(db, nq, maxq, !wantWhole);
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
    const int64_t cap = db->enumChunkCap;
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
            igd_enum_queries<true><<<egrid, 256, 0, st>>>(db->v, db->d_qc, db->d_qs, db->d_qe, (int)qa, (int)qb, nullptr,
                                                          db->d_qoff, qoff[qa], zeroCopy ? hostDst : db->d_enumOut[b]);
            e = hipEventRecord(db->evFill[b], st);
            if (e == hipSuccess) e = hipStreamWaitEvent(db->copyStream, db->evFill[b], 0);
            if (e == hipSuccess && !zeroCopy)
                e = hipMemcpyAsync(hostDst, db->d_enumOut[b], (size_t)nh * sizeof(igd_hip_hit), hipMemcpyDeviceToHost, db->copyStream);
            if (e == hipSuccess) e = hipEventRecord(db->evCopy[b], db->copyStream);
            if (e != hipSuccess) break;
        }
        if (sink && k >= 1 && prevB > prevA) {            // hand out the previous chunk while this one is produced
            if (qoff[prevB] > qoff[prevA]) e = hipEventSynchronize(db->evCopy[(k - 1) & 1]);
            if (e == hipSuccess) sinkRc = sink(ctx, prevA, prevB, qoff, db->h_enumPin[(k - 1) & 1]);
        }
        prevA = qa; prevB = qb;
        qa = qb;
        if (nh > 0 || sink) k++;
    }
    {   // both streams are drained whatever happened: after a failed call an earlier chunk's copy may still be writing into
        // `whole` or a pinned chunk buffer, which are released / reused right below
        const hipError_t e1 = hipStreamSynchronize(db->copyStream), e2 = hipStreamSynchronize(st);
        if (e == hipSuccess) e = e1;
        if (e == hipSuccess) e = e2;
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess && sink && sinkRc == 0 && prevB > prevA) sinkRc = sink(ctx, prevA, prevB, qoff, db->h_enumPin[(k - 1) & 1]);
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

// ------------------------------------------------------------------------------------------
// Seqpare (`search -q f.bed -s`), seqOverlaps src/igd_search.c:354-451.
// For every dataset m and every contig of the query file the reference repeatedly takes the best
// remaining (query, record) pair -- strict '>' while scanning queries in order, pairs of a query in
// discovery order -- and drops the pair's query ("row") and record ("column" = (index in tile, first
// tile of the QUERY): the reference stores idx_t = n1 for every tile, :291,:337).  That is greedy
// matching by the total order (similarity descending, position ascending), and the groups
// (query contig, dataset) are independent: entries are made by the `-f` enumeration kernel, brought
// together per group and put in that total order by two stable radix sorts (igd_sortscan.hpp: by
// similarity descending, then by group -- ties keep the reference's scan order), and each group is
// resolved by ONE WAVE, 64 candidates at a time, with hash sets of the taken rows and columns.
// Per dataset the accepted similarities are finally added up in double, contig by contig in the
// query file's order and in acceptance order -- the reference's order of additions.
#define SQ_CAP 1024                  // groups up to this size keep their hash sets in LDS
#define SQ_TAB (2 * SQ_CAP)
struct SeqArgs {
    const igd_hip_hit *ent;          // (q, idx_f, idx_g, sm bits), query-major, discovery order
    const uint32_t *vals;            // entry numbers: grouped, inside a group by (similarity desc, position asc)
    const int64_t *goff;             // [nGroups*nFiles + 1]
    const int32_t *q_qs;             // query starts (n1 = qs / nbp)
    float *sel;                      // accepted similarities, at goff[g] + i
    int32_t *nsel;                   // [nGroups*nFiles]
    int32_t *x_rows; unsigned long long *x_cols;   // HBM hash sets of groups > SQ_CAP: 4 slots per entry, preset to ~0
    unsigned int *next;
    int64_t nG;                      // number of groups
    int32_t nbp;
};

__device__ __forceinline__ uint32_t sq_hash32(uint32_t x) { x *= 0x9E3779B1u; return x ^ (x >> 15); }
__device__ __forceinline__ uint32_t sq_hash64(unsigned long long x)
{
    x *= 0x9E3779B97F4A7C15ull;
    return (uint32_t)(x >> 32) ^ (uint32_t)x;
}

// One wave per group.  Candidates arrive in the greedy order; 64 at a time: every lane looks its
// row and column up in the hash sets of what was accepted before this batch, then the batch is
// settled lane by lane (an accepted earlier lane knocks out later lanes that share its row or
// column), and the survivors are inserted and written out in lane order = acceptance order.
__global__ void __launch_bounds__(64) k_seq_greedy(SeqArgs a)
{
    __shared__ int32_t l_rows[SQ_TAB];
    __shared__ unsigned long long l_cols[SQ_TAB];
    __shared__ unsigned int gShared;
    const int lane = threadIdx.x;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (;;) {
        if (lane == 0) gShared = atomicAdd(a.next, 1u);
        __syncthreads();
        const int64_t g = gShared;
        __syncthreads();
        if (g >= a.nG) return;
        const int64_t off = a.goff[g];
        const int64_t n = a.goff[g + 1] - off;
        if (n == 0) { if (lane == 0) a.nsel[g] = 0; continue; }
        const bool small = n <= SQ_CAP;
        int32_t *rows = small ? l_rows : a.x_rows + 4 * off;
        unsigned long long *cols = small ? l_cols : a.x_cols + 4 * off;
        uint32_t mask = SQ_TAB - 1;
        if (small) {
            for (int i = lane; i < SQ_TAB; i += 64) { l_rows[i] = -1; l_cols[i] = ~0ull; }
        } else {
            uint32_t t = 1;
            while ((int64_t)t < 2 * n) t <<= 1;               // <= 4n slots, preset by the host
            mask = t - 1;
        }
        __syncthreads();
        int32_t nacc = 0;
        for (int64_t base = 0; base < n; base += 64) {
            const int64_t p = base + lane;
            bool valid = p < n;
            igd_hip_hit h;
            h.q = 0; h.idx = 0; h.start = 0; h.end = 0;
            if (valid) h = a.ent[a.vals[off + p]];
            const float x = __int_as_float(h.end);
            valid = valid && (x > 0.0f);
            if (__ballot(valid) == 0) break;                    // sorted: nothing positive is left
            const int32_t r = h.q;
            const unsigned long long k = ((unsigned long long)(uint32_t)(a.q_qs[h.q] / a.nbp) << 32) | (uint32_t)h.start;
            bool out = !valid;
            if (valid) {                                        // accepted in an earlier batch?
                for (uint32_t s = sq_hash32((uint32_t)r) & mask;; s = (s + 1) & mask) {
                    const int32_t v = __hip_atomic_load(&rows[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // not from a stale L1 line
                    if (v == r) { out = true; break; }
                    if (v == -1) break;
                }
                if (!out)
                    for (uint32_t s = sq_hash64(k) & mask;; s = (s + 1) & mask) {
                        const unsigned long long v = __hip_atomic_load(&cols[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (v == k) { out = true; break; }
                        if (v == ~0ull) break;
                    }
            }
            const int lim = (n - base) < 64 ? (int)(n - base) : 64;
            for (int j = 0; j < lim; j++) {                     // settle the batch in candidate order
                const int alive = __builtin_amdgcn_readlane((int)!out, j);
                if (!alive) continue;
                const int32_t rj = __builtin_amdgcn_readlane(r, j);
                const uint32_t klo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)k, j);
                const uint32_t khi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(k >> 32), j);
                const unsigned long long kj = ((unsigned long long)khi << 32) | klo;
                if (lane > j && (r == rj || k == kj)) out = true;
            }
            const unsigned long long accm = __ballot(!out);
            if (!out) {
                for (uint32_t s = sq_hash32((uint32_t)r) & mask;; s = (s + 1) & mask)
                    if (atomicCAS(&rows[s], -1, r) == -1) break;
                for (uint32_t s = sq_hash64(k) & mask;; s = (s + 1) & mask)
                    if (atomicCAS(&cols[s], ~0ull, k) == ~0ull) break;
                a.sel[off + nacc + __popcll(accm & lt)] = x;
            }
            nacc += __popcll(accm);
            __syncthreads();
        }
        if (lane == 0) a.nsel[g] = nacc;
        __syncthreads();
    }
}

// first sort key: similarity descending (positive floats order like their bit patterns)
__global__ void k_seq_key1(const igd_hip_hit *__restrict__ ent, int64_t n, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        keys[i] = ~(uint32_t)ent[i].end;
        vals[i] = (uint32_t)i;
    }
}

// second sort key: the group (query contig * nFiles + dataset) of the entry now at position i; group histogram
__global__ void k_seq_key2(const igd_hip_hit *__restrict__ ent, const uint32_t *__restrict__ vals, int64_t n,
                           const int32_t *__restrict__ qgrp, int32_t nFiles, uint32_t *__restrict__ keys, uint32_t *__restrict__ gcnt)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const igd_hip_hit h = ent[vals[i]];
        const uint32_t k = (uint32_t)qgrp[h.q] * (uint32_t)nFiles + (uint32_t)h.idx;
        keys[i] = k;
        atomicAdd(&gcnt[k], 1u);
    }
}

// sm[m] += maxf in the reference's order: contigs of the query file outermost, acceptance order inside
__global__ void k_seq_accumulate(const float *__restrict__ sel, const int32_t *__restrict__ nsel, const int64_t *__restrict__ goff,
                                 int32_t nGroups, int32_t nFiles, double *__restrict__ sums)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= nFiles) return;
    double acc = sums[m];                                     // 0, or the running sum of the earlier contigs (igd_hip_seqpare_add)
    for (int32_t c = 0; c < nGroups; c++) {
        const int64_t g = (int64_t)c * nFiles + m;
        const int64_t o = goff[g];
        const int32_t k = nsel[g];
        for (int32_t i = 0; i < k; i++) acc += (double)sel[o + i];
    }
    sums[m] = acc;
}

static int seqpare_core(igd_hip_db *db, const int32_t *ichr, const int32_t *qs, const int32_t *qe, int64_t nq,
                        const int32_t *qgroup, int32_t nGroups, double *sums, bool carry)
{
    if (!db || !sums || nq < 0 || nq > max_batch() || nGroups < 0 || (nq > 0 && (!ichr || !qs || !qe || !qgroup)) ||
        (int64_t)nGroups * db->nFiles >= 0x7fffffffLL) {
        snprintf(g_err, sizeof g_err, "igd_hip_seqpare: bad argument (batch limit %lld queries)", (long long)max_batch());
        return IGD_HIP_ERR_ARG;
    }
    if (db->gType != 1) {
        snprintf(g_err, sizeof g_err, "igd_hip_seqpare: needs a gType-1 database (seq_overlaps reads 16-byte records)");
        return IGD_HIP_ERR_ARG;
    }
    if (!carry) for (int32_t m = 0; m < db->nFiles; m++) sums[m] = 0.0;
    if (nq == 0 || db->nT == 0 || nGroups == 0 || db->nFiles == 0) return IGD_HIP_OK;
    HIPCHK(hipSetDevice(db->device));
    hipStream_t st = db->stream;
    int rc = ensure_qstage(db, nq);
    if (rc != IGD_HIP_OK) return rc;
    const int64_t nG = (int64_t)nGroups * db->nFiles;
    int32_t *d_qgrp = nullptr, *d_nsel = nullptr, *x_rows = nullptr;
    int64_t *d_qcount = nullptr, *d_qoff = nullptr, *d_bsum = nullptr, *d_goff = nullptr, *d_dig = nullptr,
            *d_sums = nullptr, *d_tot = nullptr;
    unsigned long long *x_cols = nullptr;
    uint32_t *kA = nullptr, *vA = nullptr, *kB = nullptr, *vB = nullptr, *d_gcnt = nullptr, *d_hist = nullptr;
    float *d_sel = nullptr;
    double *d_out = nullptr;
    unsigned int *d_next = nullptr;
    igd_hip_hit *d_ent = nullptr;
    uint32_t *h_gcnt = nullptr;
    auto cleanup = [&]() {
        void *ps[] = { d_qgrp, d_nsel, x_rows, d_qcount, d_qoff, d_bsum, d_goff, d_dig, d_sums,
                       d_tot, x_cols, kA, vA, kB, vB, d_gcnt, d_hist, d_sel, d_out, d_next, d_ent };
        for (void *p : ps) if (p) (void)hipFree(p);
        free(h_gcnt);
    };
#define EH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_err(#x, e_, __FILE__, __LINE__); cleanup(); return IGD_HIP_ERR_DEVICE; } } while (0)
#define DA(p, n) do { if ((rc = dalloc(&(p), (size_t)(n), nullptr)) != IGD_HIP_OK) { cleanup(); return rc; } } while (0)
    DA(d_qcount, nq); DA(d_qoff, nq + 1);
    DA(d_bsum, nq / IGD_SCAN_TILE + 2); DA(d_qgrp, nq);
    EH(hipMemcpyAsync(db->d_qc, ichr, (size_t)nq * 4, hipMemcpyHostToDevice, st));
    EH(hipMemcpyAsync(db->d_qs, qs, (size_t)nq * 4, hipMemcpyHostToDevice, st));
    EH(hipMemcpyAsync(db->d_qe, qe, (size_t)nq * 4, hipMemcpyHostToDevice, st));
    EH(hipMemcpyAsync(d_qgrp, qgroup, (size_t)nq * 4, hipMemcpyHostToDevice, st));
    igd_enum_queries<false, true><<<db->grid * 4, 256, 0, st>>>(db->v, db->d_qc, db->d_qs, db->d_qe, 0, (int)nq, d_qcount, nullptr, 0, nullptr);
    {
        const int sb = (int)((nq + IGD_SCAN_TILE - 1) / IGD_SCAN_TILE);
        k_scan64_sums<<<sb, IGD_SCAN_BLOCK, 0, st>>>(d_qcount, (int)nq, d_bsum);
        k_scan64_apply<<<sb, IGD_SCAN_BLOCK, 0, st>>>(d_qcount, (int)nq, d_bsum, d_qoff);
    }
    int64_t E = 0;
    EH(hipMemcpyAsync(&E, d_qoff + nq, 8, hipMemcpyDeviceToHost, st));
    EH(hipStreamSynchronize(st));
    if (E == 0) { cleanup(); return IGD_HIP_OK; }
    if (E >= 0xffffffffLL) { cleanup(); snprintf(g_err, sizeof g_err, "igd_hip_seqpare: %lld overlaps exceed one batch", (long long)E); return IGD_HIP_ERR_ARG; }
    DA(d_ent, E);
    igd_enum_queries<true, true><<<db->grid * 4, 256, 0, st>>>(db->v, db->d_qc, db->d_qs, db->d_qe, 0, (int)nq, nullptr, d_qoff, 0, d_ent);
    // order the entries: stable radix sort by similarity (descending), then by group (query contig * nFiles +
    // dataset) -> inside a group: similarity descending, ties in the reference's scan order
    DA(kA, E); DA(vA, E); DA(kB, E); DA(vB, E); DA(d_gcnt, nG + 1); DA(d_goff, nG + 1); DA(d_tot, 8);
    EH(hipMemsetAsync(d_gcnt, 0, (size_t)(nG + 1) * 4, st));
    {
        const int64_t nh = rs_blocks(E) * 256;
        const int64_t nsum = ((nh > nG + 1 ? nh : nG + 1) + SCAN_TILE - 1) / SCAN_TILE + 1;
        DA(d_hist, nh); DA(d_dig, nh); DA(d_sums, nsum);
        k_seq_key1<<<db->grid * 4, 256, 0, st>>>(d_ent, E, kA, vA);
        EH(radix_sort_pairs(&kA, &vA, &kB, &vB, E, 32, d_hist, d_dig, d_sums, d_tot, st));
        k_seq_key2<<<db->grid * 4, 256, 0, st>>>(d_ent, vA, E, d_qgrp, db->nFiles, kA, d_gcnt);
        EH((exclusive_scan<uint32_t, int64_t>(d_gcnt, nG + 1, d_goff, d_sums, d_tot, st)));
        int bits = 0;
        while (bits < 32 && ((uint64_t)(nG - 1) >> bits)) bits++;
        EH(radix_sort_pairs(&kA, &vA, &kB, &vB, E, bits, d_hist, d_dig, d_sums, d_tot, st));
    }
    // largest group decides whether the HBM hash sets of k_seq_greedy are needed
    h_gcnt = (uint32_t *)malloc((size_t)nG * 4);
    if (!h_gcnt) { cleanup(); return IGD_HIP_ERR_NOMEM; }
    EH(hipMemcpyAsync(h_gcnt, d_gcnt, (size_t)nG * 4, hipMemcpyDeviceToHost, st));
    EH(hipStreamSynchronize(st));
    uint32_t maxG = 0;
    for (int64_t g = 0; g < nG; g++) if (h_gcnt[g] > maxG) maxG = h_gcnt[g];
    if (maxG > SQ_CAP) {
        DA(x_rows, 4 * E); DA(x_cols, 4 * E);
        EH(hipMemsetAsync(x_rows, 0xff, (size_t)E * 16, st));
        EH(hipMemsetAsync(x_cols, 0xff, (size_t)E * 32, st));
    }
    DA(d_sel, E); DA(d_nsel, nG); DA(d_next, 16); DA(d_out, db->nFiles);
    EH(hipMemsetAsync(d_next, 0, 4, st));
    if (carry) EH(hipMemcpyAsync(d_out, sums, (size_t)db->nFiles * 8, hipMemcpyHostToDevice, st));   // the running sums continue
    else EH(hipMemsetAsync(d_out, 0, (size_t)db->nFiles * 8, st));
    {
        SeqArgs a;
        a.ent = d_ent; a.vals = vA; a.goff = d_goff; a.q_qs = db->d_qs; a.sel = d_sel; a.nsel = d_nsel;
        a.x_rows = x_rows; a.x_cols = x_cols;
        a.next = d_next; a.nG = nG; a.nbp = db->nbp;
        const int64_t want = nG < (int64_t)db->grid * 3 ? nG : (int64_t)db->grid * 3;
        k_seq_greedy<<<(unsigned)want, 64, 0, st>>>(a);
        k_seq_accumulate<<<(db->nFiles + 63) / 64, 64, 0, st>>>(d_sel, d_nsel, d_goff, nGroups, db->nFiles, d_out);
    }
    EH(hipGetLastError());
    EH(hipMemcpyAsync(sums, d_out, (size_t)db->nFiles * 8, hipMemcpyDeviceToHost, st));
    EH(hipStreamSynchronize(st));
#undef EH
#undef DA
    cleanup();
    return IGD_HIP_OK;
}

extern "C" int igd_hip_seqpare(igd_hip_db *db, const int32_t *ichr, const int32_t *qs, const int32_t *qe, int64_t nq,
                               const int32_t *qgroup, int32_t nGroups, double *sums)
{
    return seqpare_core(db, ichr, qs, qe, nq, qgroup, nGroups, sums, false);
}
extern "C" int igd_hip_seqpare_add(igd_hip_db *db, const int32_t *ichr, const int32_t *qs, const int32_t *qe, int64_t nq,
                                   const int32_t *qgroup, int32_t nGroups, double *sums)
{
    return seqpare_core(db, ichr, qs, qe, nq, qgroup, nGroups, sums, true);
}

extern "C" int igd_hip_hitmap(igd_hip_db *db, int use_v, int32_t v, uint32_t *hitmap, int64_t *total)
{
    if (!db || !hitmap) {
        snprintf(g_err, sizeof g_err, "igd_hip_hitmap: bad argument");
        return IGD_HIP_ERR_ARG;
    }
    if (total) *total = 0;
    if (db->gType != 1) {
        snprintf(g_err, sizeof g_err, "igd_hip_hitmap: needs a gType-1 database (the reference's getMap reads 16-byte records)");
        return IGD_HIP_ERR_ARG;
    }
    const size_t cells = (size_t)db->nFiles * (size_t)db->nFiles;
    if (cells == 0 || db->nT == 0) return IGD_HIP_OK;
    if (cells * 4 > ((size_t)64 << 30)) {
        snprintf(g_err, sizeof g_err, "igd_hip_hitmap: %d x %d matrix does not fit", db->nFiles, db->nFiles);
        return IGD_HIP_ERR_NOMEM;
    }
    HIPCHK(hipSetDevice(db->device));
    uint32_t *d_map = nullptr;
    u64 *d_tot = nullptr;
    int rc;
    if ((rc = dalloc(&d_map, cells, nullptr)) != IGD_HIP_OK) return rc;
    if ((rc = dalloc(&d_tot, 1, nullptr)) != IGD_HIP_OK) { (void)hipFree(d_map); return rc; }
    hipStream_t st = db->stream;
    hipError_t e = hipMemsetAsync(d_map, 0, cells * 4, st);
    if (e == hipSuccess) e = hipMemsetAsync(d_tot, 0, 8, st);
    if (e == hipSuccess) {
        int grid = db->nT < 256 * 16 ? db->nT : 256 * 16;
        if (use_v) igd_hitmap_tiles<true><<<grid, IGD_MAP_WG, 0, st>>>(db->v, v, d_map, d_tot);
        else igd_hitmap_tiles<false><<<grid, IGD_MAP_WG, 0, st>>>(db->v, v, d_map, d_tot);
        e = hipGetLastError();
    }
    std::vector<uint32_t> h;
    u64 tot = 0;
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) { h.resize(cells); e = hipMemcpy(h.data(), d_map, cells * 4, hipMemcpyDeviceToHost); }
    if (e == hipSuccess) e = hipMemcpy(&tot, d_tot, 8, hipMemcpyDeviceToHost);
    (void)hipFree(d_map); (void)hipFree(d_tot);
    if (e != hipSuccess) { set_err("hitmap", e, __FILE__, __LINE__); return IGD_HIP_ERR_DEVICE; }
    for (size_t c = 0; c < cells; c++) hitmap[c] += h[c];            // hitmap[][]++ semantics: added to
    if (total) *total = (int64_t)tot;
    return IGD_HIP_OK;
}

extern "C" int igd_hip_batch_stats(igd_hip_db *db, const int32_t *d_ichr, const int32_t *d_qs,
                                   const int32_t *d_qe, int64_t nq, int32_t v, int rule, igd_hip_stats *out)
{
    if (!db || !out || nq < 0 || nq > IGD_MAX_BATCH) return IGD_HIP_ERR_ARG;
    memset(out, 0, sizeof *out);
    if (nq == 0) return IGD_HIP_OK;
    HIPCHK(hipSetDevice(db->device));
    u64 *d_acc = nullptr;
    int64_t *d_h = nullptr, *d_t = nullptr;
    int rc;
    if ((rc = dalloc(&d_acc, 4, nullptr)) != IGD_HIP_OK) return rc;
    if ((rc = dalloc(&d_h, (size_t)db->nFiles + 1, nullptr)) != IGD_HIP_OK) { (void)hipFree(d_acc); return rc; }
    if ((rc = dalloc(&d_t, 1, nullptr)) != IGD_HIP_OK) { (void)hipFree(d_acc); (void)hipFree(d_h); return rc; }
    hipStream_t st = db->stream;
    (void)hipMemsetAsync(d_acc, 0, 32, st);
    (void)hipMemsetAsync(d_h, 0, ((size_t)db->nFiles + 1) * 8, st);
    (void)hipMemsetAsync(d_t, 0, 8, st);
    k_batch_stats<<<(int)((nq + 255) / 256), 256, 0, st>>>(db->v, d_ichr, d_qs, d_qe, (int)nq, rule, d_acc);
    bool saved = db->evOn;
    db->evOn = false;
    rc = igd_hip_search_dev(db, d_ichr, d_qs, d_qe, nq, v, rule, 0, d_h, d_t, st);
    db->evOn = saved;
    u64 acc[4] = {0, 0, 0, 0};
    int64_t tot = 0;
    hipError_t e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipMemcpy(acc, d_acc, 32, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(&tot, d_t, 8, hipMemcpyDeviceToHost);
    (void)hipFree(d_acc); (void)hipFree(d_h); (void)hipFree(d_t);
    if (rc != IGD_HIP_OK) return rc;
    if (e != hipSuccess) { set_err("batch_stats", e, __FILE__, __LINE__); return IGD_HIP_ERR_DEVICE; }
    out->queries = (int64_t)acc[0]; out->pairs = (int64_t)acc[1];
    out->S = (int64_t)acc[2]; out->B = (int64_t)acc[3]; out->H = tot;
    return IGD_HIP_OK;
}

// ------------------------------------------------------------------------------------------
// instrumentation: compulsory traffic of the scan kernel for one batch (include/igd_hip.h)
__global__ void k_unit_traffic(DbView db, const int32_t *__restrict__ firstQ, const int32_t *__restrict__ pairN,
                               const int32_t *__restrict__ spill, int epoch, int path /* 0 bucket, 1 merge join exact, 2 merge join compact */,
                               int rankOK, u64 *__restrict__ acc /* units, records, pairs (bucket path), queries of rank-method tiles */)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    u64 nu = 0, nr = 0, np = 0, nd = 0;
    if (u < db.nUnits) {
        const Unit un = db.units[u];
        if (un.n > 0) {
            const int lj = UNIT_J(un);
            const int lb = lj < IGD_SHORT_TILES - 1 ? lj : IGD_SHORT_TILES - 1;
            const bool firstUnit = UNIT_FLAGS(un) & 1;
            if (path == 1) {        // exactly the test of issue_unit: the candidate range of the unit's tile is not empty
                if (firstQ[un.tile + 1] > firstQ[un.tile - lb]) { nu = 1; nr = (u64)un.n; }
            } else if (path == 2) { // exactly the test of s_issue: first-tile queries, or a marked tile with earlier queries
                const int f0 = firstQ[un.tile], c0 = firstQ[un.tile + 1] - f0;
                const int cl = spill[un.tile] == epoch ? f0 - firstQ[un.tile - lb] : 0;
                if (c0 | cl) { nu = 1; nr = (u64)un.n; }
                if (firstUnit) {
                    if (rankOK && c0 >= IGD_DENSE_MIN) nd = (u64)c0;
                }
            } else if (pairN[un.tile] != 0) {             // negative: the tile went to heavy_bucket_body
                nu = 1; nr = (u64)un.n;
                if (firstUnit) np = (u64)(pairN[un.tile] < 0 ? -pairN[un.tile] : pairN[un.tile]);
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        nu += __shfl_down(nu, o); nr += __shfl_down(nr, o); np += __shfl_down(np, o); nd += __shfl_down(nd, o);
    }
    if ((threadIdx.x & 63) == 0) {
        if (nu) atomicAdd(&acc[0], nu);
        if (nr) atomicAdd(&acc[1], nr);
        if (np) atomicAdd(&acc[2], np);
        if (nd) atomicAdd(&acc[3], nd);
    }
}

extern "C" int igd_hip_batch_traffic(igd_hip_db *db, const int32_t *d_ichr, const int32_t *d_qs, const int32_t *d_qe,
                                     int64_t nq, int32_t v, int rule, int flags, igd_hip_traffic *out)
{
    if (!db || !out || nq < 0 || nq > IGD_MAX_BATCH) return IGD_HIP_ERR_ARG;
    if (db->inner) {                                     // (the kernels of such a database run on its re-tiled copy)
        db->inner->vnest = rule == IGD_HIP_RULE_NEST ? 1 : 0;
        return igd_hip_batch_traffic(db->inner, d_ichr, d_qs, d_qe, nq, v, IGD_HIP_RULE_FLAT, flags, out);
    }
    memset(out, 0, sizeof *out);
    if (nq == 0 || db->nT == 0 || db->nFiles == 0) return IGD_HIP_OK;
    HIPCHK(hipSetDevice(db->device));
    u64 *d_acc = nullptr;
    int64_t *d_h = nullptr;
    int rc;
    // an order promise of the caller's own batches is settled first: the sync that closes this measurement clears the
    // device's sticky report, and no broken batch may go unreported
    if (db->promised && (rc = igd_hip_sync(db, db->stream)) != IGD_HIP_OK) return rc;
    if ((rc = dalloc(&d_acc, 4, nullptr)) != IGD_HIP_OK) return rc;
    if ((rc = dalloc(&d_h, (size_t)db->nFiles + 1, nullptr)) != IGD_HIP_OK) { (void)hipFree(d_acc); return rc; }
    hipStream_t st = db->stream;
    (void)hipMemsetAsync(d_acc, 0, 32, st);
    (void)hipMemsetAsync(d_h, 0, ((size_t)db->nFiles + 1) * 8, st);
    const bool saved = db->evOn;
    db->evOn = false;
    rc = igd_hip_search_dev(db, d_ichr, d_qs, d_qe, nq, v, rule, flags & ~IGD_HIP_FLAG_ZERO_FIRST, d_h, nullptr, st);
    db->evOn = saved;
    int32_t ctl[4] = {0, 0, 0, 0};
    hipError_t e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipMemcpy(ctl, db->d_ctl, sizeof ctl, hipMemcpyDeviceToHost);
    if (rc == IGD_HIP_OK && e == hipSuccess) {
        const int mode = (flags & IGD_HIP_FLAG_SORTED) ? 1 : (flags & IGD_HIP_FLAG_BUCKET) ? 2 : 0;
        const bool sortedPath = mode == 1 || (mode == 0 && ctl[CTL_UNSORTED] != db->epoch);
        const bool useV = (v != IGD_HIP_NO_VALUE_FILTER && db->gType == 1);
        const bool packed = db->packed && !(flags & IGD_HIP_FLAG_EXACT) && (!useV || db->packedV);
        const int path = !sortedPath ? 0 : packed ? 2 : 1;
        k_unit_traffic<<<(db->nUnits + 255) / 256, 256, 0, st>>>(db->v, db->d_firstQ, db->d_pairN, db->d_spill, db->epoch, path,
                                                               ctl[CTL_NOTSTART] != db->epoch ? 1 : 0, d_acc);
        u64 acc[4] = {0, 0, 0, 0};
        e = hipStreamSynchronize(st);
        if (e == hipSuccess) e = hipMemcpy(acc, d_acc, 32, hipMemcpyDeviceToHost);
        const int recB = packed ? (useV ? 8 : 6) : (useV ? 16 : 12);
        out->units = (int64_t)acc[0];
        out->records = (int64_t)acc[1];
        out->record_bytes = (int64_t)acc[1] * recB;
        out->unit_bytes = (int64_t)sizeof(Unit) * db->nUnits + (path == 2 ? 8ll * (db->nT + 1) : sortedPath ? 4ll * (db->nT + 1) : 8ll * db->nT);
        // merge join, compact image: one 4-byte word per query (qw0), the compacted later-tile words (later[]: every entry is
        // read at least once), the starts (q_qs) of the tiles the rank method handles; exact arrays: qw, qs, qe;
        // bucket path: 8 B per pair
        int64_t nLaterWords = 0;
        if (path == 2) {
            const int64_t nb = (nq + ((int64_t)1 << db->lbShift) - 1) >> db->lbShift;
            std::vector<int32_t> hdr((size_t)nb * 2);
            e = hipMemcpy(hdr.data(), db->d_laterHdr, (size_t)nb * 8, hipMemcpyDeviceToHost);
            for (int64_t b = 0; b < nb; b++) nLaterWords += hdr[(size_t)b * 2];
        }
        out->query_bytes = path == 2 ? 4ll * nq + 4ll * nLaterWords + 4ll * (int64_t)acc[3]
                         : path == 1 ? 12ll * nq : 8ll * (int64_t)acc[2];
        out->slab_bytes = db->ldsHits ? (int64_t)db->grid * db->nFiles * (path == 2 ? 4 : 8) : 8ll * db->nFiles;   // (merge join: 32-bit rows)
        out->total = out->record_bytes + out->unit_bytes + out->query_bytes + out->slab_bytes;
    }
    (void)hipFree(d_acc); (void)hipFree(d_h);
    (void)igd_hip_sync(db, st);                            // consumes a promise made for this measurement batch
    if (rc != IGD_HIP_OK) return rc;
    if (e != hipSuccess) { set_err("batch_traffic", e, __FILE__, __LINE__); return IGD_HIP_ERR_DEVICE; }
    return IGD_HIP_OK;
}

// ------------------------------------------------------------------------------------------
// instrumentation: what the memory system of THIS box gives simple streaming kernels (bench.py quotes
// it next to the roofline): a float4 copy (the guide's 6.29 TB/s measurement), a float4 read-only sum,
// pinned D2H / H2D copies.
__global__ __launch_bounds__(256) void k_copy16(const float4 *__restrict__ in, float4 *__restrict__ outp, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i + 3 * stride < n; i += 4 * stride) {          // four independent 16-byte loads in flight per lane
        const float4 a = in[i], b = in[i + stride], c = in[i + 2 * stride], d = in[i + 3 * stride];
        outp[i] = a; outp[i + stride] = b; outp[i + 2 * stride] = c; outp[i + 3 * stride] = d;
    }
    for (; i < n; i += stride) outp[i] = in[i];
}
__global__ __launch_bounds__(256) void k_read16(const float4 *__restrict__ in, size_t n, float *__restrict__ sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const float4 a = in[i], b = in[i + stride], c = in[i + 2 * stride], d = in[i + 3 * stride];
        acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + c.x + c.y + c.z + c.w + d.x + d.y + d.z + d.w;
    }
    for (; i < n; i += stride) { const float4 a = in[i]; acc += a.x + a.y + a.z + a.w; }
    if (acc == 123.456f) *sink = acc;                       // never true for the zero-filled buffer; keeps the loads
}

// one 16-byte vector per thread, non-temporal: the plainest streaming copy / read there is
typedef float igd_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_copy16_flat(const float4 *__restrict__ in, float4 *__restrict__ outp, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(__builtin_nontemporal_load((const igd_f4 *)in + i), (igd_f4 *)outp + i);
}
__global__ __launch_bounds__(256) void k_read16_flat(const float4 *__restrict__ in, size_t n, float *__restrict__ sink)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const igd_f4 a = __builtin_nontemporal_load((const igd_f4 *)in + i);
        if (a.x + a.y + a.z + a.w == 123.456f) *sink = a.x;
    }
}

extern "C" int igd_hip_measure_rates(int device, double rates[4])
{
    if (!rates) return IGD_HIP_ERR_ARG;
    for (int k = 0; k < 4; k++) rates[k] = 0.0;
    HIPCHK(hipSetDevice(device));
    const size_t bytes = (size_t)1 << 30, n16 = bytes / 16, hb = (size_t)256 << 20;
    float4 *a = nullptr, *b = nullptr;
    float *sink = nullptr;
    void *h = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t st = nullptr;
    hipError_t e = hipMalloc((void **)&a, bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&b, bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&sink, 16);
    if (e == hipSuccess) e = hipHostMalloc(&h, hb, hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMemsetAsync(a, 0, bytes, st);
    if (e == hipSuccess) e = hipMemsetAsync(b, 0, bytes, st);
    if (e == hipSuccess) memset(h, 0, hb);
    const int grid = 256 * 16, reps = 12;
    for (int which = 0; which < 4 && e == hipSuccess; which++) {
        float best = 1e30f;
        for (int r = 0; r < reps + 2 && e == hipSuccess; r++) {
            e = hipEventRecord(e0, st);
            // two shapes of each kernel, alternating: grid-stride with four loads in flight / one float4 per thread; best wins
            if (which == 0) { if (r & 1) k_copy16_flat<<<(unsigned)((n16 + 255) / 256), 256, 0, st>>>(a, b, n16); else k_copy16<<<grid, 256, 0, st>>>(a, b, n16); }
            else if (which == 1) { if (r & 1) k_read16_flat<<<(unsigned)((n16 + 255) / 256), 256, 0, st>>>(a, n16, sink); else k_read16<<<grid, 256, 0, st>>>(a, n16, sink); }
            else if (which == 2) { if (e == hipSuccess) e = hipMemcpyAsync(h, a, hb, hipMemcpyDeviceToHost, st); }
            else { if (e == hipSuccess) e = hipMemcpyAsync(a, h, hb, hipMemcpyHostToDevice, st); }
            if (e == hipSuccess) e = hipEventRecord(e1, st);
            if (e == hipSuccess) e = hipEventSynchronize(e1);
            float ms = 0.f;
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
            if (r >= 2 && ms < best) best = ms;
        }
        const double moved = which == 0 ? 2.0 * (double)bytes : which == 1 ? (double)bytes : (double)hb;
        if (e == hipSuccess && best > 0.f) rates[which] = moved / ((double)best * 1e-3) / 1e9;
    }
    if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    if (sink) (void)hipFree(sink);
    if (h) (void)hipHostFree(h);
    if (e != hipSuccess) { set_err("measure_rates", e, __FILE__, __LINE__); return IGD_HIP_ERR_DEVICE; }
    return IGD_HIP_OK;
}

extern "C" int igd_hip_profile_begin(igd_hip_db *db, int max_launches)
{
    if (!db || max_launches <= 0) return IGD_HIP_ERR_ARG;
    if (db->inner) return igd_hip_profile_begin(db->inner, max_launches);
    HIPCHK(hipSetDevice(db->device));
    while ((int)db->ev.size() < 4 * max_launches) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        db->ev.push_back(e);
    }
    db->evMax = max_launches;
    db->evUsed = 0;
    db->evSeen = 0;
    db->evOn = true;
    return IGD_HIP_OK;
}

extern "C" int igd_hip_profile_sampling(igd_hip_db *db, int every)
{
    if (!db || every < 1) return IGD_HIP_ERR_ARG;
    if (db->inner) return igd_hip_profile_sampling(db->inner, every);
    db->evEvery = every;
    return IGD_HIP_OK;
}

extern "C" int igd_hip_profile_end(igd_hip_db *db, int *n_launches, double *avg_scan_ms, double *avg_pipeline_ms)
{
    if (!db) return IGD_HIP_ERR_ARG;
    if (db->inner) return igd_hip_profile_end(db->inner, n_launches, avg_scan_ms, avg_pipeline_ms);
    HIPCHK(hipSetDevice(db->device));
    db->evOn = false;
    int n = db->evUsed;
    double scan = 0, pipe = 0;
    int npipe = 0;
    for (int i = 0; i < n; i++) {
        HIPCHK(hipEventSynchronize(db->ev[4 * i + 2]));
        float a = 0, b = 0;
        HIPCHK(hipEventElapsedTime(&a, db->ev[4 * i + 1], db->ev[4 * i + 2]));
        scan += a;
        if (i < IGD_PIPE_EVENTS) {
            HIPCHK(hipEventSynchronize(db->ev[4 * i + 3]));
            HIPCHK(hipEventElapsedTime(&b, db->ev[4 * i + 0], db->ev[4 * i + 3]));
            pipe += b; npipe++;
        }
    }
    if (n_launches) *n_launches = n;
    if (avg_scan_ms) *avg_scan_ms = n ? scan / n : 0.0;
    if (avg_pipeline_ms) *avg_pipeline_ms = npipe ? pipe / npipe : 0.0;
    db->evUsed = 0;
    return IGD_HIP_OK;
}
