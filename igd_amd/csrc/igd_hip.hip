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
#define IGD_MAX_BATCH ((long long)IGD_HIP_MAX_BATCH_DEFAULT)   // queries per device batch (include/igd_hip.h)
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
                       // compaction of the later-tile words; 0x10000 / 0x20000 the last launch without heavy_sorted_body /
                       // far_units_body, 0x40000 / 0x80000 k_query_bounds without the firstQ[] fill / the head and tail fill,
                       // 0x100000 / 0x200000 the last launch without the exact walks / the coverage sums, 0x800000 k_query_bounds
                       // without the lpos[] stores of short gaps (all WRONG counts).  (The time-stamp builds of rounds 2-5 -- bits 32,
                       // 1024, 0x400000, 0x1000000 -- left the source in round 6; their results are in LABNOTES.md.)
#endif
#ifndef IGD_ASM_MATCH
#define IGD_ASM_MATCH 1 // igd_scan_sorted's pairwise compare loop written out in assembly (0: the compiler's everywhere, 2: written out in the lean build only)
#endif
#ifndef IGD_NT_AUX
#define IGD_NT_AUX 0   // cache policy of igd_scan_sorted's record loads (measured: 2 = nt is 6 % slower -- consecutive batches find part of the image in the Infinity Cache)
#endif
#ifndef IGD_OPT_PRIO
#define IGD_OPT_PRIO 1 // igd_scan_sorted: waves lower their issue priority as they get through their share
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
        return igd_hip_max_batch_rule(getenv("IGD_HIP_MAX_BATCH"));      // (include/igd_hip.h: shared with igd_hip_lazy.c)
    }();
    return m;
}
extern "C" int64_t igd_hip_max_batch(void) { return max_batch(); }

// What this library was compiled as: bits 0..23 = IGD_EXP, bit 24 = IGD_EXP_NOMATCH.  A build with any bit of
// IGD_HIP_BUILD_WRONG_COUNTS gives WRONG counts on purpose (measurement of kernel sections): igd_hip_open refuses to
// work in such a build unless IGD_HIP_ALLOW_EXP_BUILD=1 says the caller knows (tools/valu_ab.sh does).
#define IGD_EXP_WRONG_BITS (1 | 2 | 4 | 8 | 64 | 128 | 256 | 512 | 8192 | 0x10000 | 0x20000 | 0x40000 | 0x80000 | 0x100000 | 0x200000 | 0x800000)
extern "C" unsigned igd_hip_build_flags(void) { return ((unsigned)IGD_EXP & 0xffffffu) | (IGD_EXP_NOMATCH ? 1u << 24 : 0u); }
extern "C" unsigned igd_hip_build_wrong_counts(void);     // (defined behind the engine's parts: igd_scan_direct's measurement bits count too)

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
    const uint32_t *tileBits;                   // [(nT + 31) / 32] bit t: tile t holds records (what the grouping of an unordered batch asks of tileCnt[], 32 tiles per word)
    const int4 *tileD;                          // [nT] per tile: the records that start in the NEXT tile, the contig, the tiles left in it (k_tile_desc; null: no DIRECT step)
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
    int forceDirect;              // IGD_HIP_DIRECT at open (tests): 1 every batch under IGD_HIP_FLAG_SORTED that can takes the DIRECT step, 0 none, -1 the engine decides
    int4 *d_tileD;                // DbView::tileD
    uint32_t *d_tileBits;         // DbView::tileBits
    int ldsDirect;                // dynamic LDS of igd_scan_direct
    int lastDirect;               // the last batch took the DIRECT step (igd_hip_last_scan_kernel)
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
    int hit8State, hit8Bits;                 // igd_hip_hit8 (8 bytes per overlap): 0 not looked at yet, 1 fits (hit8Bits = bits of idx), 2 does not
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
// The engine is ONE translation unit (the kernels are templates over the same views and the host code
// instantiates them), kept in parts by subject; each part is included exactly once, in this order.
#include "engine/upload.hpp"          // AoS -> SoA transposes of the .igd records at open
#include "engine/query_bounds.hpp"    // tile arithmetic, k_query_bounds (queries grouped by tile, fix list, coverage differences), k_count_pairs
#include "engine/split.hpp"           // unordered batches: two-level split into tile buckets (k_split_local / k_split_fine*), scans, scatter
#include "engine/compact_image.hpp"   // k_pack_units: the 6-byte tile-relative image and its unit descriptors
#include "engine/scan_tiles.hpp"      // igd_scan_tiles (bucket path) and its skew valve
#include "engine/scan_sorted.hpp"     // igd_scan_sorted: the merge join (pairwise and rank builds) -- the dominant kernel
#include "engine/scan_direct.hpp"     // dense sorted batches without a per-query pre-pass: k_tile_bounds, igd_scan_direct
#include "engine/tail.hpp"            // exact walks, coverage, k_reduce_slabs (last launch of a batch), k_sum_hits
#include "engine/enumerate_dev.hpp"   // `-f` enumeration kernels
#include "engine/hitmap_dev.hpp"      // `-m` hit map kernel
#include "engine/batch_stats_dev.hpp" // instrumentation: terms of the algorithmic byte model
#include "engine/host_open.hpp"       // handles: allocation, close, pinned buffers, re-tiled copy, igd_hip_open
#include "engine/host_search.hpp"     // workspaces, launches, igd_hip_search_dev / _runs_dev / _search / _search_ex, sync
#include "engine/host_group.hpp"      // device groups of one process: native RCCL all-reduce of hits[]
#include "engine/host_enumerate.hpp"  // `-f` on the host side: chunked, double-buffered
#include "engine/seqpare.hpp"         // Seqpare `-s`: kernels and host
#include "engine/host_misc.hpp"       // igd_hip_hitmap, igd_hip_batch_stats
#include "engine/measure.hpp"         // instrumentation: compulsory traffic, streaming rates of the box, launch profile
extern "C" unsigned igd_hip_build_wrong_counts(void)
{
    return (((unsigned)IGD_EXP) & (unsigned)IGD_EXP_WRONG_BITS) | (IGD_EXP_NOMATCH ? 1u << 24 : 0u) | ((IGD_D_EXP & 47) ? 1u << 25 : 0u);
}
