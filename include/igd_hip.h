/* igd_hip.h -- C ABI of the MI355X (gfx950) overlap-search engine.
 *
 * This is the device-side boundary that the three host flavours of the reference's
 * igd_search.h (include/igd_search.h, include/igd_py_abi.h, include/igdr_abi.h) are built
 * on.  Plain C: opaque handle, pointers and sizes only, no C++/torch types.
 *
 * What it replaces in the reference (databio/IGD, /root/reference):
 *   igd_hip_search / _dev    the per-query loops  getOverlaps   src/igd_search.c:696-719
 *                                                 getOverlaps_v src/igd_search.c:746-769
 *                                                 getOverlaps0  src/igd_search.c:202-225
 *                            over the kernels     get_overlaps   src/igd_search.c:454-534
 *                                                 get_overlaps_v src/igd_search.c:623-694
 *                                                 get_overlaps0  src/igd_search.c:30-112
 *                            and the hits[] accumulator          src/igd_search.c:925,491,524,654,684
 *   igd_hip_enumerate        getOverlaps_f1/_f0 src/igd_search.c:721-744,227-250 over
 *                            get_overlaps_f1/_f0 src/igd_search.c:537-620,114-200
 *   igd_hip_open             the per-tile fseek/fread of src/igd_search.c:469-476 (the whole
 *                            tile region is uploaded once, transposed AoS->SoA on the GPU)
 *
 * There is NO CPU fallback behind any of these entry points: without a usable HIP device
 * they return an error code and igd_hip_last_error() says why.
 */
#ifndef IGD_HIP_H
#define IGD_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IGD_HIP_OK          0
#define IGD_HIP_ERR_DEVICE  (-1)   /* no device / HIP runtime error          */
#define IGD_HIP_ERR_ARG     (-2)   /* bad argument                            */
#define IGD_HIP_ERR_NOMEM   (-3)   /* host or device allocation failed        */
#define IGD_HIP_ERR_UNSORTED (-4)  /* IGD_HIP_FLAG_SORTED promised, order violated */

/* Tile-visiting rules. */
#define IGD_HIP_RULE_NEST 0  /* get_overlaps / get_overlaps0 / _f0 / _f1: an EMPTY first tile
                                ends the query (loop nested in `if(nCnt[n1]>0)`, :468-532)   */
#define IGD_HIP_RULE_FLAT 1  /* get_overlaps_v: every tile n1..n2 visited (:635-691)          */

typedef struct igd_hip_db igd_hip_db;          /* opaque: one .igd resident on one GPU      */

/* Host description of an .igd (the header tables of iGD_t, src/igd_base.h:96-105, plus the
 * raw tile region of the file).  Nothing is retained after igd_hip_open returns. */
typedef struct {
    int32_t nbp;              /* tile width in bp                                           */
    int32_t gType;            /* 1: 16-byte {idx,start,end,value}; 0: 12-byte {idx,start,end}*/
    int32_t nCtg;
    int32_t nFiles;           /* length of hits[] (from _index.tsv)                          */
    const int32_t *nTile;     /* [nCtg]                                                     */
    const int32_t *nCnt;      /* contig-major, sum(nTile) entries (file order)               */
    const void *records;      /* host pointer: all tile records in file order (AoS), or NULL: */
    int64_t nRecords;         /* = sum(nCnt)                                                */
    int     fd;               /* records == NULL: read them from this descriptor ...         */
    int64_t fd_offset;        /* ... starting at this byte offset (header size of the .igd)  */
} igd_hip_desc;

/* One emitted overlap of the `-f` path: query number (position in the batch), then the
 * record as the reference prints it (src/igd_search.c:577,610). */
typedef struct { int32_t q, idx, start, end; } igd_hip_hit;

/* Exact work statistics of one batch under one rule -- the terms of the algorithmic byte
 * model.  Instrumentation; not part of a search. */
typedef struct {
    int64_t queries;  /* queries with a valid contig and n1 in range                         */
    int64_t pairs;    /* (query,tile) pairs with cnt>0 && qe > first start                   */
    int64_t S;        /* sum of scan lengths                                                */
    int64_t B;        /* sum of ceil(log2(cnt+1))                                           */
    int64_t H;        /* hits                                                               */
} igd_hip_stats;

int         igd_hip_device_count(void);               /* <=0: none usable                    */
const char *igd_hip_last_error(void);                 /* thread-local, never NULL            */

int  igd_hip_open(const igd_hip_desc *desc, int device, igd_hip_db **out);
void igd_hip_close(igd_hip_db *db);
int  igd_hip_device(const igd_hip_db *db);
int32_t igd_hip_nfiles(const igd_hip_db *db);
int64_t igd_hip_resident_bytes(const igd_hip_db *db); /* HBM held by the SoA image + tables  */

/* `v` of the search calls: records with value < v are not counted (get_overlaps_v's
 * `value>=v`, src/igd_search.c:652,682).  IGD_HIP_NO_VALUE_FILTER switches the predicate
 * off (get_overlaps).  Ignored for gType 0, which stores no value (:1024-1025).  Whether a
 * CLI `-v N` selects the filtered kernel (only N>0, :1027) is the HOST's decision. */
#define IGD_HIP_NO_VALUE_FILTER INT32_MIN

/* How the engine groups a batch's queries by tile (flags of the search calls):
 *   0                     the device decides: one pass checks whether the batch is ordered by
 *                         (contig index, start) -- a position-sorted BED -- and, if so, reads
 *                         the queries in place (merge join); otherwise it counting-sorts the
 *                         (query,tile) pairs.  Same result either way.
 *   IGD_HIP_FLAG_SORTED   the caller PROMISES that order, which skips enqueueing the bucket
 *                         kernels.  The promise is verified on the device: if it does not hold,
 *                         that batch adds nothing and igd_hip_sync returns IGD_HIP_ERR_UNSORTED.
 *                         The promise is (contig index, START) -- starts that decrease inside one
 *                         tile break it too, whichever step the batch takes (without the promise
 *                         such a batch is counted by the merge join's pairwise compares).
 *   IGD_HIP_FLAG_BUCKET   the caller KNOWS the batch is not in that order (the command line tool's
 *                         parser saw a line out of order): no order check, no merge-join launch,
 *                         always the counting sort.  Never wrong -- the counting sort takes any
 *                         order -- only slower than the merge join on a batch that is sorted after all. */
#define IGD_HIP_FLAG_SORTED 1
#define IGD_HIP_FLAG_BUCKET 2
/* Besides the exact start/end/idx/value arrays the engine keeps a compact tile-relative image
 * (6 bytes per record) that the counting kernels read when the tile width is <= 32768 and
 * nFiles <= 65536.  IGD_HIP_FLAG_EXACT makes a call read the exact arrays instead (tests). */
#define IGD_HIP_FLAG_EXACT 4
/* igd_hip_search_dev only: d_hits[] (and d_total) are cleared by the batch's first kernel before the
 * counts are added -- saves the caller a separate memset when it does not accumulate. */
#define IGD_HIP_FLAG_ZERO_FIRST 8
/* With IGD_HIP_FLAG_SORTED: the caller also states that no query is longer than one tile (qe - qs < nbp) -- what a
 * query file of peaks / regions is, and what the command line tool finds out while it parses.  A dense batch (>= 28 queries
 * per tile on average) then takes the DIRECT step: no per-query pre-pass, the scan kernel reads q_qs / q_qe itself
 * (engine/scan_direct.hpp).  VERIFIED like the order: a longer query is found where it is read and its later tiles are
 * walked exactly -- it costs time, never a count. */
#define IGD_HIP_FLAG_SHORT 16

/* Host-buffer search.  ichr[i] = contig index (as get_id returns; <0 or >=nCtg: skipped).
 * hits[0..nFiles) is caller-allocated and is ADDED to (reference semantics :491).
 * *total (may be NULL) receives the number of overlaps of this batch.  Blocking. */
int igd_hip_search(igd_hip_db *db, const int32_t *ichr, const int32_t *qs, const int32_t *qe,
                   int64_t nq, int32_t v, int rule, int64_t *hits, int64_t *total);
/* same with `flags`.  An IGD_HIP_FLAG_SORTED promise that the device finds broken is not an
 * error here: the call repeats that slice with the device choosing the grouping. */
int igd_hip_search_ex(igd_hip_db *db, const int32_t *ichr, const int32_t *qs, const int32_t *qe,
                      int64_t nq, int32_t v, int rule, int flags, int64_t *hits, int64_t *total);

/* Device-resident search: all pointers are device pointers on db's GPU; d_hits
 * (int64[nFiles]) is ADDED to; d_total (int64[1], may be NULL) is ADDED to.  Enqueues on
 * `stream` (a hipStream_t; NULL = the engine's own stream) and returns without waiting.
 * nq must be <= igd_hip_max_batch().  One call at a time per database (shared workspace). */
int igd_hip_search_dev(igd_hip_db *db, const int32_t *d_ichr, const int32_t *d_qs,
                       const int32_t *d_qe, int64_t nq, int32_t v, int rule, int flags,
                       int64_t *d_hits, int64_t *d_total, void *stream);
/* The same for a position-sorted batch given as contig RUNS instead of one contig number per query: d_run_start (device,
 * nCtg + 1 int32) with run_start[0] = 0, run_start[nCtg] = nq and the queries [run_start[c], run_start[c + 1]) lying on contig c
 * in non-decreasing order of start -- what a position-sorted BED is (the command line tool's reader keeps its queries that
 * way, igdc_queries_group_contigs).  The grouping kernel then reads 8 instead of 12 bytes per query.  Implies
 * IGD_HIP_FLAG_SORTED (IGD_HIP_FLAG_BUCKET is refused); a table that is not monotone or does not cover [0, nq), like queries out
 * of order, is a broken promise: the batch adds nothing and igd_hip_sync returns IGD_HIP_ERR_UNSORTED. */
int igd_hip_search_runs_dev(igd_hip_db *db, const int32_t *d_run_start, const int32_t *d_qs, const int32_t *d_qe,
                            int64_t nq, int32_t v, int rule, int flags, int64_t *d_hits, int64_t *d_total, void *stream);
int64_t igd_hip_max_batch(void);   /* queries per call of the host-buffer entry points (2^24; test-only IGD_HIP_MAX_BATCH lowers it) */
/* The rule behind igd_hip_max_batch(), in ONE place: the engine (igd_hip.hip) and the host flavours' lazy binding
 * (igd_hip_lazy.c, which answers without mapping the engine) both evaluate this -- they cannot drift apart. */
#define IGD_HIP_MAX_BATCH_DEFAULT ((int64_t)1 << 24)
static inline int64_t igd_hip_max_batch_rule(const char *env /* getenv("IGD_HIP_MAX_BATCH") or NULL */)
{
    long long x = 0;
    if (env && *env) { int neg = 0; const char *p = env; if (*p == '-') { neg = 1; p++; } while (*p >= '0' && *p <= '9' && x < ((long long)1 << 40)) x = x * 10 + (*p++ - '0'); if (neg) x = -x; }
    return x >= 1 && x < IGD_HIP_MAX_BATCH_DEFAULT ? (int64_t)x : IGD_HIP_MAX_BATCH_DEFAULT;
}
int  igd_hip_sync(igd_hip_db *db, void *stream);      /* wait + surface async errors         */
int  igd_hip_sync_spin(igd_hip_db *db, void *stream); /* the same, polling hipStreamQuery instead of sleeping on the signal */

/* Several devices driven by one process (the C host's form): the database resident on each device of the
 * group (igd_hip_open once per device; a device may be listed twice), a query set cut into contiguous slabs, one per device,
 * and the path's ONE exchange -- the sum of the per-device hits[nFiles] vectors, the reference's single accumulator
 * (src/igd_search.c:925,1032-1039) -- as an RCCL all-reduce (ncclInt64, ncclSum) on the engines' own streams, over xGMI.
 * librccl is mapped when the first group is created.  When it cannot be used (not loadable, communicator refused, a device
 * listed twice, IGD_MULTI_REDUCE=host) the vectors are added on the host; igd_hip_group_reduce_kind() returns "rccl" or
 * "host", igd_hip_group_reduce_note() the reason for "host"; IGD_MULTI_REDUCE=rccl makes create fail instead of falling back.
 * igd_hip_group_search: hits[] is ADDED to, *total = overlaps of the whole set.  Blocking.  The group does not own the
 * databases (destroy the group first, then close them). */
typedef struct igd_hip_group igd_hip_group;
int  igd_hip_group_create(igd_hip_db *const *dbs, int n, igd_hip_group **out);
void igd_hip_group_destroy(igd_hip_group *g);
const char *igd_hip_group_reduce_kind(const igd_hip_group *g);
const char *igd_hip_group_reduce_note(const igd_hip_group *g);
int  igd_hip_group_search(igd_hip_group *g, const int32_t *ichr, const int32_t *qs, const int32_t *qe, int64_t nq,
                          int32_t v, int rule, int flags, int64_t *hits, int64_t *total);

/* `-f`: full enumeration in reference order (queries in batch order; per query tiles
 * ascending, record index DESCENDING inside a tile; rule NEST, no value filter).
 * qoff[0..nq] receives the exclusive scan of per-query counts; *out is malloc'd by the
 * callee (free with igd_hip_free) and holds qoff[nq] records.  Blocking. */
int  igd_hip_enumerate(igd_hip_db *db, const int32_t *ichr, const int32_t *qs,
                       const int32_t *qe, int64_t nq, int64_t *qoff, igd_hip_hit **out,
                       int64_t *total);
void igd_hip_free(void *p);

/* The same enumeration, STREAMED: the overlaps are produced in chunks of contiguous query ranges
 * (<= 32 MiB of records each, a whole query never split) and each chunk is handed to `sink` from
 * pinned host memory while the next chunks are being filled and copied -- the caller (the command
 * line tool's formatter, getOverlaps_f1 src/igd_search.c:721-744) works on chunk k while chunk k+1
 * crosses PCIe.  qoff[0..nq] is complete before the first sink call; a chunk covers queries [q0,q1),
 * `hits` points at overlap number qoff[q0] (so overlap h of the batch is hits[h - qoff[q0]]) and is
 * only valid during the call.  Chunks arrive in order and together cover [0,nq) exactly once, also
 * the queries without overlaps.  A non-zero return of the sink stops the enumeration.  Blocking. */
typedef int (*igd_hip_enum_sink)(void *ctx, int64_t q0, int64_t q1, const int64_t *qoff, const igd_hip_hit *hits);
int  igd_hip_enumerate_stream(igd_hip_db *db, const int32_t *ichr, const int32_t *qs, const int32_t *qe,
                              int64_t nq, int64_t *qoff, igd_hip_enum_sink sink, void *ctx, int64_t *total);

/* The same stream in HALF the bytes (round 6; `-f` is bound by the bytes that cross PCIe, 0.81 of the link at 16 per overlap):
 * one overlap = 8 bytes.  `q` is implied by qoff[]; start is kept whole; (end - start) and idx share the second word, split
 * per DATABASE: idx takes idx_bits = ceil(log2(nFiles)) low bits, the length the 32 - idx_bits above them (1900 files: 11 + 21
 * bits, lengths up to 2 097 151 bp).  igd_hip_hit8_idx_bits() returns that split, or -1 when a record of this database does not
 * fit it (a length of 2^(32 - idx_bits) or more, or end < start): such a database streams through igd_hip_enumerate_stream only.
 * Expansion: start = (int32_t)h.start, idx = h.lenidx & ((1u << idx_bits) - 1), end = start + (int32_t)(h.lenidx >> idx_bits)
 * (igd_hip_hit8_expand).  Order, chunking, qoff and the sink's contract are those of igd_hip_enumerate_stream. */
typedef struct { uint32_t start, lenidx; } igd_hip_hit8;
typedef int (*igd_hip_enum_sink8)(void *ctx, int64_t q0, int64_t q1, const int64_t *qoff, const igd_hip_hit8 *hits, int idx_bits);
int  igd_hip_hit8_idx_bits(igd_hip_db *db);
int  igd_hip_enumerate_stream8(igd_hip_db *db, const int32_t *ichr, const int32_t *qs, const int32_t *qe,
                               int64_t nq, int64_t *qoff, igd_hip_enum_sink8 sink, void *ctx, int64_t *total);
static inline igd_hip_hit igd_hip_hit8_expand(igd_hip_hit8 h, int idx_bits, int32_t q)
{
    igd_hip_hit r;
    r.q = q;
    r.start = (int32_t)h.start;
    r.idx = (int32_t)(idx_bits ? (h.lenidx & ((1u << idx_bits) - 1u)) : 0u);
    r.end = (int32_t)(h.start + (idx_bits < 32 ? (h.lenidx >> idx_bits) : 0u));
    return r;
}

/* `-m`: dataset x dataset hit map, getMap src/igd_search.c:772-826 (use_v = 0) and getMap_v
 * :829-886 (use_v = 1: both records need value > v, strictly).  hitmap is nFiles x nFiles uint32,
 * row-major, caller-allocated, ADDED to; *total (may be NULL) receives the number of pairs.
 * gType-1 databases only.  Blocking. */
int igd_hip_hitmap(igd_hip_db *db, int use_v, int32_t v, uint32_t *hitmap, int64_t *total);

/* Seqpare (`search -q f.bed -s`): seqOverlaps src/igd_search.c:354-451 over
 * seq_overlaps :253-352.  Queries as the reference orders them: the contigs of the query file in
 * first-seen order, inside a contig by start (ties in file order); qgroup[i] = number of the query's
 * contig in that order (0..nGroups-1, non-decreasing).  sums[m] (nFiles doubles) receives the sum of
 * the greedily matched similarities of dataset m, added up in the reference's order; the caller
 * finishes with sm = sums/(Nq + nr - sums) (:446-449).  gType-1 databases; one batch; blocking. */
int igd_hip_seqpare(igd_hip_db *db, const int32_t *ichr, const int32_t *qs, const int32_t *qe, int64_t nq,
                    const int32_t *qgroup, int32_t nGroups, double *sums);
/* The same for a query file beyond one batch: the contigs of the file are passed range by range, in
 * order; sums[] is NOT cleared but continued, so the additions happen in the reference's order. */
int igd_hip_seqpare_add(igd_hip_db *db, const int32_t *ichr, const int32_t *qs, const int32_t *qe, int64_t nq,
                        const int32_t *qgroup, int32_t nGroups, double *sums);

/* `igd create`: intervals -> the tile region of an .igd, on the GPU.
 * Replaces igd_add (src/igd_base.c:118-169: replicate into tiles start/nbp..(end-1)/nbp),
 * igd_saveT (:333-364: per-tile append in input order) and igd_save (:396-461: per-tile
 * radix_sort_intv, src/igd_base.h:196-249, then contig-major concatenation).  The records come
 * out in EXACTLY the reference's order, including its (unstable) order of equal starts.
 * Input: n intervals in input order (files in glob order, lines in file order), every one with
 * 0 <= start < end; ctg[] = contig numbers in first-seen order, file[] = index of the source
 * file (gdata_t.idx); value may be NULL (0).  Host arrays; nothing is retained.
 * Output: out_fd < 0 -> the records come back in pinned host memory (igd_hip_created.records);
 * out_fd >= 0 -> the engine writes the complete .igd (header of SURVEY.md App. A with ctgName[],
 * zero-padded to 40 bytes, then the tiles) to that descriptor through two pinned staging buffers,
 * device->host copies overlapping the write()s, and records stays NULL. */
typedef struct {
    int32_t nbp, gType, nCtg;
    int64_t n;
    const int32_t *ctg, *start, *end, *value, *file;
    const char *const *ctgName;   /* [nCtg], needed only with out_fd >= 0                       */
    int out_fd;
} igd_hip_create_desc;
typedef struct {
    int32_t *nTile;           /* [nCtg]   tiles per contig = 1 + max (end-1)/nbp               */
    int32_t *nCnt;            /* [nTiles] records per tile, contig-major (the header table)     */
    int64_t nTiles, nRecords;
    void *records;            /* nRecords x 16 (gType 1) or 12 (gType 0) bytes, file order, pinned */
} igd_hip_created;
int  igd_hip_create(const igd_hip_create_desc *d, int device, igd_hip_created *out);
void igd_hip_created_free(igd_hip_created *c);

/* What the loaded library was compiled as.  igd_hip_build_flags(): bits 0..23 = the IGD_EXP experiment mask of the build
 * (0 in a shipped library), bit 24 = IGD_EXP_NOMATCH.  igd_hip_build_wrong_counts(): the subset of those bits that make the
 * kernels give WRONG counts on purpose (section-by-section measurement builds, tools/valu_ab.sh).  igd_hip_open refuses
 * such a library unless IGD_HIP_ALLOW_EXP_BUILD=1 is set, and igd_amd.Database refuses it outright. */
unsigned igd_hip_build_flags(void);
unsigned igd_hip_build_wrong_counts(void);

/* Instrumentation ------------------------------------------------------------------- */
/* Exact algorithmic-work terms for a device-resident batch (blocking). */
int igd_hip_batch_stats(igd_hip_db *db, const int32_t *d_ichr, const int32_t *d_qs,
                        const int32_t *d_qe, int64_t nq, int32_t v, int rule,
                        igd_hip_stats *out);
/* COMPULSORY traffic of the dominant kernel for one batch: the bytes igd_scan_tiles cannot avoid moving
 * through HBM when every visited unit's records are read exactly once -- the denominator-free part of
 * a roofline fraction that stays <= 1 (the algorithmic bytes above price the REFERENCE's re-reads).
 * Runs the batch once (grouping as `flags` say) into scratch counters and then counts, on the device,
 * the units the scan kernel visited.  Blocking. */
typedef struct {
    int64_t units;          /* units (<= 320-record chunks of a tile) with at least one candidate query     */
    int64_t records;        /* records in them                                                            */
    int64_t record_bytes;   /* records x bytes per record of the image read (6 / 8 compact, 12 / 16 exact) */
    int64_t unit_bytes;     /* 48-byte descriptors of ALL units + the per-tile query ranges / marks read     */
    int64_t query_bytes;    /* per-query words the kernel reads, each once (4 B/query in the compact merge join) */
    int64_t slab_bytes;     /* the workgroups' private counter rows written at the end of the kernel       */
    int64_t total;          /* sum of the four                                                            */
} igd_hip_traffic;
int igd_hip_batch_traffic(igd_hip_db *db, const int32_t *d_ichr, const int32_t *d_qs, const int32_t *d_qe,
                          int64_t nq, int32_t v, int rule, int flags, igd_hip_traffic *out);
/* What this box's memory system delivers to simple streaming kernels, measured now (GB/s, 1e9):
 * rates[0] float4 copy kernel, read+write bytes (the guide's 6.29 TB/s figure is this measurement);
 * rates[1] float4 read-only kernel; rates[2] pinned device->host copy; rates[3] pinned host->device. */
int igd_hip_measure_rates(int device, double rates[4]);
/* HIP-event timing of the launches made by igd_hip_search_dev on their own stream:
 * begin() arms up to max_launches slots, end() waits and returns the number of launches
 * seen plus the average duration (ms) of the dominant scan kernel and of the whole
 * pipeline (bucket + scan + reduce). */
int igd_hip_profile_begin(igd_hip_db *db, int max_launches);
int igd_hip_profile_end(igd_hip_db *db, int *n_launches, double *avg_scan_ms,
                        double *avg_pipeline_ms);
/* Time only every `every`-th launch (default 1): an event is a packet of its own in the stream between two kernels and
 * costs the job about as much as a small kernel (10^6-query steps: 104 -> 95 us when the pipeline pair is dropped,
 * another ~4 us for the scan pair), so a job that is itself being timed samples.  Sticky per database. */
int igd_hip_profile_sampling(igd_hip_db *db, int every);
/* Name of the dominant kernel as rocprofv3 --kernel-trace prints it (for profiles/). */
const char *igd_hip_scan_kernel_name(void);
/* ... and the one the last batch of `db` actually ran on: "igd_scan_sorted" (merge join over the compact image) or
 * "igd_scan_tiles" (bucket path, exact arrays).  Waits for the batch. */
const char *igd_hip_last_scan_kernel(igd_hip_db *db);

#ifdef __cplusplus
}
#endif
#endif
