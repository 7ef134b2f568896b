/* igd_core.h -- host-side core shared by the three ABI flavours (CLI, Python handle, R).
 *
 * Internal header (not installed): the public faces are include/igd_search.h,
 * include/igd_py_abi.h and include/igdr_abi.h.  Everything here is plain C99; the only
 * thing it calls for the search itself is the HIP engine of include/igd_hip.h -- there is
 * no CPU search path in this library.
 *
 * Reference counterparts (databio/IGD, /root/reference):
 *   igdc_open           get_igdinfo  src/igd_base.c:269-323  (+ whole tile region, once)
 *   igdc_load_index     get_fileinfo src/igd_base.c:235-267
 *   igdc_get_id         get_id       src/igd_base.c:325-331  (khash str->int, src/khash.h)
 *   igdc_parse_bed      parse_bed    src/igd_base.c:53-72
 *   igdc_lines_*        ks_getuntil  src/kseq.h:82-130 over gzread (src/igd_base.h:192)
 *   igdc_read_queries   the read/parse/lookup part of getOverlaps src/igd_search.c:708-714
 */
#ifndef IGD_CORE_H
#define IGD_CORE_H
#include <stdint.h>
#include <stdio.h>
#include "igd_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define IGDC_MAX_DEVICES 16
typedef struct igdc_db {
    int32_t nbp, gType, nCtg, nFiles;
    int32_t *nTile;          /* [nCtg]                                                  */
    int32_t *nCntFlat;       /* sum(nTile), contig-major (file order)                   */
    int32_t **nCnt;          /* [nCtg] -> into nCntFlat                                 */
    int64_t *tBase;          /* byte offset in the .igd of every 64th tile (flat numbering): a per-tile table would be
                                1.5 MB of fresh pages at roadmap scale -- 0.75 ms of page faults at every open, as much as
                                the reference needs for its whole header -- for a number that is one short sum away   */
    void    *reserved_;
    char   **cName;          /* [nCtg], each a 40-byte buffer as stored in the file     */
    char   **fileName;       /* [nFiles]                                                */
    int32_t *fileNr;
    double  *fileMd;
    int64_t  nTileTotal, nRecords, dataOff;
    int32_t *dict;           /* open-addressing contig dictionary                        */
    int32_t  dictCap;
    igd_hip_db *dev;         /* the database resident on the GPU (NULL until attached)   */
    /* multi-GPU: the same database resident on further devices; dev == devs[0] */
    int32_t     ndev;
    igd_hip_db *devs[IGDC_MAX_DEVICES];
    igd_hip_group *grp;      /* the devices as one group: slabs + ONE RCCL all-reduce of hits[] (igd_hip_group_search) */
} igdc_db;

/* byte offset of tile j of contig c in the .igd (what the reference keeps as tIdx[c][j], src/igd_base.c:288-303) */
static inline int64_t igdc_tile_off(const igdc_db *db, int32_t c, int32_t j)
{
    const int64_t t = (db->nCnt[c] - db->nCntFlat) + j;
    int64_t n = 0;
    for (int64_t k = t & ~(int64_t)63; k < t; k++) n += db->nCntFlat[k];
    return db->tBase[t >> 6] + n * (db->gType == 0 ? 12 : 16);
}

/* header tables of an .igd (no tile data is read) */
igdc_db *igdc_open(const char *igd_path);
/* "<igd minus last .ext>_index.tsv" */
char    *igdc_index_path(const char *igd_path);
int      igdc_load_index(igdc_db *db, const char *tsv_path);
void     igdc_close(igdc_db *db);
int32_t  igdc_get_id(const igdc_db *db, const char *chrm);

/* Put the tile region on the GPU (igd_hip_open).  _path maps the file; _fp reads through an
 * already open stream (the CLI flavour's global fP).  Returns IGD_HIP_OK or an IGD_HIP_ERR_*. */
int igdc_attach_path(igdc_db *db, const char *igd_path, int device);
int igdc_attach_fp(igdc_db *db, FILE *fp, int device);
/* Multi-GPU, one process driving several devices: the database is replicated on
 * devices[0..n) (uploaded by n host threads at once; the same device may be listed twice), and
 * igdc_search_multi gives device r the r-th CONTIGUOUS slab of the batch (one host thread per
 * device), then adds the n per-device vectors of nFiles counts into hits[] -- the reference keeps one
 * hits[] for all queries and prints it once (src/igd_search.c:925,1032-1039); a sum of non-negative
 * integers, so any partition of the queries gives the identical vector.  The sum is the engine group's RCCL all-reduce
 * (ncclAllReduce of int64[nFiles] on every engine's stream, igd_hip_group_search); a host-side add of the n vectors only
 * where RCCL cannot be used (igd_hip.h).  "IGD_DEVICES=0,1,.." selects the devices for the command line tool. */
int igdc_attach_path_multi(igdc_db *db, const char *igd_path, const int *devices, int n);
int igdc_search_multi(igdc_db *db, const int32_t *ichr, const int32_t *qs, const int32_t *qe, int64_t nq,
                      int32_t v, int rule, int flags, int64_t *hits, int64_t *total);
/* "0,1,2" -> devices[]; returns the count (0 when the variable is unset or empty) */
int igdc_devices_from_env(int *devices, int max);

/* ONE query interval without the GPU.  The reference answers `-r` / get_overlaps() by reading the one to few tiles the
 * interval touches (fseek/fread of cnt records, src/igd_search.c:469-476) -- a few KB -- while putting the database on
 * the GPU means uploading all of it (851 MB at roadmap scale) for a single wave's work: for the single-interval entry
 * points of the three flavours (`igd search -r`, get_overlaps*, search_1) the host therefore reads those tiles itself
 * (pread) and counts.  Batches -- query files, search_n -- never come here: they have no CPU path.
 * Semantics = the engine's (and the reference's, src/igd_search.c:454-534 / :623-694 / :30-112): rule NEST or FLAT,
 * hits are the records with lob <= start < qe && end > qs [&& value >= v], lob = the tile's start in later tiles.
 * `emit` (may be NULL) is called per overlap in the reference's -f order (tiles ascending, record index descending).
 * Returns the number of overlaps, or -1 on an I/O error. */
/* one overlap: dataset, start, end, and where the record sits -- its index inside tile `tile` of the contig */
typedef void (*igdc_emit_fn)(void *ctx, int32_t idx, int32_t start, int32_t end, int32_t in_tile, int32_t tile);
int64_t igdc_walk_one(const igdc_db *db, int fd, int32_t ichr, int32_t qs, int32_t qe, int32_t v, int use_v, int rule,
                      int64_t *hits, igdc_emit_fn emit, void *ctx);

/* SMALL query files on the host (igd_hostpath.c; product code).  The reference starts cheaply -- header only, then the
 * tiles a query touches (src/igd_base.c:269-323, src/igd_search.c:469-476) -- while the engine costs a fixed ~0.18 s of HIP
 * start-up and upload: files of at most igdc_host_limit() queries (IGD_HOST_MAX_QUERIES; 0 = every file goes to the GPU) are
 * counted with pread() on the .igd (per-thread tile buffers) by a few host threads, the reference's per-query algorithm.  NOT a
 * fallback: the choice depends on the number of queries only, never on whether a device is usable; larger files have
 * no CPU path.  The engine's own entry points (igd_hip.h) never come here. */
typedef struct igdc_map igdc_map;
int64_t   igdc_host_limit(void);                            /* counting searches: 250 000 queries per usable thread, <= 4e6 */
int64_t   igdc_host_limit_enum(void);                       /* `-f`: 25 000 per usable thread */
int       igdc_host_probably_small(const char *qfile);       /* by file size: parse it before starting the engine */
igdc_map *igdc_map_open(const igdc_db *db, int fd);           /* fd stays the caller's */
void      igdc_map_close(igdc_map *m);
/* hits[] is ADDED to; *total = overlaps of the batch.  v = IGD_HIP_NO_VALUE_FILTER: no filter.  0 on success. */
int igdc_search_host(const igdc_db *db, const igdc_map *m, const int32_t *ichr, const int32_t *qs, const int32_t *qe,
                     int64_t nq, int32_t v, int rule, int64_t *hits, int64_t *total);
/* `-f` (rule NEST, the reference's order): qoff[0..nq] offsets, *out malloc'd (free()), entries as igd_hip_enumerate's */
int igdc_enumerate_host(const igdc_db *db, const igdc_map *m, const int32_t *ichr, const int32_t *qs, const int32_t *qe,
                        int64_t nq, int64_t *qoff, igd_hip_hit **out, int64_t *total);

/* handle flavours: small batch and no engine resident -> host; else the engine, attached (igdc_attach_path) when first needed.
 * Returns IGD_HIP_OK or the engine's error code. */
int igdc_search_auto(igdc_db *db, const char *path, int device, const int32_t *ichr, const int32_t *qs, const int32_t *qe,
                     int64_t nq, int32_t v, int rule, int flags, int64_t *hits, int64_t *total);

/* BED line -> (contig, start, end).  Mutates `line`.  require_chr=1 is the CLI rule
 * (name starts with "chr", shorter than 40, end > 0: src/igd_base.c:69); require_chr=0 is
 * the rule of the Python/R forks (>= 3 fields: src_py/igd_base.c:44). */
char *igdc_parse_bed(char *line, int32_t *st, int32_t *en, int require_chr);

/* '\n'-separated lines of a plain or gzip file */
typedef struct igdc_lines igdc_lines;
igdc_lines *igdc_lines_open(const char *path);
char       *igdc_lines_next(igdc_lines *r, int64_t *len);   /* NULL at end; buffer is reused */
void        igdc_lines_close(igdc_lines *r);

/* accepted queries of one file, as arrays */
typedef struct {
    int64_t n, cap;
    int32_t *ichr, *qs, *qe;
    int32_t unsorted;        /* 0 while every pushed query was >= the previous by (contig, start) */
    int32_t max_len;         /* largest qe - qs pushed (0 for an empty set): queries shorter than a tile let a dense sorted
                              * file take the engine's DIRECT step (IGD_HIP_FLAG_SHORT, verified on the device) */
} igdc_queries;
/* the engine flags a parsed query set earns: IGD_HIP_FLAG_SORTED when it is ordered, + IGD_HIP_FLAG_SHORT when no query is
 * as long as a tile of `nbp` bp */
int  igdc_queries_flags(const igdc_queries *q, int32_t nbp);
/* returns 0, or -1 when the file cannot be opened.  Lines whose contig is not in the
 * database are dropped here (the reference drops them in get_overlaps, :456-457). */
int  igdc_read_queries(const igdc_db *db, const char *qfile, int require_chr, igdc_queries *out);
void igdc_queries_free(igdc_queries *q);
/* every contig ONE run ordered by start, only the runs out of contig order (a sorted BED with another chromosome order
 * than the database's): put the runs into contig order, clear q->unsorted, return 1; otherwise 0.  Hits-only searches. */
int  igdc_queries_group_contigs(igdc_queries *q, int32_t nCtg);
int  igdc_queries_push(igdc_queries *q, int32_t ichr, int32_t qs, int32_t qe);

#ifdef __cplusplus
}
#endif
#endif
