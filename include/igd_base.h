/* igd_base.h -- CLI/libigd flavour: the shared types and process-wide state that the
 * reference declares in /root/reference/src/igd_base.h, restricted to what the overlap
 * search path uses.  A program written against the reference's header (src/igd.c,
 * src/igd_search.c's callers) compiles against this one unchanged for that path.
 *
 * Only the LAYOUT of the public structs and the NAMES/PROTOTYPES of the functions are
 * shared with the reference (that is the ABI); the implementation behind them is
 * igd_amd/csrc/igd_cli_abi.c over the HIP engine (include/igd_hip.h).
 *
 * Not provided (outside the hot path): the create-side types
 * (igd_t, ctg_t, tile_t, ...), igd_add/igd_save, the AIList/Seqpare helpers.
 */
#ifndef __IGD_BASE_H__
#define __IGD_BASE_H__
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One tile record as stored in a gType-1 .igd: 16 bytes, idx first
 * (reference: src/igd_base.h:41-46). */
typedef struct {
    int32_t idx;      /* source dataset = row of <db>_index.tsv */
    int32_t start;    /* 0-based, half-open [start,end)          */
    int32_t end;
    int32_t value;    /* BED column 5 (signal), 0 if absent      */
} gdata_t;

/* gType-0 record, 12 bytes (reference: src/igd_base.h:48-52). */
typedef struct {
    int32_t idx;
    int32_t start;
    int32_t end;
} gdata0_t;

/* One row of <db>_index.tsv (reference: src/igd_base.h:54-58). */
typedef struct {
    char   *fileName;
    int32_t nr;       /* number of regions in the dataset        */
    double  md;       /* mean region width (fraction truncated)  */
} info_t;

/* Header tables of an open .igd (reference: src/igd_base.h:96-105). */
typedef struct {
    int32_t   nFiles;
    info_t   *finfo;
    char      fname[64];
    int32_t   nbp, gType, nCtg;
    char    **cName;
    int32_t  *nTile;
    int32_t **nCnt;
    int64_t **tIdx;   /* byte offset of every tile in the file   */
} iGD_t;

/* One overlap as Seqpare records it, and a growing list of them (reference: src/igd_base.h:108-119). */
typedef struct {
    int32_t idx_t;    /* FIRST tile of the query (also for records met in later tiles, src/igd_search.c:291,:337) */
    int32_t idx_g;    /* index of the record inside its tile      */
    int32_t idx_f;    /* dataset of the record                    */
    float   sm;       /* similarity overlap / (|q| + |r| - overlap), single precision */
} overlap_t;

typedef struct {
    int32_t    nn, mm;   /* entries used / allocated               */
    overlap_t *olist;    /* realloc'ed by seq_overlaps              */
} overlaps_t;

/* Process-wide state of the CLI flavour (reference: src/igd_base.h:135-140; defined in
 * src/igd.c:14-19, here in igd_cli_abi.c so that a plain `-ligd` link works, and a program
 * that defines them itself still links: ELF resolves to the executable's copy). */
extern void     *hc;        /* contig-name dictionary built by get_igdinfo        */
extern iGD_t    *IGD;       /* the open database                                  */
extern gdata_t  *gData;     /* reference's one-tile cache; kept NULL here          */
extern gdata0_t *gData0;
extern int32_t   preIdx, preChr, tile_size;
extern FILE     *fP;        /* the .igd, opened by the caller before searching     */

/* src/igd_base.c:53-72 */
char    *parse_bed(char *s, int32_t *st_, int32_t *en_);
/* src/igd_base.c:74-94: last index in [t0,tc] with start < qe, -1 if none (host helper) */
int32_t  bSearch(gdata_t *gdata, int32_t t0, int32_t tc, int32_t qe);
/* src/igd_base.c:325-331 */
int32_t  get_id(const char *chrm);
/* src/igd_base.c:235-267 */
info_t  *get_fileinfo(char *ifName, int32_t *nFiles);
/* src/igd_base.c:269-323: also (re)builds `hc` */
iGD_t   *get_igdinfo(char *igdFile);

#ifdef __cplusplus
}
#endif
#endif
