/* igd_search.h -- CLI/libigd flavour of the overlap-search API: same names, argument
 * meaning, return values and error behaviour as /root/reference/src/igd_search.h:15-41,
 * executed on an MI355X through include/igd_hip.h.
 *
 * Contract kept from the reference:
 *   - preconditions are the globals of igd_base.h: IGD/hc set by get_igdinfo(), fP opened on
 *     the .igd by the caller (src/igd_search.c:974);
 *   - `hits` is caller-allocated (nFiles int64), caller-zeroed, and is ADDED to;
 *   - unknown contig, tile past the end, unopenable query file: silent, returns 0;
 *   - get_overlaps/getOverlaps(/0) return 0 (the reference never increments its counter
 *     there, src/igd_search.c:533), get_overlaps_v/getOverlaps_v return the overlap count;
 *   - not thread-safe (process-wide state), like the reference.
 * New, and the only observable differences:
 *   - query FILES (getOverlaps*, igd_search -q) run on the GPU: the first such call uploads the
 *     whole tile region once (instead of fseek/fread per tile, :469-476).  Batches have no CPU
 *     path: without a usable HIP device the call prints the reason on stderr and returns like
 *     the reference's silent failures (0 / nothing added); igd_engine_status() keeps the code.
 *     No library function ever ends the host process;
 *   - single intervals (get_overlaps*, igd_search -r) are answered on the host from the
 *     interval's own tiles, exactly as the reference reads them, while no engine is resident.
 * GPU selection: environment variables IGD_DEVICE (default 0) / IGD_DEVICES=0,1,.. (query slabs).
 *
 * Seqpare: seqOverlaps (src/igd_search.c:354-451) runs on the GPU -- the matching needs all queries at
 * once; its per-query helper seq_overlaps (:253-352, appends ONE interval's overlaps and similarities to
 * the caller's overlaps_t) is a single interval and answered on the host like get_overlaps.
 */
#ifndef __IGD_SEARCH_H__
#define __IGD_SEARCH_H__
#include "igd_base.h"

#ifdef __cplusplus
extern "C" {
#endif

/* one query -------------------------------------------------------------------------- */
int32_t get_overlaps  (char *chrm, int32_t qs, int32_t qe, int64_t *hits);             /* :454-534 */
int32_t get_overlaps_v(char *chrm, int32_t qs, int32_t qe, int32_t v, int64_t *hits);  /* :623-694 */
int32_t get_overlaps0 (char *chrm, int32_t qs, int32_t qe, int64_t *hits);             /* :30-112  */
int32_t get_overlaps_f1(char *chrm, int32_t qs, int32_t qe);                           /* :537-620 */
int32_t get_overlaps_f0(char *chrm, int32_t qs, int32_t qe);                           /* :114-200 */

/* a BED / BED.gz file of queries ----------------------------------------------------- */
int64_t getOverlaps  (char *qFile, int64_t *hits);                                     /* :696-719 */
int64_t getOverlaps_v(char *qFile, int64_t *hits, int32_t v);                          /* :746-769 */
int64_t getOverlaps0 (char *qFile, int64_t *hits);                                     /* :202-225 */
int64_t getOverlaps_f1(char *qFile);                                                   /* :721-744 */
int64_t getOverlaps_f0(char *qFile);                                                   /* :227-250 */

/* dataset x dataset hit map (`-m`): hitmap[nFiles][nFiles], caller-zeroed, incremented; returns
 * the number of pairs counted */
int64_t getMap(uint32_t **hitmap);                                                     /* :772-826 */
int64_t getMap_v(uint32_t **hitmap, int32_t v);                                        /* :829-886 */

/* `-s`: Seqpare similarity of the query file with every dataset; sm[nFiles] --------- */
void seq_overlaps(char *chrm, int32_t qs, int32_t qe, overlaps_t *olp);                 /* :253-352 */
void seqOverlaps(char *qFile, double *sm);                                             /* :354-451 */

/* `igd search <db.igd> [-q file | -r chr s e | -m] [-v N] [-f] [-o name] [-c]`        :889-1079 */
int igd_search(int argc, char **argv);

/* Not in the reference.  The library never ends the host process: when the GPU engine cannot be used
 * (no device, out of memory, a batch beyond its limits) the failing call says why on stderr and returns
 * like the reference's silent failures (0 / hits[] untouched, src/igd_search.c:457,462,701-702);
 * this returns the engine's code of the FIRST such failure (0: none).  igd_search() returns non-zero
 * instead of printing a table of zeros. */
int igd_engine_status(void);

#ifdef __cplusplus
}
#endif
#endif
