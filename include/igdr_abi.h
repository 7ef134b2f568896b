/* igdr_abi.h -- the flavour of the search API that the reference's R package binds
 * (/root/reference/IGDr/src/igd_search.h:50-58, IGDr/src/igd_base.h, IGDr/src/igd_create.h:11;
 * R side: IGDr/R/IGDr.R:26-158 via .Call, IGDr/R/create.R:35,48 via .C).
 *
 * Two groups of entry points:
 *   (1) plain-C and `.C` ones -- pointers to R vectors, no R headers needed: always built
 *       and tested here through ctypes;
 *   (2) `.Call` ones taking/returning SEXP -- compiled only with -DIGDR_HAVE_R (R's headers);
 *       this build container has no R, so group (2) is source-complete but UNVERIFIED.
 * R integer vectors are 32-bit, hence get_overlaps32 (IGDr/src/igd_search.c:105-186): counts
 * are computed in int64 on the GPU and narrowed on the way out.
 * Query lines / contig names: any name, >= 3 fields (IGDr/src/igd_base.c:45).  Rule NEST.
 */
#ifndef IGDR_ABI_H
#define IGDR_ABI_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct iGD_t iGD_t;

iGD_t *open_iGD(char *igdFile);                                   /* IGDr/src/igd_base.c:162-223 */
void   close_iGD(iGD_t *iGD);
int32_t get_id(iGD_t *iGD, const char *chrm);
void   get_overlaps  (iGD_t *iGD, char *chrm, int32_t qs, int32_t qe, int64_t *hits);  /* igd_search.c:25-103  */
void   get_overlaps32(iGD_t *iGD, char *chrm, int32_t qs, int32_t qe, int32_t *hits);  /* igd_search.c:105-186 */

/* `.C` entry points: every argument is a pointer to an R vector */
void search_1(char **igdFile, char **qchr, int32_t *qs, int32_t *qe, int64_t *hits);   /* :189-194 */
void getOverlaps(char **igdFile, char **qFile, int64_t *hits);                         /* :196-217 */
void create_iGD(char **iPath, char **oPath, char **igdName, int *binsize);             /* igd_create.c */
void create_iGD_f(char **iPath, char **oPath, char **igdName, int *binsize);

/* batch form used by search_nr (IGDr/src/igd_search.c:340-355): n queries by contig NAME,
 * counted into 32-bit hits in ONE GPU batch */
void igdr_search_n32(iGD_t *iGD, int32_t n, const char *const *chrm, const int32_t *qs,
                     const int32_t *qe, int32_t *hits);

#ifdef IGDR_HAVE_R
#include <Rinternals.h>
SEXP iGD_new(SEXP igd_file);
SEXP iGD_free(SEXP igdr);
SEXP search_1r(SEXP igdr, SEXP qchrm, SEXP qs, SEXP qe);
SEXP search_nr(SEXP igdr, SEXP n, SEXP qchrm, SEXP qs, SEXP qe);
SEXP get_cid(SEXP igdr, SEXP chrom);
SEXP get_nbp(SEXP igdr);
SEXP get_nfiles(SEXP igdr);
SEXP get_nCtgs(SEXP igdr);
SEXP get_binLen(SEXP igdr, SEXP ichr, SEXP bin);
SEXP get_binData(SEXP igdr, SEXP ichr, SEXP bin);
#endif

/* Not in the reference.  This library never ends the R session: when the GPU engine cannot be used the
 * failing call says why on stderr and returns like the reference's silent failures (NULL handle /
 * hits untouched); the .Call entry points raise an R error instead.  igd_engine_status(): the engine's
 * code of the last such failure (0: none), igd_engine_clear() resets it. */
int  igd_engine_status(void);
void igd_engine_clear(void);

#ifdef __cplusplus
}
#endif
#endif
