/* igd_py_abi.h -- the handle-based flavour of the search API that the reference's Cython
 * wrapper binds (/root/reference/src_py/igd_py.pyx:6-19; C side src_py/igd_base.h,
 * src_py/igd_search.h:14-17, src_py/igd_create.h).  Same names and prototypes, so the
 * unchanged igd_py.pyx compiles and links against libigd_py.so; `iGD_t` is opaque to the
 * wrapper (pyx:7-8), which leaves its layout free.
 *
 * Behaviour kept from src_py:
 *   - query lines are accepted with >= 3 tab fields, ANY contig name (src_py/igd_base.c:44);
 *   - rule NEST only, gType 1 only, no value filter (src_py/igd_search.c:25-102);
 *   - hits: caller-owned int64[nFiles], caller-zeroed, ADDED to;
 *   - getOverlaps returns sum(hits[0..nFiles)) after the search (src_py/igd_search.c:124-127).
 * Differences: close_iGD on a never-opened handle is safe (the reference frees garbage);
 * the search runs on the GPU selected by IGD_DEVICE; without a usable GPU the process stops.
 */
#ifndef IGD_PY_ABI_H
#define IGD_PY_ABI_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct iGD_t iGD_t;

iGD_t  *iGD_init(void);                                   /* src_py/igd_base.c:341-348 */
int32_t get_nFiles(iGD_t *iGD);
void    open_iGD(iGD_t *iGD, char *igdFile);              /* src_py/igd_base.c:161-222 */
void    close_iGD(iGD_t *iGD);                            /* src_py/igd_base.c:350-366 */
/* src_py/igd_create.c:18-143.  The reference appends "/" and "*" to the CALLER's buffers (:22-31),
 * which the Cython wrapper passes as immutable bytes objects; here the paths are copied first and
 * iPath/oPath are only read */
void    create_iGD(iGD_t *iGD, char *iPath, char *oPath, char *igdName, int tile_size);
void    get_overlaps(iGD_t *iGD, char *chrm, int32_t qs, int32_t qe, int64_t *hits);  /* src_py/igd_search.c:25-102 */
int64_t getOverlaps(iGD_t *iGD, char *qFile, int64_t *hits);                          /* src_py/igd_search.c:104-128 */

/* Not in the reference.  This library never ends the interpreter: when the GPU engine cannot be used
 * the failing call says why on stderr and returns like the reference's silent failures (handle left
 * closed / hits[] untouched / 0); igd_engine_status() returns the engine's code of the last such
 * failure (0: none; igd_hip_last_error() has the text), igd_engine_clear() resets it. */
int  igd_engine_status(void);
void igd_engine_clear(void);

#ifdef __cplusplus
}
#endif
#endif
