/* igd_create_host.h -- host side of `igd create`; see igd_create.c. */
#ifndef IGD_CREATE_HOST_H
#define IGD_CREATE_HOST_H
#include <stdint.h>
#include <stdio.h>
#ifdef __cplusplus
extern "C" {
#endif

/* which reading loop of the reference is followed */
enum {
    IGDC_CREATE_GLOB   = 0,   /* create_igd       src/igd_create.c:25-121  (default)                */
    IGDC_CREATE_LIST   = 1,   /* create_igd_f     src/igd_create.c:124-243 (-f: text file of paths) */
    IGDC_CREATE_GTYPE0 = 2,   /* create_igd0      src/igd_create.c:246-343 (-s 0: 12-byte records)  */
    IGDC_CREATE_BED4   = 3    /* create_igd_bed4  src/igd_create.c:346-433 (-s 2: one file, dataset
                                                                            name in column 4)      */
};
/* whose progress text is printed */
enum { IGDC_MSG_CLI = 0, IGDC_MSG_PY = 1, IGDC_MSG_QUIET = 2, IGDC_MSG_R = 3 };

typedef struct {
    const char *ipath;        /* glob pattern (ends in '*'), list file, or BED4 file                 */
    const char *opath;        /* output directory, ends in '/'                                       */
    const char *name;         /* database name: <opath><name>.igd, <opath><name>_index.tsv           */
    int32_t nbp;              /* tile width                                                          */
    int mode, msg;
    int linebuf;              /* gzgets buffer of the flavour: 1024 (CLI), 256 (CLI -s 0, Python, R)   */
    int device;               /* GPU                                                                 */
} igdc_create_opts;

/* 0; 1 when nothing was written (bad path / unreadable file: the reference returns silently);
 * < 0: an IGD_HIP_ERR_* of the engine (no GPU: there is no CPU path). */
int igdc_create(const igdc_create_opts *o);

/* `igd create <in> <out> <name> [-b 11..19] [-s 0|1|2] [-f]`, src/igd_create.c:436-501 */
int igd_create(int argc, char **argv);

#ifdef __cplusplus
}
#endif
#endif
