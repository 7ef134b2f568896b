/* igd_create.h -- CLI/libigd flavour of the reference's src/igd_create.h:10-17 (same names and
 * prototypes), implemented over the GPU engine's igd_hip_create (include/igd_hip.h).
 *
 *   create_igd       src/igd_create.c:25-121    iPath = glob pattern of BED[.gz] files
 *   create_igd0      src/igd_create.c:246-343   same, 12-byte records (gType 0)
 *   create_igd_f     src/igd_create.c:124-243   iPath = text file listing BED files
 *   create_igd_bed4  src/igd_create.c:346-433   iPath = ONE BED4+ file, dataset name in column 4
 *   igd_create       src/igd_create.c:436-501   `igd create ...` argv driver (path fix-ups, options)
 * As in the reference: oPath must end in '/', the tile width is the global `tile_size`
 * (include/igd_base.h; igd_init reads it, src/igd_base.c:522), output goes to
 * <oPath><igdName>.igd and <oPath><igdName>_index.tsv, progress text to stdout.
 * There is no CPU path: without a usable GPU these print the reason and exit.
 */
#ifndef IGD_CREATE_ABI_H
#define IGD_CREATE_ABI_H
#include "igd_base.h"
#ifdef __cplusplus
extern "C" {
#endif
void create_igd(char *iPath, char *oPath, char *igdName);
void create_igd0(char *iPath, char *oPath, char *igdName);
void create_igd_f(char *iPath, char *oPath, char *igdName);
void create_igd_bed4(char *iPath, char *oPath, char *igdName);
int  igd_create(int argc, char **argv);
#ifdef __cplusplus
}
#endif
#endif
