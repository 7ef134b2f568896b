/* igd_synth_writer.h -- see igd_synth_writer.c */
#ifndef IGD_SYNTH_WRITER_H
#define IGD_SYNTH_WRITER_H
#include <stdint.h>
typedef struct { int32_t file, ctg, start, end, value; } igdc_interval;
/* Intervals in source order; start>=end is dropped like igd_add does (src/igd_base.c:120), and so is
 * start<=-nbp (negative tile index there).  Each interval is copied to every tile
 * start/nbp..(end-1)/nbp; tiles are stable-sorted by start.  Writes <igd_path> and its _index.tsv. */
int igdc_write_igd(const char *igd_path, int32_t nbp, int32_t gType, int32_t nCtg,
                   const char *const *ctgNames, int64_t n, const igdc_interval *iv,
                   int32_t nFiles, const char *const *fileNames, const int32_t *nr,
                   const double *avg);
#endif
