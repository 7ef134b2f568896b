/* igd_create_min.h -- minimal BED -> .igd writer front (see igd_create_min.c). */
#ifndef IGD_CREATE_MIN_H
#define IGD_CREATE_MIN_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
int igdc_create_from_beds(const char *bed_glob, const char *out_dir, const char *name,
                          int32_t nbp, int32_t gType);
int igd_create_min(int argc, char **argv);
#ifdef __cplusplus
}
#endif
#endif
