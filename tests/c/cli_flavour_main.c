/* A caller written against the reference's CLI-flavour API exactly as the reference's own
 * src/igd.c + igd_search() use it: it DEFINES the process-wide globals itself (src/igd.c:14-19),
 * calls get_igdinfo / get_fileinfo, opens fP, then the search functions -- and is compiled
 * against include/igd_search.h and linked with -ligd.  Used by tests/test_gpu_golden.py.
 *
 *   cli_flavour_main <db.igd> <queries.bed> <chr> <start> <end> <v>
 */
#include <stdlib.h>
#include <string.h>
#include "igd_search.h"

void *hc;
iGD_t *IGD;
gdata_t *gData = NULL;
gdata0_t *gData0 = NULL;
int32_t preIdx, preChr, tile_size;
FILE *fP;

static void dump(const char *tag, long long ret, const int64_t *hits, int n)
{
    printf("%s ret=%lld hits=", tag, ret);
    for (int i = 0; i < n; i++) printf("%lld%s", (long long)hits[i], i + 1 < n ? "," : "");
    printf("\n");
}

int main(int argc, char **argv)
{
    if (argc < 7) return 2;
    IGD = get_igdinfo(argv[1]);
    if (!IGD) return 3;
    char tsv[4096];
    strcpy(tsv, argv[1]);
    *strrchr(tsv, '.') = '\0';
    strcat(tsv, "_index.tsv");
    IGD->finfo = get_fileinfo(tsv, &IGD->nFiles);
    const int n = IGD->nFiles;
    int64_t *hits = calloc((size_t)n, sizeof(int64_t));
    fP = fopen(argv[1], "rb");
    int32_t s = atoi(argv[4]), e = atoi(argv[5]), v = atoi(argv[6]);

    printf("nbp=%d gType=%d nCtg=%d nFiles=%d id(%s)=%d id(nope)=%d\n", IGD->nbp, IGD->gType, IGD->nCtg, n, argv[3],
           get_id(argv[3]), get_id("nope"));
    dump("getOverlaps", getOverlaps(argv[2], hits), hits, n);
    dump("getOverlaps(again,accumulates)", getOverlaps(argv[2], hits), hits, n);
    memset(hits, 0, sizeof(int64_t) * (size_t)n);
    if (IGD->gType == 1) {
        dump("getOverlaps_v", getOverlaps_v(argv[2], hits, v), hits, n);
        memset(hits, 0, sizeof(int64_t) * (size_t)n);
        dump("get_overlaps_v", get_overlaps_v(argv[3], s, e, v, hits), hits, n);
        memset(hits, 0, sizeof(int64_t) * (size_t)n);
        dump("get_overlaps", get_overlaps(argv[3], s, e, hits), hits, n);
        printf("get_overlaps_f1 ret=%d\n", get_overlaps_f1(argv[3], s, e));
    } else {
        dump("getOverlaps0", getOverlaps0(argv[2], hits), hits, n);
        memset(hits, 0, sizeof(int64_t) * (size_t)n);
        dump("get_overlaps0", get_overlaps0(argv[3], s, e, hits), hits, n);
        printf("get_overlaps_f0 ret=%d\n", get_overlaps_f0(argv[3], s, e));
    }
    dump("unknown contig", get_overlaps("chrNope", s, e, hits), hits, 0);
    printf("getOverlaps(missing file)=%lld\n", (long long)getOverlaps("/nonexistent/q.bed", hits));
    char line[] = "chr1\t12\t34\tx";
    int32_t a, b;
    char *c = parse_bed(line, &a, &b);
    printf("parse_bed -> %s %d %d\n", c ? c : "(null)", a, b);
    return 0;
}
