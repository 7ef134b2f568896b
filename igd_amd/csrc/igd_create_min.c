/* igd_create_min.c -- a MINIMAL `igd create`: BED files -> .igd + _index.tsv.
 *
 * `igd create` is outside the accelerated path (offline, run once); this
 * exists only so that the CLI, the tests and the benchmark can make databases without the
 * reference binary -- which, besides, divides by n_files/10 and dies with fewer than 10 input
 * files (/root/reference/src/igd_create.c:48,81).  It follows the reference's default mode
 * (create_igd, src/igd_create.c:25-121): files in glob order, every line split on tabs,
 * start/end = columns 2,3 (atol), value = column 5 if present else 0, contigs in first-seen
 * order, no filtering by contig name; intervals with start>=end are dropped (igd_add,
 * src/igd_base.c:120) but still count in the per-file "Number of regions".
 */
#define _GNU_SOURCE
#include <glob.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include "igd_core.h"
#include "igd_create_min.h"

typedef struct { char **name; int32_t n, cap; } namelist;

static int32_t name_id(namelist *l, const char *s)
{
    for (int32_t i = l->n - 1; i >= 0; i--)      /* BED files are grouped by contig: the last */
        if (strcmp(l->name[i], s) == 0) return i;   /* few entries almost always match       */
    if (l->n == l->cap) {
        l->cap = l->cap ? 2 * l->cap : 64;
        l->name = (char **)realloc(l->name, sizeof(char *) * (size_t)l->cap);
    }
    l->name[l->n] = strdup(s);
    return l->n++;
}

int igdc_create_from_beds(const char *bed_glob, const char *out_dir, const char *name,
                          int32_t nbp, int32_t gType)
{
    glob_t g;
    if (glob(bed_glob, 0, NULL, &g) != 0 || g.gl_pathc == 0) {
        printf("wrong dir path: %s\n", bed_glob);
        return -1;
    }
    const int32_t nFiles = (int32_t)g.gl_pathc;
    namelist ctg = {0, 0, 0};
    igdc_interval *iv = NULL;
    int64_t n = 0, cap = 0;
    int32_t *nr = (int32_t *)calloc((size_t)nFiles, sizeof(int32_t));
    double *avg = (double *)calloc((size_t)nFiles, sizeof(double));
    char **fnames = (char **)calloc((size_t)nFiles, sizeof(char *));
    /* last-contig cache: consecutive lines usually share the contig */
    char last[64] = "";
    int32_t lastId = -1;
    for (int32_t f = 0; f < nFiles; f++) {
        const char *slash = strrchr(g.gl_pathv[f], '/');
        fnames[f] = strdup(slash ? slash + 1 : g.gl_pathv[f]);
        igdc_lines *r = igdc_lines_open(g.gl_pathv[f]);
        if (!r) continue;
        char *line;
        double sum = 0;
        while ((line = igdc_lines_next(r, NULL)) != NULL) {
            char *col[5];
            int nc = 0;
            col[nc++] = line;
            for (char *p = line; *p && nc < 5; ++p)
                if (*p == '\t') { *p = '\0'; col[nc++] = p + 1; }
            if (nc < 3) continue;
            int32_t st = (int32_t)atol(col[1]), en = (int32_t)atol(col[2]);
            int32_t va = nc > 4 ? (int32_t)atol(col[4]) : 0;
            nr[f]++;
            sum += (double)en - (double)st;
            if (st >= en) continue;       /* igd_add returns before it even registers the contig */
            int32_t c;
            if (lastId >= 0 && strcmp(last, col[0]) == 0) c = lastId;
            else {
                c = name_id(&ctg, col[0]);
                strncpy(last, col[0], sizeof last - 1);
                last[sizeof last - 1] = '\0';
                lastId = strlen(col[0]) < sizeof last ? c : -1;
            }
            if (n == cap) {
                cap = cap ? 2 * cap : (1 << 16);
                iv = (igdc_interval *)realloc(iv, sizeof(igdc_interval) * (size_t)cap);
            }
            iv[n].file = f; iv[n].ctg = c; iv[n].start = st; iv[n].end = en; iv[n].value = va;
            n++;
        }
        igdc_lines_close(r);
        avg[f] = nr[f] ? sum / nr[f] : 0.0;
    }
    size_t L = strlen(out_dir) + strlen(name) + 8;
    char *path = (char *)malloc(L);
    mkdir(out_dir, 0777);
    snprintf(path, L, "%s%s%s.igd", out_dir, out_dir[strlen(out_dir) - 1] == '/' ? "" : "/", name);
    int rc = igdc_write_igd(path, nbp, gType, ctg.n, (const char *const *)ctg.name, n, iv, nFiles,
                            (const char *const *)fnames, nr, avg);
    if (rc == 0) printf("Save igd database to %s\n", path);
    free(path);
    for (int32_t i = 0; i < ctg.n; i++) free(ctg.name[i]);
    free(ctg.name);
    for (int32_t f = 0; f < nFiles; f++) free(fnames[f]);
    free(fnames); free(nr); free(avg); free(iv);
    globfree(&g);
    return rc;
}

/* `igd create <input dir or glob> <output dir> <name> [-b 11..19] [-s 0|1]` */
int igd_create_min(int argc, char **argv)
{
    if (argc < 5) {
        fprintf(stderr, "usage: igd create <input dir | \"glob\"> <output dir> <igd name> [-b <11..19>] [-s 0]\n"
                        "       (minimal writer of the MI355X build; see `igd search`)\n");
        return 0;
    }
    int32_t nbp = 16384, gType = 1;
    for (int i = 5; i < argc; i++) {
        if (strcmp(argv[i], "-b") == 0 && i + 1 < argc) {
            int b = atoi(argv[i + 1]);
            if (b > 10 && b < 20) nbp = 1 << b;
        }
        if (strcmp(argv[i], "-s") == 0 && i + 1 < argc && atoi(argv[i + 1]) == 0) gType = 0;
    }
    size_t L = strlen(argv[2]);
    char *pat = (char *)malloc(L + 4);
    strcpy(pat, argv[2]);
    if (L && pat[L - 1] == '/') strcat(pat, "*");
    else if (L && pat[L - 1] != '*') strcat(pat, "/*");
    int rc = igdc_create_from_beds(pat, argv[3], argv[4], nbp, gType);
    free(pat);
    return rc == 0 ? 0 : 1;
}
