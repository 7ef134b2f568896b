/* igd_oracle_create.c -- CPU ORACLE for `igd create`.
 *
 * >>> TEST INFRASTRUCTURE, NOT PRODUCT CODE (see igd_oracle.h). <<<
 *
 * Restates, sequentially and in plain C, what the reference does between reading BED text and
 * writing `<name>.igd` + `<name>_index.tsv`:
 *   reading loops   /root/reference/src/igd_create.c:25-121 (default), :124-243 (-f list),
 *                   :246-343 (-s 0), :346-433 (-s 2, one BED4+ file)
 *   str_splits      /root/reference/src/igd_base.c:37-51   (incl. its creeping column limit)
 *   parse_bed       /root/reference/src/igd_base.c:53-72   (only the -f mode uses it)
 *   igd_add         /root/reference/src/igd_base.c:118-169 (drop start>=end, replicate into
 *                   tiles start/nbp .. (end-1)/nbp, contigs in first-seen order)
 *   igd_saveT/save  /root/reference/src/igd_base.c:333-364, :396-461 (per-tile append in input
 *                   order, then radix_sort_intv per tile, header + tiles written contig-major)
 *   radix sort      /root/reference/src/igd_base.h:196-249 (Heng Li's klib/cgranges radix sort:
 *                   MSD byte-wise "American flag" in-place permutation, insertion sort for
 *                   buckets of <= 64; UNSTABLE, so the order of records with equal start is a
 *                   property of this exact algorithm -- restated here step for step so that the
 *                   tile bytes come out identical)
 *
 * Pinned by tests/test_oracle_create.py: byte comparison with files made by oracle/_ref/igd
 * (`create`, `create -s 0`, `create -f`, `create -s 2`) on random inputs, and by the committed
 * fixture tests/golden/create/.
 *
 * Deliberate deviations (reference UB, not behaviour):
 *   - the 40-byte contig-name fields of the header: the reference writes 40 bytes starting at a
 *     strdup'd string (src/igd_base.c:420), i.e. heap garbage after the NUL; zero-filled here.
 *   - lines with fewer than 3 columns make the reference read stale pointers (str_splits leaves
 *     splits[1..] of the previous line); skipped here.
 *   - negative start (tile index < 0, src/igd_base.c:126,160) is dropped.
 *   - the -f mode passes an uninitialised `va` (src/igd_create.c:166,188); 0 here.
 *   - no temporary data0/ files: tiles are kept in memory in the same (input) order.
 *   - fewer than 10 input files: the reference divides by n_files/10 (src/igd_create.c:48,81);
 *     no progress dots are printed in that case.
 */
#define _GNU_SOURCE
#include <glob.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <zlib.h>

#include "igd_oracle.h"

#define ORC_MAXCOUNT 268435456LL      /* src/igd_base.h:37 */

typedef struct { int32_t ctg, file, start, end, value; } crec;
typedef struct { int32_t key, src; } kv_t;

typedef struct {
    crec *r; int64_t n, cap;
    char **ctg; int32_t nctg, mctg; int32_t *mTiles;
    int32_t nbp;
    int64_t batch_total;                 /* igd->total: replicas since the last igd_saveT      */
} cstate;

static int32_t ctg_id(cstate *S, const char *name, int *absent)
{
    for (int32_t i = 0; i < S->nctg; i++)
        if (strcmp(S->ctg[i], name) == 0) { *absent = 0; return i; }
    if (S->nctg == S->mctg) {
        S->mctg = S->mctg ? 2 * S->mctg : 32;
        S->ctg = (char **)realloc(S->ctg, sizeof(char *) * (size_t)S->mctg);
        S->mTiles = (int32_t *)realloc(S->mTiles, sizeof(int32_t) * (size_t)S->mctg);
    }
    S->ctg[S->nctg] = strdup(name);
    S->mTiles[S->nctg] = 0;
    *absent = 1;
    return S->nctg++;
}

/* igd_add, src/igd_base.c:118-169 */
static void add_interval(cstate *S, const char *chrm, int32_t s, int32_t e, int32_t v, int32_t idx)
{
    if (s >= e) return;
    if (s < 0) return;                                   /* deviation: reference UB            */
    int absent;
    const int32_t c = ctg_id(S, chrm, &absent);
    const int32_t n1 = s / S->nbp, n2 = (e - 1) / S->nbp;
    if (absent) S->mTiles[c] = 1 + n2;                    /* :131 */
    if (n2 + 1 >= S->mTiles[c]) S->mTiles[c] = n2 + 1;    /* :145-147 */
    if (S->n == S->cap) {
        S->cap = S->cap ? 2 * S->cap : 4096;
        S->r = (crec *)realloc(S->r, sizeof(crec) * (size_t)S->cap);
    }
    crec *r = &S->r[S->n++];
    r->ctg = c; r->file = idx; r->start = s; r->end = e; r->value = v;
    S->batch_total += n2 - n1 + 1;                        /* igd->total++ per replica, :166 */
}

/* igd_saveT's report line, src/igd_base.c:362-363 */
static void batch_report(cstate *S, FILE *out)
{
    int64_t nt = 0;
    for (int32_t i = 0; i < S->nctg; i++) nt += S->mTiles[i];
    if (out) fprintf(out, "nCtgs, nRegions, nTiles: %i\t %lld\t %lld\n", S->nctg, (long long)S->batch_total, (long long)nt);
    S->batch_total = 0;
}

/* str_splits, src/igd_base.c:37-51: returns the number of fields it produced; *nmax follows the
 * reference's recurrence (it is overwritten with that number on every call). */
static int split_tabs(char *str, int *nmax, char **f, int fcap)
{
    int ns = 1;
    f[0] = str;
    char *ch = str;
    do {
        if (*ch == '\t') {
            if (ns < fcap) f[ns] = ch + 1;
            ns++;
            *ch = '\0';
        }
        ch++;
    } while (*ch != '\0' && ns < *nmax + 1);
    *nmax = ns;
    return ns;
}

/* ---- the sort: src/igd_base.h:196-249 ------------------------------------------------------- */
static inline int digit_of(int32_t key, int shift) { return (key >> shift) & 255; }

static void small_sort(kv_t *a, int64_t n)                /* rs_insertsort: stable, signed compare */
{
    for (int64_t i = 1; i < n; i++)
        if (a[i].key < a[i - 1].key) {
            kv_t t = a[i];
            int64_t j = i;
            for (; j > 0 && t.key < a[j - 1].key; j--) a[j] = a[j - 1];
            a[j] = t;
        }
}

static void flag_sort(kv_t *a, int64_t n, int shift)      /* rs_sort */
{
    int64_t lo[256], hi[256], first[256];
    for (int k = 0; k < 256; k++) hi[k] = 0;
    for (int64_t i = 0; i < n; i++) hi[digit_of(a[i].key, shift)]++;
    int64_t acc = 0;
    for (int k = 0; k < 256; k++) { lo[k] = first[k] = acc; acc += hi[k]; hi[k] = acc; }
    for (int k = 0; k < 256;) {
        if (lo[k] == hi[k]) { k++; continue; }
        int d = digit_of(a[lo[k]].key, shift);
        if (d == k) { lo[k]++; continue; }
        kv_t carry = a[lo[k]];
        do {                                              /* follow the displacement cycle      */
            kv_t t = a[lo[d]];
            a[lo[d]++] = carry;
            carry = t;
            d = digit_of(carry.key, shift);
        } while (d != k);
        a[lo[k]++] = carry;
    }
    if (shift) {
        const int ns = shift > 8 ? shift - 8 : 0;
        for (int k = 0; k < 256; k++) {
            const int64_t sz = hi[k] - first[k];
            if (sz > 64) flag_sort(a + first[k], sz, ns);
            else if (sz > 1) small_sort(a + first[k], sz);
        }
    }
}

void orc_tile_sort(int32_t *key, int32_t *src, int64_t n)  /* radix_sort_intv on (key, payload)   */
{
    kv_t *a = (kv_t *)malloc(sizeof(kv_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; i++) { a[i].key = key[i]; a[i].src = src[i]; }
    if (n <= 64) small_sort(a, n);
    else flag_sort(a, n, 24);
    for (int64_t i = 0; i < n; i++) { key[i] = a[i].key; src[i] = a[i].src; }
    free(a);
}

/* ---- igd_save: header + sorted tiles, src/igd_base.c:396-461 (gType 0: :463-516) ------------- */
static int save_igd(cstate *S, const char *opath, const char *name, int32_t gType)
{
    char path[2048];
    snprintf(path, sizeof path, "%s%s.igd", opath, name);
    FILE *fp = fopen(path, "wb");
    if (!fp) { printf("Can't open file %s", path); return -1; }
    int64_t nT = 0;
    int64_t *tbase = (int64_t *)malloc(sizeof(int64_t) * (size_t)(S->nctg + 1));
    for (int32_t c = 0; c < S->nctg; c++) { tbase[c] = nT; nT += S->mTiles[c]; }
    tbase[S->nctg] = nT;
    int32_t *cnt = (int32_t *)calloc((size_t)(nT > 0 ? nT : 1), sizeof(int32_t));
    for (int64_t i = 0; i < S->n; i++) {
        const crec *r = &S->r[i];
        for (int32_t j = r->start / S->nbp; j <= (r->end - 1) / S->nbp; j++) cnt[tbase[r->ctg] + j]++;
    }
    int64_t *off = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nT + 1));
    off[0] = 0;
    for (int64_t t = 0; t < nT; t++) off[t + 1] = off[t] + cnt[t];
    const int64_t R = off[nT];
    kv_t *a = (kv_t *)malloc(sizeof(kv_t) * (size_t)(R > 0 ? R : 1));
    int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nT > 0 ? nT : 1));
    memcpy(fill, off, sizeof(int64_t) * (size_t)nT);
    for (int64_t i = 0; i < S->n; i++) {                  /* input order = temp-file append order */
        const crec *r = &S->r[i];
        for (int32_t j = r->start / S->nbp; j <= (r->end - 1) / S->nbp; j++) {
            kv_t *p = &a[fill[tbase[r->ctg] + j]++];
            p->key = r->start; p->src = (int32_t)i;
        }
    }
    fwrite(&S->nbp, 4, 1, fp);
    fwrite(&gType, 4, 1, fp);
    fwrite(&S->nctg, 4, 1, fp);
    fwrite(S->mTiles, 4, (size_t)S->nctg, fp);
    fwrite(cnt, 4, (size_t)nT, fp);
    for (int32_t c = 0; c < S->nctg; c++) {
        char nm[40];
        memset(nm, 0, sizeof nm);
        strncpy(nm, S->ctg[c], 39);
        fwrite(nm, 40, 1, fp);
    }
    for (int64_t t = 0; t < nT; t++) {
        const int64_t n = cnt[t];
        if (n == 0) continue;
        kv_t *g = a + off[t];
        if (n <= 64) small_sort(g, n);
        else flag_sort(g, n, 24);
        for (int64_t i = 0; i < n; i++) {
            const crec *r = &S->r[g[i].src];
            int32_t rec[4] = { r->file, r->start, r->end, r->value };
            fwrite(rec, 4, gType == 0 ? 3 : 4, fp);
        }
    }
    fclose(fp);
    free(tbase); free(cnt); free(off); free(a); free(fill);
    return 0;
}

/* _index.tsv, src/igd_create.c:93-110 */
static void save_index(const char *opath, const char *name, char **files, int32_t nf,
                       const int32_t *nr, const double *avg, int64_t *nT, double *l_avg)
{
    char path[2048];
    snprintf(path, sizeof path, "%s%s_index.tsv", opath, name);
    FILE *fpi = fopen(path, "w");
    if (!fpi) { printf("Can't open file %s", path); return; }
    fprintf(fpi, "Index\tFile\tNumber of regions\tAvg size\n");
    *nT = 0; *l_avg = 0.0;
    for (int32_t i = 0; i < nf; i++) {
        const char *t = strrchr(files[i], '/');
        t = t ? t + 1 : files[i];
        *nT += nr[i];
        *l_avg += avg[i];
        fprintf(fpi, "%i\t%s\t%i\t%f\n", i, t, nr[i], avg[i] / nr[i]);
    }
    fclose(fpi);
}

static void cstate_free(cstate *S)
{
    for (int32_t i = 0; i < S->nctg; i++) free(S->ctg[i]);
    free(S->ctg); free(S->mTiles); free(S->r);
}

/* mode: ORC_CREATE_GLOB (default), ORC_CREATE_LIST (-f), ORC_CREATE_GTYPE0 (-s 0), ORC_CREATE_BED4 (-s 2).
 * ipath: the glob pattern (already ending in '*' as igd_create leaves it, src/igd_create.c:466-473),
 * or the list / BED4 file; opath ends in '/'.  Returns 0, or -1 if nothing was written. */
int orc_create(const char *ipath, const char *opath, const char *name, int32_t nbp, int mode, FILE *out)
{
    cstate S;
    memset(&S, 0, sizeof S);
    S.nbp = nbp;
    char **files = NULL;
    int32_t nf = 0;
    glob_t g;
    int globbed = 0;
    char line[1024];
    mkdir(opath, 0777);

    if (mode == ORC_CREATE_BED4) {                        /* src/igd_create.c:346-433 */
        if (out) fprintf(out, "igd_create 1\n");
        int nCols = 32, fcap = 0;
        char *f[40];
        int32_t *nr = NULL; double *avg = NULL;
        gzFile z = gzopen(ipath, "r");
        if (!z) return -1;
        int64_t nL = 0; int j = 0;
        while (gzgets(z, line, 1024) != NULL) {
            const int nc = split_tabs(line, &nCols, f, 40);
            if (nc < 5) continue;                         /* deviation: stale pointers in the reference */
            int32_t idx = -1;
            for (int32_t i = 0; i < nf; i++) if (strcmp(files[i], f[3]) == 0) { idx = i; break; }
            if (idx < 0) {
                if (nf == fcap) {
                    fcap = fcap ? 2 * fcap : 64;
                    files = (char **)realloc(files, sizeof(char *) * (size_t)fcap);
                    nr = (int32_t *)realloc(nr, sizeof(int32_t) * (size_t)fcap);
                    avg = (double *)realloc(avg, sizeof(double) * (size_t)fcap);
                }
                files[nf] = strdup(f[3]); nr[nf] = 0; avg[nf] = 0.0;
                idx = nf++;
            }
            const int32_t st = (int32_t)atol(f[1]), en = (int32_t)atol(f[2]);
            add_interval(&S, f[0], st, en, (int32_t)atol(f[4]), idx);
            nr[idx]++;
            avg[idx] += en - st;
            nL++;
            if (S.batch_total >= ORC_MAXCOUNT) {
                j++;
                if (out) fprintf(out, "--igd_saveT1--%i, %lld\n", j, (long long)S.batch_total);
                batch_report(&S, out);
                nL = 0;
            }
        }
        gzclose(z);
        if (nL > 0) batch_report(&S, out);
        if (out) fprintf(out, "igd_create 2\n");
        int64_t nT; double l_avg;
        save_index(opath, name, files, nf, nr, avg, &nT, &l_avg);
        if (out) fprintf(out, "igd_create 3\n");
        save_igd(&S, opath, name, 1);
        if (out) {
            fprintf(out, "igd_create 4\n");
            fprintf(out, "Total intervals, l_avg:  %lld %12.3f\n", (long long)nT, l_avg / nT);
        }
        for (int32_t i = 0; i < nf; i++) free(files[i]);
        free(files); free(nr); free(avg);
        cstate_free(&S);
        return 0;
    }

    if (mode == ORC_CREATE_LIST) {                        /* src/igd_create.c:124-163 */
        if (out) fprintf(out, "Create igd from %s: \n", ipath);
        FILE *fl = fopen(ipath, "r");
        if (!fl) { if (out) fprintf(out, "Can't open file %s", ipath); return -1; }
        char buf[1024];
        int cap = 0;
        while (fgets(buf, 1024, fl) != NULL) {
            buf[strcspn(buf, "\n")] = 0;
            gzFile z = gzopen(buf, "r");
            if (!z) continue;
            line[0] = 0;
            gzgets(z, line, 1024);
            int32_t st = 0, en = 0;
            if (orc_parse_bed(line, &st, &en)) {
                if (nf == cap) { cap = cap ? 2 * cap : 64; files = (char **)realloc(files, sizeof(char *) * (size_t)cap); }
                files[nf++] = strdup(buf);
            }
            gzclose(z);
        }
        fclose(fl);
        if (nf < 1) { if (out) fprintf(out, "Too few files (add to path /*): %i\n", nf); return -1; }
    } else {                                              /* src/igd_create.c:31-43, :252-264 */
        if (mode == ORC_CREATE_GTYPE0) { if (out) fprintf(out, "igd_create 0\n"); }
        else if (out) fprintf(out, "Create igd from %s: \n", ipath);
        if (glob(ipath, 0, NULL, &g) != 0) {
            if (out) fprintf(out, mode == ORC_CREATE_GTYPE0 ? "wrong dir path: %s" : "wrong dir path: %s\n", ipath);
            return -1;
        }
        globbed = 1;
        files = g.gl_pathv;
        nf = (int32_t)g.gl_pathc;
        if (mode == ORC_CREATE_GTYPE0 && out) fprintf(out, "igd_create 1: %i\n", nf);
    }

    int32_t *nr = (int32_t *)calloc((size_t)nf, sizeof(int32_t));
    double *avg = (double *)calloc((size_t)nf, sizeof(double));
    const int32_t nf10 = nf / 10;
    const int bufsz = mode == ORC_CREATE_GTYPE0 ? 256 : 1024;   /* src/igd_create.c:267 */
    int nCols = 16;
    char *f[24];
    int rc = 0;
    for (int32_t ig = 0; ig < nf && rc == 0; ig++) {
        gzFile z = gzopen(files[ig], "r");
        if (!z) { rc = -1; break; }                       /* the reference returns, writing nothing */
        while (gzgets(z, line, bufsz) != NULL) {
            if (mode == ORC_CREATE_LIST) {                /* :187-192 */
                int32_t st = 0, en = 0;
                char *ctg = orc_parse_bed(line, &st, &en);
                if (ctg && st >= 0 && en < 321000000) {
                    add_interval(&S, ctg, st, en, 0, ig);
                    nr[ig]++;
                    avg[ig] += en - st;
                }
            } else {                                      /* :66-72, :287-292 */
                const int nc = split_tabs(line, &nCols, f, 24);
                if (nc < 3) continue;                     /* deviation, see header */
                const int32_t st = (int32_t)atol(f[1]), en = (int32_t)atol(f[2]);
                int32_t va = 0;
                if (mode == ORC_CREATE_GLOB && nCols > 4) va = (int32_t)atol(f[4]);
                add_interval(&S, f[0], st, en, va, ig);
                nr[ig]++;
                avg[ig] += en - st;
            }
            if (S.batch_total > ORC_MAXCOUNT) batch_report(&S, out);   /* batch boundary, :73-77 + :84 */
        }
        gzclose(z);
        if (mode != ORC_CREATE_GTYPE0 && nf10 > 0 && (ig + 1) % nf10 == 0 && out) fprintf(out, ".");   /* :81 */
    }
    if (rc == 0) {
        if (mode == ORC_CREATE_GTYPE0) S.batch_total = 0;   /* igd0_saveT prints nothing */
        else batch_report(&S, out);
        if (mode != ORC_CREATE_GTYPE0 && out) fprintf(out, "\n");
        int64_t nT; double l_avg;
        save_index(opath, name, files, nf, nr, avg, &nT, &l_avg);
        if (mode == ORC_CREATE_GTYPE0 && out) fprintf(out, "igd_create 3\n");
        save_igd(&S, opath, name, mode == ORC_CREATE_GTYPE0 ? 0 : 1);
        if (out) {
            if (mode == ORC_CREATE_GTYPE0) fprintf(out, "igd_create 4\n");
            else fprintf(out, "Save igd database to %s%s.igd\n", opath, name);
            fprintf(out, "Total intervals, l_avg:  %lld %12.3f\n", (long long)nT, l_avg / nT);
        }
    }
    free(nr); free(avg);
    if (globbed) globfree(&g);
    else { for (int32_t i = 0; i < nf; i++) free(files[i]); free(files); }
    cstate_free(&S);
    return rc;
}

/* `igd create <in> <out> <name> [-b n] [-s 0|1|2] [-f]`, src/igd_create.c:436-501 */
int orc_igd_create(int argc, char **argv, FILE *out)
{
    if (argc < 5) return 0;
    char ipath[1024], opath[1024], ftmp[2200];
    strcpy(ipath, argv[2]);
    strcpy(opath, argv[3]);
    const char *dbname = argv[4];
    int dtype = 1, ftype = 0;
    int32_t nbp = 16384;
    for (int i = 5; i < argc; i++) {
        if (strcmp(argv[i], "-s") == 0 && i + 1 < argc) dtype = atoi(argv[i + 1]);
        if (strcmp(argv[i], "-b") == 0 && i + 1 < argc) {
            const int n = atoi(argv[i + 1]);
            if (n > 10 && n < 20) nbp = 1 << n;
        }
        if (strcmp(argv[i], "-f") == 0) ftype = 1;
    }
    if (opath[strlen(opath) - 1] != '/') strcat(opath, "/");
    if (ftype == 0 && dtype != 2) {
        if (ipath[strlen(ipath) - 1] == '/') strcat(ipath, "*");
        else if (ipath[strlen(ipath) - 1] != '*') strcat(ipath, "/*");
    }
    struct stat st;
    snprintf(ftmp, sizeof ftmp, "%s%s.igd", opath, dbname);
    if (stat(ftmp, &st) == 0) {
        if (out) fprintf(out, "The igd database file %s exists!\n", ftmp);
        return 0;
    }
    const int mode = dtype == 0 ? ORC_CREATE_GTYPE0 : dtype == 2 ? ORC_CREATE_BED4 : ftype == 1 ? ORC_CREATE_LIST : ORC_CREATE_GLOB;
    orc_create(ipath, opath, dbname, nbp, mode, out);
    return 0;
}
