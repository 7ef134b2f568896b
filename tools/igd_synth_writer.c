/* igd_synth_writer.c -- .igd FORMAT writer of the synthetic data generator (tools/, not product).
 *
 * Makes test and benchmark databases without a GPU: intervals -> tiles, each tile STABLE-sorted by
 * start (SURVEY.md App. A layout).  This is not `igd create`: the product's create runs on the GPU
 * (igd_amd/csrc/igd_create.{c,hip}) and reproduces the reference's tile order exactly; counts do
 * not depend on the order of equal starts, which is all the synthetic databases are used for.
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "igd_core.h"
#include "igd_synth_writer.h"

typedef struct { int32_t idx, start, end, value; } rec16;

static void stable_sort_by_start(rec16 *a, rec16 *tmp, int64_t n)
{
    /* bottom-up merge sort: stable, so equal starts keep source order */
    for (int64_t i = 1; i < n; i++) {                     /* short runs by insertion */
        if ((i & 15) == 0) continue;
        rec16 x = a[i];
        int64_t lo = i & ~(int64_t)15, j = i;
        while (j > lo && a[j - 1].start > x.start) { a[j] = a[j - 1]; j--; }
        a[j] = x;
    }
    rec16 *src = a, *dst = tmp;
    for (int64_t w = 16; w < n; w <<= 1) {
        for (int64_t lo = 0; lo < n; lo += 2 * w) {
            int64_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            int64_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi) dst[k++] = (src[j].start < src[i].start) ? src[j++] : src[i++];
            while (i < mid) dst[k++] = src[i++];
            while (j < hi) dst[k++] = src[j++];
        }
        rec16 *t = src; src = dst; dst = t;
    }
    if (src != a) memcpy(a, src, sizeof(rec16) * (size_t)n);
}

int igdc_write_igd(const char *igd_path, int32_t nbp, int32_t gType, int32_t nCtg,
                   const char *const *ctgNames, int64_t n, const igdc_interval *iv,
                   int32_t nFiles, const char *const *fileNames, const int32_t *nr,
                   const double *avg)
{
    if (nbp <= 0 || nCtg < 0 || (gType != 0 && gType != 1)) return -1;
    int32_t *nTile = (int32_t *)calloc((size_t)nCtg + 1, sizeof(int32_t));
    for (int64_t i = 0; i < n; i++) {
        const igdc_interval *x = &iv[i];
        if (x->start >= x->end || x->start <= -nbp || x->ctg < 0 || x->ctg >= nCtg) continue;
        int32_t n2 = (x->end - 1) / nbp;
        if (n2 + 1 > nTile[x->ctg]) nTile[x->ctg] = n2 + 1;
    }
    int64_t *base = (int64_t *)calloc((size_t)nCtg + 1, sizeof(int64_t));
    int64_t nT = 0;
    for (int32_t c = 0; c < nCtg; c++) {
        /* a contig that only received dropped intervals still exists with one empty tile
         * (the reference creates mTiles = 1 + n2 on first sight, src/igd_base.c:132-136) */
        if (nTile[c] == 0) nTile[c] = 1;
        base[c] = nT;
        nT += nTile[c];
    }
    int64_t *off = (int64_t *)calloc((size_t)nT + 1, sizeof(int64_t));
    for (int64_t i = 0; i < n; i++) {
        const igdc_interval *x = &iv[i];
        if (x->start >= x->end || x->start <= -nbp || x->ctg < 0 || x->ctg >= nCtg) continue;
        int32_t n1 = x->start / nbp, n2 = (x->end - 1) / nbp;
        for (int32_t j = n1; j <= n2; j++) off[base[x->ctg] + j + 1]++;
    }
    int32_t *cnt = (int32_t *)calloc((size_t)nT + 1, sizeof(int32_t));
    int64_t maxc = 0;
    for (int64_t t = 0; t < nT; t++) {
        cnt[t] = (int32_t)off[t + 1];
        if (off[t + 1] > maxc) maxc = off[t + 1];
        off[t + 1] += off[t];
    }
    const int64_t total = off[nT];
    rec16 *recs = (rec16 *)malloc(sizeof(rec16) * (size_t)(total ? total : 1));
    int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nT + 1));
    rec16 *tmp = (rec16 *)malloc(sizeof(rec16) * (size_t)(maxc ? maxc : 1));
    if (!recs || !cur || !tmp) { free(nTile); free(base); free(off); free(cnt); free(recs); free(cur); free(tmp); return -1; }
    memcpy(cur, off, sizeof(int64_t) * (size_t)(nT + 1));
    for (int64_t i = 0; i < n; i++) {
        const igdc_interval *x = &iv[i];
        if (x->start >= x->end || x->start <= -nbp || x->ctg < 0 || x->ctg >= nCtg) continue;
        int32_t n1 = x->start / nbp, n2 = (x->end - 1) / nbp;
        for (int32_t j = n1; j <= n2; j++) {
            rec16 *r = &recs[cur[base[x->ctg] + j]++];
            r->idx = x->file; r->start = x->start; r->end = x->end; r->value = x->value;
        }
    }
    for (int64_t t = 0; t < nT; t++)
        if (cnt[t] > 1) stable_sort_by_start(recs + off[t], tmp, cnt[t]);

    int rc = -1;
    FILE *fp = fopen(igd_path, "wb");
    if (fp) {
        int32_t head[3] = {nbp, gType, nCtg};
        fwrite(head, sizeof head, 1, fp);
        fwrite(nTile, sizeof(int32_t), (size_t)nCtg, fp);
        fwrite(cnt, sizeof(int32_t), (size_t)nT, fp);
        for (int32_t c = 0; c < nCtg; c++) {
            char name[40];
            memset(name, 0, sizeof name);
            strncpy(name, ctgNames[c], 39);
            fwrite(name, 40, 1, fp);
        }
        if (gType == 1)
            fwrite(recs, sizeof(rec16), (size_t)total, fp);
        else
            for (int64_t i = 0; i < total; i++) fwrite(&recs[i], 12, 1, fp);
        rc = (fflush(fp) == 0 && !ferror(fp)) ? 0 : -1;
        fclose(fp);
    }
    free(nTile); free(base); free(off); free(cnt); free(recs); free(cur); free(tmp);
    if (rc != 0) return rc;
    char *tsv = igdc_index_path(igd_path);
    fp = fopen(tsv, "w");
    free(tsv);
    if (!fp) return -1;
    fprintf(fp, "Index\tFile\tNumber of regions\tAvg size\n");
    for (int32_t i = 0; i < nFiles; i++)
        fprintf(fp, "%i\t%s\t%i\t%f\n", i, fileNames[i], nr ? nr[i] : 0, avg ? avg[i] : 0.0);
    fclose(fp);
    return 0;
}
