/* ingest_san_main.c -- harness for tests/test_sanitizers.py: the host query-ingest path
 * (igdc_open + igdc_load_index + igdc_read_queries, threaded and sequential) without the GPU.
 * usage: ingest_san <db.igd> <queries.bed[.gz]> <require_chr>   -> prints n and a checksum */
#include <stdio.h>
#include <stdlib.h>
#include "igd_core.h"

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    igdc_db *db = igdc_open(argv[1]);
    if (!db) return 3;
    char *tsv = igdc_index_path(argv[1]);
    igdc_load_index(db, tsv);
    free(tsv);
    igdc_queries q;
    if (igdc_read_queries(db, argv[2], atoi(argv[3]), &q) != 0) { igdc_close(db); return 4; }
    unsigned long long h = 1469598103934665603ULL;
    for (int64_t i = 0; i < q.n; i++) {
        h = (h ^ (unsigned)q.ichr[i]) * 1099511628211ULL;
        h = (h ^ (unsigned)q.qs[i]) * 1099511628211ULL;
        h = (h ^ (unsigned)q.qe[i]) * 1099511628211ULL;
    }
    printf("%lld %llu %d\n", (long long)q.n, h, q.unsorted);
    igdc_queries_free(&q);
    igdc_close(db);
    return 0;
}
