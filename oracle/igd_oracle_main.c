/* igd_oracle_main.c -- command-line front of the CPU ORACLE (test infrastructure only).
 *   igd_oracle search <db.igd> -q <bed[.gz]> [-v N] [-f]      same stdout as the reference
 *   igd_oracle search <db.igd> -r chr start end [-v N] [-f]
 *   igd_oracle stats  <db.igd> -q <bed[.gz]> [-v N]            work statistics
 */
#include "igd_oracle.h"
#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv)
{
    if (argc >= 2 && strcmp(argv[1], "search") == 0)
        return orc_igd_search(argc, argv, stdout);
    if (argc >= 2 && strcmp(argv[1], "create") == 0)
        return orc_igd_create(argc, argv, stdout);
    if (argc >= 5 && strcmp(argv[1], "stats") == 0) {
        orc_db *db = orc_open(argv[2]);
        if (!db) { fprintf(stderr, "cannot open %s\n", argv[2]); return 1; }
        const char *q = NULL; int32_t v = 0;
        for (int i = 3; i < argc; i++) {
            if (!strcmp(argv[i], "-q") && i + 1 < argc) q = argv[i + 1];
            if (!strcmp(argv[i], "-v") && i + 1 < argc) v = atoi(argv[i + 1]);
        }
        int32_t *c, *s, *e;
        int64_t n = orc_read_queries(db, q, &c, &s, &e);
        if (n < 0) { fprintf(stderr, "cannot open %s\n", q); return 1; }
        int64_t *hits = (int64_t *)calloc((size_t)orc_nfiles(db) + 1, 8);
        orc_reset_stats(db);
        int64_t tot = orc_search_batch(db, c, s, e, n, v, hits);
        const orc_stats *st = orc_get_stats(db);
        printf("{\"queries\": %lld, \"pairs\": %lld, \"S\": %lld, \"H\": %lld, \"B\": %lld, \"total\": %lld}\n",
               (long long)n, (long long)st->pairs, (long long)st->S, (long long)st->H,
               (long long)st->B, (long long)tot);
        free(hits); free(c); free(s); free(e);
        orc_close(db);
        return 0;
    }
    fprintf(stderr, "usage: igd_oracle search|stats <db.igd> [options]\n");
    return 0;
}
