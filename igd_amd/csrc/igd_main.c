/* igd_main.c -- `igd` command: `igd search ...` (the accelerated path, same flags and
 * stdout as /root/reference/src/igd.c:21-38 + src/igd_search.c:889-1079) and a minimal
 * `igd create` (format writer only).  The process-wide globals of the reference's igd.c
 * live in igd_cli_abi.c. */
#include <stdio.h>
#include <string.h>
#include <sysexits.h>

#include "igd_search.h"
#include "igd_create_host.h"

static int usage(int code)
{
    fprintf(stderr,
            "igd (MI355X-native overlap search)\n"
            "usage:   igd <command> [options]\n"
            "         search    Search an igd database on the GPU\n"
            "         create    Create an igd database\n");
    return code;
}

int main(int argc, char **argv)
{
    if (argc < 2) return usage(0);
    if (strcmp(argv[1], "search") == 0) return igd_search(argc, argv);
    if (strcmp(argv[1], "create") == 0) return igd_create(argc, argv);
    fprintf(stderr, "Unknown command\n");
    return usage(EX_USAGE);
}
