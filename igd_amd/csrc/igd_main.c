/* igd_main.c -- `igd` command: `igd search ...` (the accelerated path, same flags and
 * stdout as /root/reference/src/igd.c:21-38 + src/igd_search.c:889-1079) and a minimal
 * `igd create` (format writer only).  The process-wide globals of the reference's igd.c
 * live in igd_cli_abi.c. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sysexits.h>
#include <unistd.h>

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

/* The command is done when its output is written: every stdio stream is flushed and the process ends without running
 * the HIP runtime's exit handlers -- freeing a gigabyte of device memory, unloading code objects and tearing the
 * context down takes as long as the whole search (60-90 ms of a 0.25 s command), and the kernel driver reclaims all of it
 * anyway.  (IGD_CLEAN_EXIT=1: leave through exit() as usual, e.g. under a leak checker.) */
int igd_hip_lazy_loaded(void);          /* libigd.so (igd_hip_lazy.c): has this process mapped the HIP engine? */
static int leave(int rc)
{
    /* a write that failed (full disk, closed pipe) must not look like success */
    if ((fflush(NULL) != 0 || ferror(stdout)) && rc == 0) rc = EX_IOERR;
    const char *e = getenv("IGD_CLEAN_EXIT");
    /* only a process that brought the HIP runtime up has those exit handlers to skip; everything else -- `-r`, small
     * query files, usage errors -- and any run under a profiler / sanitizer / coverage tool (IGD_CLEAN_EXIT=1: they
     * finalise in atexit) leaves through exit() as usual */
    if ((e && *e && *e != '0') || !igd_hip_lazy_loaded()) return rc;
    _exit(rc);
}

int main(int argc, char **argv)
{
    if (argc < 2) return usage(0);
    if (strcmp(argv[1], "search") == 0) return leave(igd_search(argc, argv));
    if (strcmp(argv[1], "create") == 0) return leave(igd_create(argc, argv));
    fprintf(stderr, "Unknown command\n");
    return usage(EX_USAGE);
}
