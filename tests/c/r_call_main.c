/* tests/c/r_call_main.c -- every `.Call` entry point of the R flavour (igdr_abi.c, -DIGDR_HAVE_R), compiled against the
 * MOCK of R's C API in tests/mock_r/ (not R: see Rinternals.h there) and called the way IGDr/R/IGDr.R:26-158 calls them;
 * results are compared with the plain-C entry points of the same library.
 *   r_call_main <db.igd> [gpu]     -- `gpu`: also search_nr (one GPU batch)
 * Prints "R-CALL-OK" and exits 0 when everything agrees. */
#include <setjmp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "igdr_abi.h"
#include <Rdefines.h>

extern jmp_buf mock_error_jmp;
extern int mock_error_armed;
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "r_call_main: check failed at line %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    const int gpu = argc > 2 && !strcmp(argv[2], "gpu");
    SEXP obj = iGD_new(Rf_mkString(argv[1]));
    CHECK(obj && obj->type == S4SXP && obj->slot && obj->slot->type == EXTPTRSXP && obj->slot->finalizer);
    CHECK(mock_protect_depth == 0);
    SEXP ref = GET_SLOT(obj, Rf_install("ref"));
    iGD_t *h = open_iGD(argv[1]);
    CHECK(h != NULL);
    const int nfiles = INTEGER(get_nfiles(ref))[0], nctg = INTEGER(get_nCtgs(ref))[0], nbp = INTEGER(get_nbp(ref))[0];
    CHECK(nfiles > 0 && nctg > 0 && nbp > 0 && mock_protect_depth == 0);
    /* get_cid: known and unknown contigs, equal to the plain get_id */
    CHECK(INTEGER(get_cid(ref, Rf_mkString("chr1")))[0] == get_id(h, "chr1"));
    CHECK(INTEGER(get_cid(ref, Rf_mkString("chrNope")))[0] == -1);
    /* search_1r against get_overlaps32 */
    const char *chr[4] = {"chr1", "chr2", "chrNope", "chr1"};
    const int qs[4] = {1000, 100000, 5, 70000}, qe[4] = {90000, 260000, 50, 70000};
    int32_t *want = (int32_t *)calloc((size_t)nfiles, sizeof(int32_t));
    for (int k = 0; k < 4; k++) {
        SEXP r = search_1r(ref, Rf_mkString(chr[k]), Rf_ScalarInteger(qs[k]), Rf_ScalarInteger(qe[k]));
        CHECK(r->type == INTSXP && LENGTH(r) == nfiles && mock_protect_depth == 0);
        memset(want, 0, (size_t)nfiles * sizeof(int32_t));
        get_overlaps32(h, (char *)chr[k], qs[k], qe[k], want);
        CHECK(memcmp(INTEGER(r), want, (size_t)nfiles * sizeof(int32_t)) == 0);
    }
    /* get_binLen / get_binData: 1-based contig and bin; out of range -> NULL */
    CHECK(get_binLen(ref, Rf_ScalarInteger(nctg + 1), Rf_ScalarInteger(1)) == R_NilValue);
    CHECK(get_binData(ref, Rf_ScalarInteger(0), Rf_ScalarInteger(1)) == R_NilValue);
    long records = 0;
    for (int b = 1; b <= 40; b++) {
        SEXP len = get_binLen(ref, Rf_ScalarInteger(1), Rf_ScalarInteger(b));
        if (len == R_NilValue) break;
        const int n = INTEGER(len)[0];
        SEXP dat = get_binData(ref, Rf_ScalarInteger(1), Rf_ScalarInteger(b));
        if (n < 1) { CHECK(dat == R_NilValue); continue; }
        CHECK(dat->type == VECSXP && LENGTH(dat) == 3 && LENGTH(VECTOR_ELT(dat, 0)) == n);
        for (int k = 0; k < n; k++) {
            CHECK(INTEGER(VECTOR_ELT(dat, 0))[k] >= 0 && INTEGER(VECTOR_ELT(dat, 0))[k] < nfiles);      /* idx */
            CHECK(INTEGER(VECTOR_ELT(dat, 1))[k] < INTEGER(VECTOR_ELT(dat, 2))[k]);                      /* start < end */
            if (k) CHECK(INTEGER(VECTOR_ELT(dat, 1))[k - 1] <= INTEGER(VECTOR_ELT(dat, 1))[k]);          /* sorted by start */
        }
        records += n;
        CHECK(mock_protect_depth == 0);
    }
    CHECK(records > 0);
    if (gpu) {                                            /* search_nr: n intervals in one batch == the sum of search_1r */
        SEXP names = Rf_allocVector(STRSXP, 4), s = Rf_allocVector(INTSXP, 4), e = Rf_allocVector(INTSXP, 4);
        memset(want, 0, (size_t)nfiles * sizeof(int32_t));
        for (int k = 0; k < 4; k++) {
            SET_STRING_ELT(names, k, Rf_mkChar(chr[k])); INTEGER(s)[k] = qs[k]; INTEGER(e)[k] = qe[k];
            get_overlaps32(h, (char *)chr[k], qs[k], qe[k], want);
        }
        SEXP r = search_nr(ref, Rf_ScalarInteger(4), names, s, e);
        CHECK(r->type == INTSXP && LENGTH(r) == nfiles && mock_protect_depth == 0);
        CHECK(memcmp(INTEGER(r), want, (size_t)nfiles * sizeof(int32_t)) == 0);
    }
    /* iGD_free, then the finalizer on the freed pointer (a no-op), then a call on it: an R error, not a crash */
    CHECK(iGD_free(ref) == R_NilValue && R_ExternalPtrAddr(ref) == NULL);
    ref->finalizer(ref);
    mock_error_armed = 1;
    if (setjmp(mock_error_jmp) == 0) {
        (void)get_nbp(ref);
        fprintf(stderr, "r_call_main: a call on a freed handle did not raise\n");
        return 1;
    }
    mock_error_armed = 0;
    close_iGD(h);
    free(want);
    printf("R-CALL-OK nfiles=%d nctg=%d nbp=%d records=%ld%s\n", nfiles, nctg, nbp, records, gpu ? " search_nr" : "");
    return 0;
}
