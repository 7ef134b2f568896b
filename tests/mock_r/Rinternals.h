/* tests/mock_r/Rinternals.h -- NOT R.  A minimal stand-in for the handful of R C-API names igdr_abi.c's `.Call` entry
 * points use, so that those 110 lines are compiled and run by the test-suite in an image without R
 * (tests/test_host.py::test_r_call_entry_points_against_a_mock_of_the_r_api).  It is test infrastructure only: the
 * product's libigdr.so is never built against it (Makefile: -DIGDR_HAVE_R only when `R RHOME` works), and passing
 * against this mock does not show that the code works inside R -- object layout, garbage collection and S4 classes
 * are not modelled.  What it does show: the calls type-check against R's documented prototypes and do the right
 * arithmetic on their arguments. */
#ifndef IGD_MOCK_RINTERNALS_H
#define IGD_MOCK_RINTERNALS_H
#include <stdlib.h>
#include <string.h>

#define INTSXP 13
#define STRSXP 16
#define VECSXP 19
#define CHARSXP 9
#define EXTPTRSXP 22
#define S4SXP 25

typedef struct mock_sexp {
    int type, length;
    void *data;                 /* int[] / struct mock_sexp*[] / char[] / external pointer */
    struct mock_sexp *slot;     /* S4: the one slot this mock keeps; external pointer: nothing */
    void (*finalizer)(struct mock_sexp *);
} *SEXP;

extern SEXP R_NilValue;
extern int mock_protect_depth;      /* PROTECT / UNPROTECT balance, checked by the harness */
extern char mock_last_error[256];

SEXP Rf_allocVector(unsigned type, long n);
SEXP Rf_mkChar(const char *s);
SEXP Rf_mkString(const char *s);
SEXP Rf_ScalarInteger(int v);
SEXP Rf_install(const char *name);
SEXP R_MakeExternalPtr(void *p, SEXP tag, SEXP prot);
void *R_ExternalPtrAddr(SEXP s);
void R_SetExternalPtrAddr(SEXP s, void *p);
void R_RegisterCFinalizer(SEXP s, void (*fun)(SEXP));
void Rf_error(const char *fmt, ...);
SEXP mock_make_class(const char *name);
SEXP mock_new_object(SEXP klass);
SEXP mock_set_slot(SEXP obj, SEXP name, SEXP value);
SEXP mock_get_slot(SEXP obj, SEXP name);

#define allocVector Rf_allocVector
#define mkChar Rf_mkChar
#define mkString Rf_mkString
#define install Rf_install
#define error Rf_error
#define INTEGER(x) ((int *)(x)->data)
#define LENGTH(x) ((x)->length)
#define STRING_ELT(x, i) (((SEXP *)(x)->data)[i])
#define SET_STRING_ELT(x, i, v) (((SEXP *)(x)->data)[i] = (v))
#define VECTOR_ELT(x, i) (((SEXP *)(x)->data)[i])
#define SET_VECTOR_ELT(x, i, v) (((SEXP *)(x)->data)[i] = (v))
#define CHAR(x) ((const char *)(x)->data)
#define PROTECT(x) (mock_protect_depth++, (x))
#define UNPROTECT(n) (mock_protect_depth -= (n))
#endif
