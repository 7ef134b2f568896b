/* tests/mock_r/mock_r.c -- NOT R: the few functions behind tests/mock_r/Rinternals.h */
#include <setjmp.h>
#include <stdarg.h>
#include <stdio.h>
#include "Rinternals.h"

static struct mock_sexp nil_value = {0, 0, NULL, NULL, NULL};
SEXP R_NilValue = &nil_value;
int mock_protect_depth = 0;
char mock_last_error[256] = "";
jmp_buf mock_error_jmp;
int mock_error_armed = 0;

SEXP Rf_allocVector(unsigned type, long n)
{
    SEXP s = (SEXP)calloc(1, sizeof *s);
    s->type = (int)type; s->length = (int)n;
    s->data = calloc((size_t)(n > 0 ? n : 1), type == INTSXP ? sizeof(int) : sizeof(SEXP));
    return s;
}
SEXP Rf_mkChar(const char *str)
{
    SEXP s = (SEXP)calloc(1, sizeof *s);
    s->type = CHARSXP; s->length = (int)strlen(str); s->data = strdup(str);
    return s;
}
SEXP Rf_mkString(const char *str)
{
    SEXP s = Rf_allocVector(STRSXP, 1);
    SET_STRING_ELT(s, 0, Rf_mkChar(str));
    return s;
}
SEXP Rf_ScalarInteger(int v) { SEXP s = Rf_allocVector(INTSXP, 1); INTEGER(s)[0] = v; return s; }
SEXP Rf_install(const char *name) { return Rf_mkChar(name); }
SEXP R_MakeExternalPtr(void *p, SEXP tag, SEXP prot)
{
    (void)tag; (void)prot;
    SEXP s = (SEXP)calloc(1, sizeof *s);
    s->type = EXTPTRSXP; s->data = p;
    return s;
}
void *R_ExternalPtrAddr(SEXP s) { return s && s->type == EXTPTRSXP ? s->data : NULL; }
void R_SetExternalPtrAddr(SEXP s, void *p) { s->data = p; }
void R_RegisterCFinalizer(SEXP s, void (*fun)(SEXP)) { s->finalizer = fun; }
SEXP mock_make_class(const char *name) { return Rf_mkChar(name); }
SEXP mock_new_object(SEXP klass) { (void)klass; SEXP s = (SEXP)calloc(1, sizeof *s); s->type = S4SXP; return s; }
SEXP mock_set_slot(SEXP obj, SEXP name, SEXP value) { (void)name; obj->slot = value; return obj; }
SEXP mock_get_slot(SEXP obj, SEXP name) { (void)name; return obj->slot ? obj->slot : R_NilValue; }
void Rf_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(mock_last_error, sizeof mock_last_error, fmt, ap);
    va_end(ap);
    if (mock_error_armed) longjmp(mock_error_jmp, 1);   /* R unwinds to top level; the harness to its setjmp */
    fprintf(stderr, "mock Rf_error outside a guarded call: %s\n", mock_last_error);
    exit(3);
}
