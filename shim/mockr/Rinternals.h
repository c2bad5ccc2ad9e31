/* shim/mockr/Rinternals.h -- a small MOCK of the part of R's C API that shim/bssm_amd_shim.c uses, so that the shim can be
 * compiled, linked against libbssm_amd.so and EXECUTED by the tests on machines without R (this image, the GPU box).
 *
 * This is a test double, not R: vectors (REALSXP, INTSXP, LGLSXP, STRSXP, VECSXP), CHARSXP, symbols, attributes (names,
 * dim, anything installed), language objects of the form f(x) whose function is a C callback the test supplies, R_alloc,
 * Rf_error as a longjmp back into mockr_call, Rf_warning recorded -- and a PROTECT stack that is CHECKED: every allocation
 * behaves like a garbage collection under gctorture(TRUE), i.e. every object of the current call that is neither on the
 * protect stack nor reachable from it (nor an argument of the call) is poisoned on the spot, and touching a poisoned object
 * fails the call.  A .Call entry point that returns with its protect stack unbalanced fails too (R: "stack imbalance").
 * Where R exists the real <Rinternals.h> is used; nothing of this directory is part of the product. */
#ifndef BSSM_AMD_MOCKR_RINTERNALS_H
#define BSSM_AMD_MOCKR_RINTERNALS_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct SEXPREC* SEXP;
typedef ptrdiff_t R_xlen_t;
typedef int Rboolean;
#define NILSXP 0
#define SYMSXP 1
#define CLOSXP 3
#define LANGSXP 6
#define CHARSXP 9
#define LGLSXP 10
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
#define VECSXP 19
#define EXTPTRSXP 22
extern SEXP R_NilValue, R_NamesSymbol, R_DimSymbol, R_GlobalEnv;
extern double R_NaReal, R_NegInf, R_PosInf;
extern int R_NaInt;
#define NA_INTEGER R_NaInt
#define NA_LOGICAL R_NaInt
#define NA_REAL R_NaReal
SEXP Rf_getAttrib(SEXP, SEXP);
SEXP Rf_setAttrib(SEXP, SEXP, SEXP);
SEXP Rf_install(const char*);
SEXP Rf_allocVector(unsigned int, R_xlen_t);
SEXP Rf_allocMatrix(unsigned int, int, int);
SEXP Rf_alloc3DArray(unsigned int, int, int, int);
SEXP Rf_allocArray(unsigned int, SEXP);
SEXP Rf_mkNamed(unsigned int, const char**);
SEXP Rf_ScalarReal(double);
SEXP Rf_ScalarInteger(int);
SEXP Rf_ScalarLogical(int);
SEXP Rf_mkString(const char*);
SEXP Rf_mkChar(const char*);
SEXP Rf_duplicate(SEXP);
SEXP Rf_coerceVector(SEXP, unsigned int);
SEXP Rf_lang2(SEXP, SEXP);
SEXP Rf_eval(SEXP, SEXP);
SEXP Rf_protect(SEXP);
void Rf_unprotect(int);
#define PROTECT(s) Rf_protect(s)
#define UNPROTECT(n) Rf_unprotect(n)
int Rf_length(SEXP);
R_xlen_t Rf_xlength(SEXP);
int Rf_asInteger(SEXP);
int Rf_asLogical(SEXP);
double Rf_asReal(SEXP);
int Rf_isNull(SEXP);
int Rf_isFunction(SEXP);
int TYPEOF(SEXP);
double* REAL(SEXP);
int* INTEGER(SEXP);
int* LOGICAL(SEXP);
SEXP VECTOR_ELT(SEXP, R_xlen_t);
SEXP SET_VECTOR_ELT(SEXP, R_xlen_t, SEXP);
SEXP STRING_ELT(SEXP, R_xlen_t);
void SET_STRING_ELT(SEXP, R_xlen_t, SEXP);
const char* CHAR(SEXP);
char* R_alloc(size_t, int);
void Rf_error(const char*, ...) __attribute__((noreturn, format(printf, 1, 2)));
void Rf_warning(const char*, ...) __attribute__((format(printf, 1, 2)));
void R_CheckUserInterrupt(void);

/* ---- the test harness's side of the mock (not R API) ---- */
typedef SEXP (*mockr_closure_fn)(SEXP arg, void* user);
SEXP mockr_closure(mockr_closure_fn fn, void* user);          /* an R function object whose body is a C callback */
SEXP mockr_extptr(void);                                      /* an external pointer (only its attributes are read) */
/* run a .Call entry point: fn(args[0], ..., args[nargs-1]); NULL when it raised (message: mockr_last_error) */
SEXP mockr_call(void* fn, int nargs, SEXP* args);
const char* mockr_last_error(void);
const char* mockr_last_warning(void);
int mockr_protect_depth(void);       /* after the last mockr_call returned (0 = balanced) */
int mockr_poisoned_touches(void);    /* uses of collected objects seen so far */
long mockr_live_objects(void);
void mockr_gctorture(int on);        /* default on */
void mockr_reset(void);              /* free every object and R_alloc block; call between tests */
void* mockr_lookup(const char* name, int* nargs);             /* the routines registered through R_registerRoutines */
int mockr_registered_count(void);
int mockr_dynamic_symbols(void);     /* last value passed to R_useDynamicSymbols (-1: never called) */
#ifdef __cplusplus
}
#endif
#endif
