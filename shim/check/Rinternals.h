/* Declarations of the subset of R's C API that shim/bssm_amd_shim.c uses -- for `gcc -fsyntax-only` in images without
 * R (this one).  NOT R: nothing here is ever linked or run; where R exists the real <Rinternals.h> is used instead
 * (the package build does not have shim/check on its include path). */
#ifndef BSSM_AMD_SHIM_CHECK_RINTERNALS_H
#define BSSM_AMD_SHIM_CHECK_RINTERNALS_H
#include <stddef.h>
typedef struct SEXPREC* SEXP;
typedef ptrdiff_t R_xlen_t;
typedef int Rboolean;
#define NILSXP 0
#define LGLSXP 10
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
#define VECSXP 19
extern SEXP R_NilValue, R_NamesSymbol, R_DimSymbol, R_GlobalEnv;
extern double R_NaReal, R_NegInf;
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
SEXP Rf_mkString(const char*);
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
const char* CHAR(SEXP);
char* R_alloc(size_t, int);
void Rf_error(const char*, ...) __attribute__((noreturn));
void Rf_warning(const char*, ...);
void R_CheckUserInterrupt(void);
#endif
