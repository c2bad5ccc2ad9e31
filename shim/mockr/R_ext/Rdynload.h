/* shim/mockr/R_ext/Rdynload.h -- see ../Rinternals.h (a mock for tests, not R) */
#ifndef BSSM_AMD_MOCKR_RDYNLOAD_H
#define BSSM_AMD_MOCKR_RDYNLOAD_H
#include "../Rinternals.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef void* (*DL_FUNC)();
typedef struct { const char* name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef struct _DllInfo { int unused; } DllInfo;
int R_registerRoutines(DllInfo*, const void* c_routines, const R_CallMethodDef* call_routines, const void* fortran, const void* external);
Rboolean R_useDynamicSymbols(DllInfo*, Rboolean);
Rboolean R_forceSymbols(DllInfo*, Rboolean);
#ifdef __cplusplus
}
#endif
#endif
