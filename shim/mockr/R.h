/* shim/mockr/R.h -- see Rinternals.h in this directory (a mock for tests, not R) */
#ifndef BSSM_AMD_MOCKR_R_H
#define BSSM_AMD_MOCKR_R_H
#include "Rinternals.h"
#endif
