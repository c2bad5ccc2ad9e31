// user_model_tu.hip -- translation unit of a run-time compiled user model.
//
// bssm's ssm_nlg / ssm_sde models are user C++ compiled at run time (Rcpp::sourceCpp) and called through function
// pointers (src/model_ssm_nlg.h:10-21, src/model_ssm_sde.h:8-11).  Device code cannot call host pointers, so here the
// user's snippet of __device__ functions is compiled INTO the filter kernels: bssm_user_model_build() (api.cpp) runs
//   hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -I<csrc> -DBSSM_USER_M=<pad(m)> -DBSSM_USER_NSTATES=<m> -DBSSM_USER_SNIPPET='"<file>"'
//         user_model_tu.hip -o libbssm_user_<hash>.so
// and registers the two launchers below.  The snippet's contract is documented in pf_functors.h.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#define BSSM_USER_MODEL 1
#define BSSM_FN __host__ __device__ inline   // every user function is needed on the device (kernels) and on the host (psi / EKF set-up)
// what the reference passes besides (alpha, theta, known_params): the time index and the known_tv_params matrix
// (src/model_ssm_nlg.h:10-21).  A snippet that defines BSSM_USER_TV takes it as the first argument of every function.
struct bssm_user_tv { int t; const double* known_tv; int rows, cols; };
#include BSSM_USER_SNIPPET

#ifndef BSSM_USER_M
#error "BSSM_USER_M (padded state dimension: 1, 2 or 4) must be defined"
#endif
#ifndef BSSM_USER_NSTATES
#define BSSM_USER_NSTATES BSSM_USER_M   // states of the model (3 runs in the kernels padded to 4)
#endif

#include "pf_filter_impl.h"

extern "C" hipError_t bssm_user_launch_wg(const bssm::FilterBatch* b, hipStream_t stream) {
  bssm::FilterBatch local = *b;
  local.user_slot = 0;
  return bssm::launch_wg_mk<BSSM_USER_M, BSSM_USER_M>(local, stream);
}
extern "C" hipError_t bssm_user_launch_grid(const bssm::FilterBatch* b, const bssm::GridWorkspace* ws, hipStream_t stream,
                                            bssm::StepTiming* timing) {
  bssm::FilterBatch local = *b;
  local.user_slot = 0;
  return bssm::launch_grid_mk<BSSM_USER_M, BSSM_USER_M>(local, *ws, stream, timing);
}

// host entry points for the psi-APF / EKF set-up (mode search, linearisation) of a user-defined ssm_nlg model: (t, known_tv
// matrix as a HOST pointer with its dimensions) first -- ignored by snippets without BSSM_USER_TV
#ifdef BSSM_USER_NLG
#ifdef BSSM_USER_TV
#define BSSM_HTV bssm_user_tv{t, ktv, rows, cols},
#else
#define BSSM_HTV
#endif
extern "C" int bssm_user_host_has_tv() {
#ifdef BSSM_USER_TV
  return 1;
#else
  return 0;
#endif
}
extern "C" void bssm_user_host_T(int t, const double* ktv, int rows, int cols, const double* a, const double* th, const double* kn, double* out) { (void)t; (void)ktv; (void)rows; (void)cols; bssm_user_T(BSSM_HTV a, th, kn, out); }
extern "C" double bssm_user_host_Z(int t, const double* ktv, int rows, int cols, const double* a, const double* th, const double* kn) { (void)t; (void)ktv; (void)rows; (void)cols; return bssm_user_Z(BSSM_HTV a, th, kn); }
#ifdef BSSM_USER_HFN
extern "C" double bssm_user_host_H(int t, const double* ktv, int rows, int cols, const double* a, const double* th, const double* kn) { (void)t; (void)ktv; (void)rows; (void)cols; return bssm_user_H(BSSM_HTV a, th, kn); }
#endif
#ifdef BSSM_USER_RFN
extern "C" void bssm_user_host_R(int t, const double* ktv, int rows, int cols, const double* a, const double* th, const double* kn, double* out) { (void)t; (void)ktv; (void)rows; (void)cols; bssm_user_R(BSSM_HTV a, th, kn, out); }
#endif
#ifdef BSSM_USER_JACOBIANS
extern "C" void bssm_user_host_Tg(int t, const double* ktv, int rows, int cols, const double* a, const double* th, const double* kn, double* out) { (void)t; (void)ktv; (void)rows; (void)cols; bssm_user_Tg(BSSM_HTV a, th, kn, out); }
extern "C" void bssm_user_host_Zg(int t, const double* ktv, int rows, int cols, const double* a, const double* th, const double* kn, double* out) { (void)t; (void)ktv; (void)rows; (void)cols; bssm_user_Zg(BSSM_HTV a, th, kn, out); }
#endif
#endif
