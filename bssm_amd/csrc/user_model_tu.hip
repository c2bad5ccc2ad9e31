// user_model_tu.hip -- translation unit of a run-time compiled user model.
//
// bssm's ssm_nlg / ssm_sde models are user C++ compiled at run time (Rcpp::sourceCpp) and called through function
// pointers (src/model_ssm_nlg.h:10-21, src/model_ssm_sde.h:8-11).  Device code cannot call host pointers, so here the
// user's snippet of __device__ functions is compiled INTO the filter kernels: bssm_user_model_build() (api_filters.cpp) compiles this
// file SIX times side by side, one part per (sampler, kernel family) (-DBSSM_USER_PART=0...5: sampler = part % 3 -- bootstrap,
// psi-APF, EKF particle filter --, family = part / 3 -- workgroup-resident, step kernels; 4-8 s each instead of ~27 s in a row),
//   hipcc -O3 -std=c++17 -fPIC -c --offload-arch=gfx950 -I<csrc> -DBSSM_USER_M=<pad(m)> -DBSSM_USER_NSTATES=<m>
//         -DBSSM_USER_SNIPPET='"<file>"' -DBSSM_USER_PART=<k> user_model_tu.hip -o part<k>.o
// links them into libbssm_user_<hash>.so and registers the two launchers of part 0, which dispatch on the sampler.
// Without BSSM_USER_PART the file is one self-contained unit (every sampler).  The snippet's contract: pf_functors.h.
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
#error "BSSM_USER_M (padded state dimension: 1, 2, 4, 8 or 16) must be defined"
#endif
#ifndef BSSM_USER_NSTATES
#define BSSM_USER_NSTATES BSSM_USER_M   // states of the model (3 runs in the kernels padded to 4)
#endif

#ifdef BSSM_USER_PART   // 0 ... 5: sampler = part % 3 (bootstrap, psi-APF, EKF particle filter), kernel family = part / 3 (workgroup-resident, step)
#define BSSM_UM_SAMPLER ((BSSM_USER_PART) % 3)
#define BSSM_FN_PSI_MASK (1 << BSSM_UM_SAMPLER)
#define BSSM_UM_HAS_WG ((BSSM_USER_PART) < 3)
#define BSSM_UM_HAS_GRID ((BSSM_USER_PART) >= 3)
#if (BSSM_USER_PART) % 3 == 0
#define BSSM_UM_WG bssm_user_launch_wg_p0
#define BSSM_UM_GRID bssm_user_launch_grid_p0
#elif (BSSM_USER_PART) % 3 == 1
#define BSSM_UM_WG bssm_user_launch_wg_p1
#define BSSM_UM_GRID bssm_user_launch_grid_p1
#else
#define BSSM_UM_WG bssm_user_launch_wg_p2
#define BSSM_UM_GRID bssm_user_launch_grid_p2
#endif
#else
#define BSSM_UM_HAS_WG 1
#define BSSM_UM_HAS_GRID 1
#define BSSM_UM_WG bssm_user_launch_wg
#define BSSM_UM_GRID bssm_user_launch_grid
#endif

#ifdef BSSM_USER_PART
// the host functions whose BODIES depend on the part (which kernels they can return) get part-specific names: as templates they
// are weak symbols, and the link of the six parts would otherwise keep ONE body for all of them
#define BSSM_UM_CAT2(a, b) a##b
#define BSSM_UM_CAT(a, b) BSSM_UM_CAT2(a, b)
#define launch_wg_mk BSSM_UM_CAT(launch_wg_mk_part, BSSM_USER_PART)
#define launch_grid_mk BSSM_UM_CAT(launch_grid_mk_part, BSSM_USER_PART)
#define wg_fn_kernel BSSM_UM_CAT(wg_fn_kernel_part, BSSM_USER_PART)
#define wg_fn_kernel_exact BSSM_UM_CAT(wg_fn_kernel_exact_part, BSSM_USER_PART)
#define wg_fn_kernel_lean4 BSSM_UM_CAT(wg_fn_kernel_lean4_part, BSSM_USER_PART)
#define v2_fn_kernels BSSM_UM_CAT(v2_fn_kernels_part, BSSM_USER_PART)
#define launch_row_mk BSSM_UM_CAT(launch_row_mk_part, BSSM_USER_PART)
#define row_fn_kernels BSSM_UM_CAT(row_fn_kernels_part, BSSM_USER_PART)
#endif

#include "pf_filter_impl.h"

// the launchers of this unit's samplers and kernel family
#if BSSM_UM_HAS_WG
extern "C" hipError_t BSSM_UM_WG(const bssm::FilterBatch* b, hipStream_t stream) {
  bssm::FilterBatch local = *b;
  local.user_slot = 0;
  return bssm::launch_wg_mk<BSSM_USER_M, BSSM_USER_M>(local, stream);
}
#endif
#if BSSM_UM_HAS_GRID
extern "C" hipError_t BSSM_UM_GRID(const bssm::FilterBatch* b, const bssm::GridWorkspace* ws, hipStream_t stream,
                                   bssm::StepTiming* timing) {
  bssm::FilterBatch local = *b;
  local.user_slot = 0;
  return bssm::launch_grid_mk<BSSM_USER_M, BSSM_USER_M>(local, *ws, stream, timing);
}
#endif

#if !defined(BSSM_USER_PART) || BSSM_USER_PART == 0
extern "C" void bssm_user_set_options(const bssm::Options& (*getter)()) { bssm::user_options_getter() = getter; }
#endif
#if defined(BSSM_USER_PART) && BSSM_USER_PART == 0
// part 0 carries the entry points the main library looks up: they dispatch on the batch's sampler to the part that holds its kernels
extern "C" hipError_t bssm_user_launch_wg_p1(const bssm::FilterBatch*, hipStream_t);
extern "C" hipError_t bssm_user_launch_grid_p0(const bssm::FilterBatch*, const bssm::GridWorkspace*, hipStream_t, bssm::StepTiming*);
extern "C" hipError_t bssm_user_launch_wg_p2(const bssm::FilterBatch*, hipStream_t);
extern "C" hipError_t bssm_user_launch_grid_p1(const bssm::FilterBatch*, const bssm::GridWorkspace*, hipStream_t, bssm::StepTiming*);
extern "C" hipError_t bssm_user_launch_grid_p2(const bssm::FilterBatch*, const bssm::GridWorkspace*, hipStream_t, bssm::StepTiming*);
#ifdef BSSM_USER_BSF_ONLY   // more than 8 states: only the bootstrap sampler's parts (0 and 3) are built
extern "C" hipError_t bssm_user_launch_wg(const bssm::FilterBatch* b, hipStream_t stream) {
  return b->psi == 0 ? bssm_user_launch_wg_p0(b, stream) : hipErrorInvalidValue;
}
extern "C" hipError_t bssm_user_launch_grid(const bssm::FilterBatch* b, const bssm::GridWorkspace* ws, hipStream_t stream,
                                            bssm::StepTiming* timing) {
  return b->psi == 0 ? bssm_user_launch_grid_p0(b, ws, stream, timing) : hipErrorInvalidValue;
}
#elif defined(BSSM_USER_NO_EKF)   // 9 ... 16 states with Jacobians and a constant R: bootstrap and psi-APF (parts 0, 1, 3, 4)
extern "C" hipError_t bssm_user_launch_wg(const bssm::FilterBatch* b, hipStream_t stream) {
  return b->psi == 0 ? bssm_user_launch_wg_p0(b, stream) : (b->psi == 1 ? bssm_user_launch_wg_p1(b, stream) : hipErrorInvalidValue);
}
extern "C" hipError_t bssm_user_launch_grid(const bssm::FilterBatch* b, const bssm::GridWorkspace* ws, hipStream_t stream,
                                            bssm::StepTiming* timing) {
  return b->psi == 0 ? bssm_user_launch_grid_p0(b, ws, stream, timing) : (b->psi == 1 ? bssm_user_launch_grid_p1(b, ws, stream, timing) : hipErrorInvalidValue);
}
#else
extern "C" hipError_t bssm_user_launch_wg(const bssm::FilterBatch* b, hipStream_t stream) {
  return b->psi == 0 ? bssm_user_launch_wg_p0(b, stream) : (b->psi == 1 ? bssm_user_launch_wg_p1(b, stream) : bssm_user_launch_wg_p2(b, stream));
}
extern "C" hipError_t bssm_user_launch_grid(const bssm::FilterBatch* b, const bssm::GridWorkspace* ws, hipStream_t stream,
                                            bssm::StepTiming* timing) {
  return b->psi == 0 ? bssm_user_launch_grid_p0(b, ws, stream, timing)
                     : (b->psi == 1 ? bssm_user_launch_grid_p1(b, ws, stream, timing) : bssm_user_launch_grid_p2(b, ws, stream, timing));
}
#endif
#endif

#if !defined(BSSM_USER_PART) || BSSM_USER_PART == 0

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
#endif  // host entry points: whole unit or part 0
