// pf_functors.h -- device-side registry of the built-in non-linear / SDE model functors.
//
// bssm's ssm_nlg and ssm_sde models are arbitrary user C++ behind function pointers
// (src/model_ssm_nlg.h:10-21, src/model_ssm_sde.h:8-11) compiled at run time from the snippets in
// R/cpp_example_models.R; device code cannot call those, so the shipped examples are provided as
// device functors selected by id.  theta / known_params keep the snippets' meaning.
#pragma once
#include "pf_device.h"

namespace bssm {

enum { FN_NLG_GROWTH = 1, FN_NLG_AR_EXP = 2, FN_NLG_SIN_EXP = 3, FN_NLG_LINEAR_GAUSSIAN = 4, FN_SDE_GBM = 5, FN_SDE_POISSON_OU = 6,
       FN_USER_NLG = 7, FN_USER_SDE = 8 };  // 7 / 8: functions supplied by the user and compiled at run time (user_model_tu.hip)
// The filter kernels carry the model family as a template value: FN_NONE (linear-state models), FN_ANY (functor model,
// id read from the parameter block and dispatched per call) or a model id > 0 fixed at compile time -- the per-particle
// code then contains one model's functions only (C5 device phase: 72 -> 62 ms with the growth model fixed)
constexpr int FN_NONE = 0, FN_ANY = -1;
__host__ __device__ inline bool fn_is_sde(int id) { return id == FN_SDE_GBM || id == FN_SDE_POISSON_OU || id == FN_USER_SDE; }

// ---- user-defined models (the reference's ssm_nlg / ssm_sde take arbitrary user C++ compiled at run time by
// Rcpp::sourceCpp behind function pointers, src/model_ssm_nlg.h:10-21, src/model_ssm_sde.h:8-11).  The analogue here:
// the user's snippet of __device__ functions is compiled by hipcc at run time INTO the filter kernels (one small
// shared library per snippet, user_model_tu.hip), selected by FN_USER_NLG / FN_USER_SDE.  theta and known_params arrive
// as in bssm_fn_model (up to 8 doubles each).  A snippet defines BSSM_USER_NLG and/or BSSM_USER_SDE and the functions:
// Every function is declared with BSSM_FN (= __host__ __device__ inline; the psi / EKF set-up runs them on the host).
//   BSSM_USER_NLG: void bssm_user_T(const double* alpha, const double* theta, const double* known, double* out);   T_fn
//                  double bssm_user_Z(const double* alpha, const double* theta, const double* known);              Z_fn
//   + BSSM_USER_JACOBIANS (needed by the psi-APF and the EKF particle filter, as in the reference):
//                  void bssm_user_Tg(alpha, theta, known, double* out /* m x m, column-major */);                  T_gn
//                  void bssm_user_Zg(alpha, theta, known, double* out /* m */);                                    Z_gn
//   BSSM_USER_SDE: double bssm_user_drift(double x, const double* theta);  bssm_user_diffusion;  bssm_user_ddiffusion;
//                  double bssm_user_log_obs(double y, double x, const double* theta);   normalised log-density of y | x
// H (observation sd), R, a1 and P1 are state-independent numbers the caller passes with the model -- unless the snippet
// supplies the reference's state-dependent H_fn / R_fn (src/model_ssm_nlg.h:12-13; evaluated at the particle for the
// observation density and at the ANCESTOR for the transition, src/model_ssm_nlg.cpp:952, 1125; in the psi-APF weight and the
// EKF proposal too, :892-932, 1212-1256):
//   BSSM_USER_HFN: double bssm_user_H(const double* alpha, const double* theta, const double* known);   observation sd
//   BSSM_USER_RFN: void bssm_user_R(alpha, theta, known, double* out /* m x k, column-major */);         R_fn
//   BSSM_USER_TV:  the reference's time argument and `known_tv_params` matrix (src/model_ssm_nlg.h:10-21: every function is
//                  called as f(t, alpha, theta, known_params, known_tv_params)).  With this macro defined EVERY bssm_user_*
//                  function of the snippet takes a leading `const bssm_user_tv& tv` argument: tv.t is the time index the
//                  reference passes (Z, H at the time of the observation; T, R at the time of the ancestor), tv.known_tv the
//                  matrix, column-major, tv.rows x tv.cols, exactly as the caller handed it over.
// The state may have up to 4 components (BSSM_USER_M = 1, 2 or 4 is the padded dimension the kernels are instantiated
// for; a 3-state model leaves the fourth component alone) and R any m x k matrix, k <= m (bootstrap filter / smoother /
// bootstrap IS correction; psi-APF and EKF particle filter: up to 8 states since late round 4; any constant R, or a state-dependent R_fn / H_fn).
// per-filter parameter block (doubles): id, L, positive, x0, theta[8], known[8], H (sd), R diagonal[2], log-normaliser,
// user slot, number of states, known_tv_params, full R (MP x MP, column-major, zero-padded; user models), inverse factor of R R'
enum { FP_ID = 0, FP_L = 1, FP_POS = 2, FP_X0 = 3, FP_THETA = 4, FP_KNOWN = 12, FP_H = 20, FP_R = 21, FP_LOGC = 23, FP_SLOT = 24 /* user-model registry slot (host only) */,
       FP_M = 25,
       // user models with BSSM_USER_TV: the known_tv_params matrix (device pointer as a bit pattern; rows, columns)
       FP_TVPTR = 26, FP_TVROWS = 27, FP_TVCOLS = 28,
       FP_TEND = 29,   // ssm_sde: length of the interval one Milstein call covers (1 in every filter; bssm_milstein: the caller's t)
       // user models: where theta / known_params start in the block -- FP_THETA / FP_KNOWN for up to 8 entries each (every shipped
       // model), else the tail behind the two matrices (round 6: the reference has no cap on either, src/model_ssm_nlg.h:23-31)
       FP_THOFF = 30, FP_KNOFF = 31,
       // full R (MP x MP, column-major, zero-padded; user models: up to 16 x 16 since round 5) and the inverse lower factor of
       // R R' for the psi-APF's transition-density ratio with any constant R (MP x MP, column-major)
       FP_RMAT = 32, FP_LONG_MAX = 64 /* most theta / known_params entries of a user model */,
       FP_SIZE = 32 + 512 + 2 * FP_LONG_MAX };  // (FP_SIZE: the largest block -- 16 x 16 matrices, user models of up to 16 states, both tails)
// The two matrices sit back to back in the padded dimension of the model, so that a block is as short as its model needs: the growth
// model's (MP = 2) is 38 doubles, and a post-correction job packs, uploads and reads fp_size(MP) per draw -- with fixed 16 x 16 slots a
// 2 340-draw job moved 10 MB of zeros and its call went from 40 to 47 ms.
#if defined(__HIPCC_RTC__)
#define BSSM_FP_HD __device__
#elif defined(__HIPCC__)
#define BSSM_FP_HD __host__ __device__
#else
#define BSSM_FP_HD
#endif
BSSM_FP_HD constexpr int fp_rlinv(int mp) { return FP_RMAT + mp * mp; }   // offset of the inverse lower factor of R R'
BSSM_FP_HD constexpr int fp_size(int mp) { return FP_RMAT + 2 * mp * mp; }  // doubles of one block without the tails
BSSM_FP_HD constexpr int fp_tail(int n_theta, int n_known) { return (n_theta > 8 ? n_theta : 0) + (n_known > 8 ? n_known : 0); }
// theta / known_params as a user snippet sees them (wave-uniform offsets read from the block itself)
#define BSSM_FP_UTH(fp_) ((const double*)((fp_) + (int)(fp_)[FP_THOFF]))
#define BSSM_FP_UKN(fp_) ((const double*)((fp_) + (int)(fp_)[FP_KNOFF]))

#if defined(BSSM_USER_MODEL) && defined(BSSM_USER_TV)
template <class PF>
__device__ __forceinline__ bssm_user_tv fn_user_tv(PF fp, int t) {
  return bssm_user_tv{t, (const double*)(unsigned long long)__double_as_longlong(fp[FP_TVPTR]), (int)fp[FP_TVROWS], (int)fp[FP_TVCOLS]};
}
#define BSSM_UTV(fp_, t_) fn_user_tv((fp_), (t_)),
#else
#define BSSM_UTV(fp_, t_)
#endif

// doubles per time point of the psi-APF's approximating rows (`amats`): T~_t (MP x MP), C~_t (MP); a user model with a
// state-dependent R_fn (BSSM_USER_RFN) carries the approximating R~_t R~_t' = R_fn(mode_t) R_fn(mode_t)' behind them, because the
// two covariances of the transition-density ratio then differ (src/model_ssm_nlg.cpp:919-932)
template <int MP>
__host__ __device__ constexpr int fn_amat_row() {
#if defined(BSSM_USER_MODEL) && defined(BSSM_USER_NLG) && defined(BSSM_USER_RFN)
  return MP * MP + MP + MP * MP;
#else
  return MP * MP + MP;
#endif
}

// ---- ssm_nlg: alpha_{t+1} = T(alpha_t) + R eta,  y_t ~ N(Z(alpha_t), H^2)   (src/model_ssm_nlg.cpp:1118-1126, 943-956)
// (t: the time index the reference passes to the model functions; only user models with BSSM_USER_TV read it)
// (PF / PR / PA: pointer types of the parameter block, the observation row and the approximating rows -- plain pointers, or
// constant-address-space pointers from as_uniform_table, which turn the loads of these wave-uniform values into s_load)
template <int MP, class PF>
__device__ __forceinline__ void nlg_T_fn(int id, PF fp, const double (&a)[MP], double (&x)[MP], int t = 0) {
  switch (id) {
    case FN_NLG_GROWTH: {  // R/cpp_example_models.R:415-428
      const double dT = fp[FP_KNOWN + 0], K = fp[FP_KNOWN + 1];
      const double ea = tl_exp(a[0]);
      const double r = fdiv(ea, 1.0 + ea);
      const double e = tl_exp(r * dT);
      x[0] = a[0];
      if constexpr (MP > 1) x[1] = fdiv(K * a[1] * e, K + a[1] * (e - 1.0));
    } break;
    case FN_NLG_AR_EXP: x[0] = fp[FP_THETA + 0] * (1.0 - fp[FP_THETA + 1]) + fp[FP_THETA + 1] * a[0]; break;  // :296-300
    case FN_NLG_SIN_EXP: x[0] = sin(a[0]); break;                                                            // :726-730
#if defined(BSSM_USER_MODEL) && defined(BSSM_USER_NLG)
    case FN_USER_NLG:
#pragma unroll
      for (int j = 0; j < MP; j++) x[j] = 0.0;  // (the snippet writes its own states: the padding components must not stay undefined)
      bssm_user_T(BSSM_UTV(fp, t) &a[0], BSSM_FP_UTH(fp), BSSM_FP_UKN(fp), &x[0]);
      break;
#endif
    default: x[0] = a[0]; break;                                                                             // linear_gaussian
  }
}
template <int MP, class PF>
__device__ __forceinline__ double nlg_Z_fn(int id, PF fp, const double (&x)[MP], int t = 0) {
  switch (id) {
#if defined(BSSM_USER_MODEL) && defined(BSSM_USER_NLG)
    case FN_USER_NLG: return bssm_user_Z(BSSM_UTV(fp, t) &x[0], BSSM_FP_UTH(fp), BSSM_FP_UKN(fp));
#endif
    case FN_NLG_GROWTH: return MP > 1 ? x[MP > 1 ? 1 : 0] : x[0];  // second state (:393-400)
    case FN_NLG_AR_EXP:
    case FN_NLG_SIN_EXP: return tl_exp(x[0]);
    default: return x[0];
  }
}

// Jacobian of T_fn (T_gn, R/cpp_example_models.R:432-450 etc.), MP x MP column-major: the mode-linearised model of the
// psi-APF set-up when it runs on the device (fn_setup_kernel); built-in models only (a user model's Jacobians are host
// entry points of its library)
template <int MP>
__device__ __forceinline__ void nlg_T_gn(int id, const double* __restrict__ fp, const double (&a)[MP], double (&Tg)[MP * MP]) {
#pragma unroll
  for (int q = 0; q < MP * MP; q++) Tg[q] = 0.0;
  switch (id) {
    case FN_NLG_GROWTH:
      if constexpr (MP > 1) {
        const double dT = fp[FP_KNOWN + 0], K = fp[FP_KNOWN + 1];
        const double e0 = exp(a[0]), r = e0 / (1.0 + e0), er = exp(r * dT);
        const double den = K + a[1] * (er - 1.0), tmp = er / (den * den);
        Tg[0] = 1.0;
        Tg[1] = dT * K * a[1] * (K - a[1]) * tmp * r / (1.0 + e0);
        Tg[3] = K * K * tmp;
      }
      break;
    case FN_NLG_AR_EXP: Tg[0] = fp[FP_THETA + 1]; break;
    case FN_NLG_SIN_EXP: Tg[0] = cos(a[0]); break;
    default: Tg[0] = 1.0; break;
  }
}

// ---- ssm_sde: Milstein scheme with 2^L sub-steps per unit time (src/milstein.cpp:18-48)
__device__ __forceinline__ double sde_drift(int id, const double* __restrict__ fp, double x) {
#if defined(BSSM_USER_MODEL) && defined(BSSM_USER_SDE)
  if (id == FN_USER_SDE) return bssm_user_drift(x, BSSM_FP_UTH(fp));
#endif
  return id == FN_SDE_GBM ? fp[FP_THETA + 0] * x : tl_exp(fp[FP_THETA + 0]) * (fp[FP_THETA + 1] - x);
}
__device__ __forceinline__ double sde_diffusion(int id, const double* __restrict__ fp, double x) {
#if defined(BSSM_USER_MODEL) && defined(BSSM_USER_SDE)
  if (id == FN_USER_SDE) return bssm_user_diffusion(x, BSSM_FP_UTH(fp));
#endif
  return id == FN_SDE_GBM ? fmax(0.0, fp[FP_THETA + 1] * x) : fp[FP_R];  // poisson_ou: exp(theta2), precomputed
}
__device__ __forceinline__ double sde_ddiffusion(int id, const double* __restrict__ fp, double x) {
#if defined(BSSM_USER_MODEL) && defined(BSSM_USER_SDE)
  if (id == FN_USER_SDE) return bssm_user_ddiffusion(x, BSSM_FP_UTH(fp));
#endif
  return id == FN_SDE_GBM ? (x > 0.0 ? fp[FP_THETA + 1] : 0.0) : 0.0;
}
__device__ __forceinline__ double milstein(int id, const double* __restrict__ fp, RngKey key, uint32_t purpose, uint32_t t,
                                           uint32_t i, double x) {
  const int L = (int)fp[FP_L];
  const int nsub = 1 << L;
  const double dt = fp[FP_TEND] / (double)nsub, sdt = sqrt(dt);  // (FP_TEND = 1 in the filters: the same bits as 1 / nsub)
  const bool positive = fp[FP_POS] != 0.0;
  for (int k = 0; k < nsub; k += 2) {
    double z0, z1;
    rng_normal_pair(key, purpose, t, i, (uint32_t)(k >> 1), z0, z1);
#pragma unroll
    for (int h = 0; h < 2; h++) {
      if (k + h >= nsub) break;
      const double dB = (h ? z1 : z0) * sdt;
      const double df = sde_diffusion(id, fp, x);
      x += sde_drift(id, fp, x) * dt + df * dB + 0.5 * df * sde_ddiffusion(id, fp, x) * (dB * dB - dt);
      if (positive) x = fabs(x);
    }
  }
  return x;
}

// ---- unified entry points used by the filter kernels when the FN template flag is set
template <int MP, int FID = 0>
__device__ __forceinline__ void functor_initial(const double* __restrict__ fp, const double* __restrict__ ini, RngKey key,
                                                uint32_t i, double (&x)[MP]) {
  const int id = FID > 0 ? FID : (int)fp[FP_ID];
  if (fn_is_sde(id)) {  // alpha_0 = milstein(x0) (src/model_ssm_sde.cpp:25-28)
    x[0] = milstein(id, fp, key, RNG_INIT, 0u, i, fp[FP_X0]);
#pragma unroll
    for (int c = 1; c < MP; c++) x[c] = 0.0;
  } else {  // a1 + psd_chol(P1) u (src/model_ssm_nlg.cpp:1069-1081)
    double z[MP];
    rng_normals<MP, false>(key, RNG_INIT, 0u, i, z);
#pragma unroll
    for (int j = 0; j < MP; j++) {
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < MP; c++) s += ini[MP + j + MP * c] * z[c];
      x[j] = ini[j] + s;
    }
  }
}

template <int MP, int KP, int FID = 0>
__device__ __forceinline__ void functor_propagate(const double* __restrict__ fp, RngKey key, const u32x4& blk0, uint32_t t,
                                                  uint32_t i, const double (&a)[MP], double (&x)[MP]) {
  const int id = FID > 0 ? FID : (int)fp[FP_ID];
  if (fn_is_sde(id)) {
    x[0] = milstein(id, fp, key, RNG_SDE, t, i, a[0]);
#pragma unroll
    for (int c = 1; c < MP; c++) x[c] = 0.0;
  } else {
    double z[KP];
    rng_normals_blk<KP, false>(blk0, key, RNG_TRANSITION, t, i, z);  // blk0 = the step block (functor models: WIDE blocks)
#if defined(BSSM_USER_MODEL) && defined(BSSM_USER_NLG)
    if (id == FN_USER_NLG) {  // alpha' = T(a) + R(a) eta with any m x k matrix R (src/model_ssm_nlg.cpp:1118-1126)
#pragma unroll
      for (int j = 0; j < MP; j++) x[j] = 0.0;
      bssm_user_T(BSSM_UTV(fp, (int)t) &a[0], BSSM_FP_UTH(fp), BSSM_FP_UKN(fp), &x[0]);
#ifdef BSSM_USER_RFN
      double Rm[MP * MP];
      {
        constexpr int mu = BSSM_USER_NSTATES <= MP ? BSSM_USER_NSTATES : MP;  // states of the user's model (3 pads to MP = 4)
        double Ru[MP * MP];
#pragma unroll
        for (int q = 0; q < MP * MP; q++) Ru[q] = 0.0;
        bssm_user_R(BSSM_UTV(fp, (int)t) &a[0], BSSM_FP_UTH(fp), BSSM_FP_UKN(fp), &Ru[0]);  // m x k, column-major with leading dimension m
#pragma unroll
        for (int c = 0; c < MP; c++)
#pragma unroll
          for (int j = 0; j < MP; j++) Rm[j + MP * c] = j < mu ? Ru[j + mu * c] : 0.0;  // columns >= k were left zero
      }
#else
      const double* __restrict__ Rm = fp + FP_RMAT;  // (constant R: read from the parameter block where it is used -- wave-uniform loads)
#endif
#pragma unroll
      for (int j = 0; j < MP; j++) {
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < KP; c++) acc += Rm[j + MP * c] * z[c];
        x[j] += acc;
      }
      return;
    }
#endif
    nlg_T_fn<MP>(id, fp, a, x, (int)t);
#pragma unroll
    for (int j = 0; j < MP; j++) if (j < KP) x[j] += fp[FP_R + j] * z[j];  // R is diagonal in every shipped model
  }
}

template <int MP, int FID = 0>
__device__ __forceinline__ double functor_log_weight(const double* __restrict__ fp, const double* __restrict__ row,
                                                     const double (&x)[MP], int t = 0) {
  if (row[RH_FLAG] == 0.0) return 0.0;
  const double* __restrict__ ob = row + ROW_HDR;
  const int id = FID > 0 ? FID : (int)fp[FP_ID];
  const double y = ob[OB_Y];
#if defined(BSSM_USER_MODEL) && defined(BSSM_USER_SDE)
  if (id == FN_USER_SDE) return bssm_user_log_obs(y, x[0], BSSM_FP_UTH(fp));
#endif
  if (id == FN_SDE_POISSON_OU) {  // R::dpois(y, exp(x), log): y x - e^x - lgamma(y + 1) (host puts lgamma in OB_AUX)
    return y * x[0] - tl_exp(x[0]) - ob[OB_AUX];
  }
  // normalised Gaussian log-density (dmvnorm(..., lwr = true, logd = true), src/dmvnorm.cpp:10-13)
#if defined(BSSM_USER_MODEL) && defined(BSSM_USER_NLG) && defined(BSSM_USER_HFN)
  if (id == FN_USER_NLG) {  // H_fn at the particle (src/model_ssm_nlg.cpp:950-953)
    const double h = bssm_user_H(BSSM_UTV(fp, t) &x[0], BSSM_FP_UTH(fp), BSSM_FP_UKN(fp));
    const double q = (y - bssm_user_Z(BSSM_UTV(fp, t) &x[0], BSSM_FP_UTH(fp), BSSM_FP_UKN(fp))) / h;
    return -0.9189385332046727 - tl_log(h) - 0.5 * q * q;  // -0.5 log(2 pi) - log h - q^2 / 2
  }
#endif
  const double mu = id == FN_SDE_GBM ? tl_log(x[0]) : nlg_Z_fn<MP>(id, fp, x, t);
  const double q = (y - mu) / fp[FP_H];
  return fp[FP_LOGC] - 0.5 * q * q;
}

// observation standard deviation at a state (H_fn, src/model_ssm_nlg.h:12): a number for every model except user snippets
// with BSSM_USER_HFN
template <int MP, class PF>
__device__ __forceinline__ double nlg_H_at(int id, PF fp, const double (&x)[MP], int t) {
#if defined(BSSM_USER_MODEL) && defined(BSSM_USER_NLG) && defined(BSSM_USER_HFN)
  if (id == FN_USER_NLG) return bssm_user_H(BSSM_UTV(fp, t) &x[0], BSSM_FP_UTH(fp), BSSM_FP_UKN(fp));
#endif
  return fp[FP_H];
}
template <int MP>
__device__ __forceinline__ double dmvnorm_cov_small(const double (&x)[MP], const double (&mean)[MP], const double (&S)[MP * MP]);
#if defined(BSSM_USER_MODEL) && defined(BSSM_USER_NLG) && defined(BSSM_USER_RFN)
// R(a) R(a)' of a user snippet with BSSM_USER_RFN (R_fn at the ancestor, src/model_ssm_nlg.cpp:924-925), MP x MP zero-padded
template <int MP, class PF>
__device__ __forceinline__ void user_RR_at(PF fp, const double (&a)[MP], int t, double (&RR)[MP * MP]) {
  constexpr int mu = BSSM_USER_NSTATES <= MP ? BSSM_USER_NSTATES : MP;
  double Ru[MP * MP];
#pragma unroll
  for (int q = 0; q < MP * MP; q++) Ru[q] = 0.0;
  bssm_user_R(BSSM_UTV(fp, t) &a[0], BSSM_FP_UTH(fp), BSSM_FP_UKN(fp), &Ru[0]);  // m x k, leading dimension m; columns >= k stay zero
#pragma unroll
  for (int j = 0; j < MP; j++)
#pragma unroll
    for (int i = 0; i < MP; i++) {
      double acc = 0.0;
#pragma unroll
      for (int q = 0; q < MP; q++) acc += ((i < mu) ? Ru[i + mu * q] : 0.0) * ((j < mu) ? Ru[j + mu * q] : 0.0);
      RR[i + MP * j] = acc;
    }
}
#endif

// R R' at the ancestor as the EKF particle filter needs it (src/model_ssm_nlg.cpp:1214-1216, 1240-1242): R_fn of a snippet with
// BSSM_USER_RFN, else the constant matrix R of a user model with 3 / 4 states (FP_RMAT, zero-padded), else diag(R_j^2)
template <int MP, class PF>
__device__ __forceinline__ void nlg_RR_at(int id, PF fp, const double (&a)[MP], int t, double (&RR)[MP * MP]) {
#if defined(BSSM_USER_MODEL) && defined(BSSM_USER_NLG) && defined(BSSM_USER_RFN)
  if (id == FN_USER_NLG) { user_RR_at<MP>(fp, a, t, RR); return; }
#endif
  if constexpr (MP > 2) {
#pragma unroll
    for (int j = 0; j < MP; j++)
#pragma unroll
      for (int i = 0; i < MP; i++) {
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < MP; q++) acc += fp[FP_RMAT + i + MP * q] * fp[FP_RMAT + j + MP * q];
        RR[i + MP * j] = acc;
      }
  } else {
#pragma unroll
    for (int q = 0; q < MP * MP; q++) RR[q] = 0.0;
#pragma unroll
    for (int j = 0; j < MP; j++) RR[j + MP * j] = fp[FP_R + j] * fp[FP_R + j];
  }
}

// psi-APF weight of ssm_nlg (src/model_ssm_nlg.cpp:880-935 minus scales_t): observation-density ratio between
// the model and its mode-linearised Gaussian approximation, and for t > 0 the transition-density ratio evaluated
// at the ANCESTOR.  H and R are constant in every shipped model, so the Gaussian normalisers cancel exactly.
//   row  : observation row of time t (y, approximating D_t in OB_D, approximating Z_t after the header, 1/H in OB_PSIIH)
//   amat : approximating T_{t-1} (MP x MP) and C_{t-1} (MP) -- only read when t > 0
template <int MP, int FID = 0, class PF, class PR, class PA>
__device__ __forceinline__ double functor_psi_weight(PF fp, PR row, PA amat, int t, const double (&x)[MP],
                                                     const double (&aprev)[MP]) {
  if (row[RH_FLAG] == 0.0) return 0.0;  // y_t missing: weights are ones, transition ratio included (:1040-1056)
  const int id = FID > 0 ? FID : (int)fp[FP_ID];
  double w = 0.0;
  {
    const PR ob = row + ROW_HDR;
    const double y = ob[OB_Y], ih = ob[OB_PSIIH];
    double lin = ob[OB_D];
#pragma unroll
    for (int c = 0; c < MP; c++) lin += ob[SER_HDR + c] * x[c];
    const double q2 = (y - lin) * ih;
#if defined(BSSM_USER_MODEL) && defined(BSSM_USER_NLG) && defined(BSSM_USER_HFN)
    if (id == FN_USER_NLG) {
      // state-dependent H_fn: the model's sd at the particle against the approximating sd H_fn(mode_t) = 1 / ih of the row; the
      // normalisers no longer cancel (src/model_ssm_nlg.cpp:892-900)
      const double h = nlg_H_at<MP>(id, fp, x, t);
      const double q1 = (y - nlg_Z_fn<MP>(id, fp, x, t)) / h;
      w = (-tl_log(h) - 0.5 * q1 * q1) - (tl_log(ih) - 0.5 * q2 * q2);
    } else
#endif
    {
      const double q1 = (y - nlg_Z_fn<MP>(id, fp, x, t)) * ih;
      w = 0.5 * (q2 * q2 - q1 * q1);
    }
  }
  if (t > 0) {
    double mean[MP];
    nlg_T_fn<MP>(id, fp, aprev, mean, t - 1);
#if defined(BSSM_USER_MODEL) && defined(BSSM_USER_NLG)
#if defined(BSSM_USER_RFN)
    if (id == FN_USER_NLG) {
      // state-dependent R_fn: N(x; T(a), R(a) R(a)') against N(x; T~ a + C~, R~ R~') with R~ = R_fn(mode_{t-1}) from the host
      // (behind T~, C~ in the approximating row): two general densities (src/model_ssm_nlg.cpp:919-932, src/dmvnorm.cpp:4-37)
      double am[MP], RR[MP * MP], RRa[MP * MP];
#pragma unroll
      for (int j = 0; j < MP; j++) {
        double v = amat[MP * MP + j];
#pragma unroll
        for (int c = 0; c < MP; c++) v += amat[j + MP * c] * aprev[c];
        am[j] = v;
      }
#pragma unroll
      for (int q = 0; q < MP * MP; q++) RRa[q] = amat[MP * MP + MP + q];
      user_RR_at<MP>(fp, aprev, t - 1, RR);
      w -= dmvnorm_cov_small<MP>(x, am, RRa) - dmvnorm_cov_small<MP>(x, mean, RR);
    } else
#else
    if (id == FN_USER_NLG) {
      // user-defined models, any constant m x k matrix R: both densities share the covariance R R', so the normalisers cancel
      // and the ratio is a difference of two quadratic forms in (R R')^-1 = Linv' Linv.  Linv (FP_RLINV, from the host) is
      // the inverse lower Cholesky factor of R R' on the components with a non-zero diagonal, zero elsewhere: components
      // without noise carry no density term, as in dmvnorm (src/dmvnorm.cpp:17-21, 40-45)
      double da[MP], dm[MP];
#pragma unroll
      for (int j = 0; j < MP; j++) {
        double am = amat[MP * MP + j];
#pragma unroll
        for (int c = 0; c < MP; c++) am += amat[j + MP * c] * aprev[c];
        da[j] = x[j] - am; dm[j] = x[j] - mean[j];
      }
#pragma unroll
      for (int i = 0; i < MP; i++) {
        double qa = 0.0, qm = 0.0;
#pragma unroll
        for (int j = 0; j <= i; j++) { const double l = fp[fp_rlinv(MP) + i + MP * j]; qa += l * da[j]; qm += l * dm[j]; }
        w += 0.5 * (qa * qa - qm * qm);
      }
    } else
#endif
#endif
    {
#pragma unroll
    for (int j = 0; j < MP; j++) {
      double am = amat[MP * MP + j];
#pragma unroll
      for (int c = 0; c < MP; c++) am += amat[j + MP * c] * aprev[c];
      const double ir = fp[fp_rlinv(MP) + j + MP * j];  // 1 / R_j, from the host (it was a per-particle division of two constants)
      const double da = (x[j] - am) * ir, dm = (x[j] - mean[j]) * ir;
      w += 0.5 * (da * da - dm * dm);
    }
    }
  }
  return w - row[RH_SCALE];
}

// ---- EKF-proposal particle filter of ssm_nlg (van der Merwe et al.; src/model_ssm_nlg.cpp:1148-1293).
// Every particle runs one EKF update from its ancestor: a_t = T(anc), P_t = R R' (diagonal in the shipped models),
// then att / Ptt from y_{t+1} (:1264-1293), proposal alpha' = att + psd_chol(Ptt) u, weight = g(y | alpha') N(alpha';
// T(anc), R R') / N(alpha'; att, Ptt).  The proposal is recomputed from the ancestor when the weight is evaluated
// (a few dozen flops) instead of being carried between the two places.
template <int MP>
__device__ __forceinline__ void nlg_Z_gn(int id, const double* __restrict__ fp, const double (&x)[MP], double (&Zg)[MP], int t = 0) {
#pragma unroll
  for (int c = 0; c < MP; c++) Zg[c] = 0.0;
  switch (id) {
#if defined(BSSM_USER_MODEL) && defined(BSSM_USER_NLG) && defined(BSSM_USER_JACOBIANS)
    case FN_USER_NLG: bssm_user_Zg(BSSM_UTV(fp, t) &x[0], BSSM_FP_UTH(fp), BSSM_FP_UKN(fp), &Zg[0]); break;
#endif
    case FN_NLG_GROWTH: if constexpr (MP > 1) Zg[1] = 1.0; break;
    case FN_NLG_AR_EXP:
    case FN_NLG_SIN_EXP: Zg[0] = tl_exp(x[0]); break;
    default: Zg[0] = 1.0; break;
  }
}
// lower Cholesky factor of a PSD MP x MP matrix with psd_chol's zero-diagonal rule (src/psd_chol.cpp:5-18), MP <= 2
// (MP > 2, user models with 3 or 4 states: general loops; `m` = the model's states, the threshold counts its columns; a
// padding component has a zero diagonal and drops out like any component without variance)
template <int MP>
__device__ __forceinline__ void psd_chol_small(const double (&P)[MP * MP], double (&L)[MP * MP], int m = MP) {
  const double eps = 2.220446049250313e-16;
#pragma unroll
  for (int q = 0; q < MP * MP; q++) L[q] = 0.0;
  if constexpr (MP > 2) {
    double dmax = P[0];
#pragma unroll
    for (int j = 1; j < MP; j++) dmax = fmax(dmax, P[j + MP * j]);
    const double thr = fmax(eps, eps * (double)m * dmax);
#pragma unroll
    for (int j = 0; j < MP; j++) {  // column j of the factor on the components above the threshold (left-looking)
      const bool nzj = P[j + MP * j] > thr;
      double d = P[j + MP * j];
#pragma unroll
      for (int q = 0; q < j; q++) d -= L[j + MP * q] * L[j + MP * q];
      const double ljj = nzj ? sqrt(d) : 0.0;
      L[j + MP * j] = ljj;
      const double inv = nzj ? 1.0 / ljj : 0.0;
#pragma unroll
      for (int i = j + 1; i < MP; i++) {
        double v = P[i + MP * j];
#pragma unroll
        for (int q = 0; q < j; q++) v -= L[i + MP * q] * L[j + MP * q];
        L[i + MP * j] = (nzj && P[i + MP * i] > thr) ? v * inv : 0.0;
      }
    }
  } else if constexpr (MP == 1) {
    if (P[0] > fmax(eps, eps * P[0])) L[0] = sqrt(P[0]);
  } else {
    const double thr = fmax(eps, eps * 2.0 * fmax(P[0], P[3]));
    const bool n0 = P[0] > thr, n1 = P[3] > thr;
    if (n0 && n1) {
      L[0] = sqrt(P[0]);
      L[1] = P[1] / L[0];
      L[3] = sqrt(P[3] - L[1] * L[1]);
    } else if (n0) {
      L[0] = sqrt(P[0]);
    } else if (n1) {
      L[3] = sqrt(P[3]);
    }
  }
}
// dmvnorm(x, mean, S, lwr = false, logd = true) for a covariance S (src/dmvnorm.cpp:4-37), MP <= 2
template <int MP>
__device__ __forceinline__ double dmvnorm_cov_small(const double (&x)[MP], const double (&mean)[MP], const double (&S)[MP * MP]) {
  const double l2pi = 1.8378770664093453;  // log(2 pi)
  if constexpr (MP > 2) {  // components with S_jj <= eps dropped, Cholesky of the rest, forward substitution (src/dmvnorm.cpp:14-37)
    const double eps = 2.220446049250313e-16;
    double L[MP * MP], z[MP], out = 0.0;
#pragma unroll
    for (int q = 0; q < MP * MP; q++) L[q] = 0.0;
#pragma unroll
    for (int j = 0; j < MP; j++) {
      const bool nzj = S[j + MP * j] > eps;
      double d = S[j + MP * j];
#pragma unroll
      for (int q = 0; q < j; q++) d -= L[j + MP * q] * L[j + MP * q];
      const double ljj = nzj ? sqrt(d) : 1.0;
      L[j + MP * j] = nzj ? ljj : 0.0;
      double sacc = x[j] - mean[j];
#pragma unroll
      for (int q = 0; q < j; q++) sacc -= L[j + MP * q] * z[q];
      z[j] = nzj ? sacc / ljj : 0.0;
      out += nzj ? (-0.5 * l2pi - tl_log(ljj) - 0.5 * z[j] * z[j]) : 0.0;
#pragma unroll
      for (int i = j + 1; i < MP; i++) {
        double v = S[i + MP * j];
#pragma unroll
        for (int q = 0; q < j; q++) v -= L[i + MP * q] * L[j + MP * q];
        L[i + MP * j] = (nzj && S[i + MP * i] > eps) ? v / ljj : 0.0;
      }
    }
    return out;
  } else if constexpr (MP == 1) {
    const double d = x[0] - mean[0];
    return -0.5 * (l2pi + tl_log(S[0]) + d * d / S[0]);
  } else {
    const double eps = 2.220446049250313e-16;
    const bool n0 = S[0] > eps, n1 = S[3] > eps;
    const double z0 = x[0] - mean[0], z1 = x[1] - mean[1];
    if (n0 && n1) {
      const double l00 = sqrt(S[0]), l10 = S[1] / l00, l11 = sqrt(S[3] - l10 * l10);
      const double w0 = z0 / l00, w1 = (z1 - l10 * w0) / l11;
      return -tl_log(l00) - tl_log(l11) - l2pi - 0.5 * (w0 * w0 + w1 * w1);
    } else if (n0 || n1) {
      const double v = n0 ? S[0] : S[3], z = n0 ? z0 : z1, l = sqrt(v), w = z / l;
      return -tl_log(l) - 0.5 * l2pi - 0.5 * w * w;
    }
    return 0.0;
  }
}
// dmvnorm(x, mean, L, lwr = true, logd = true): L is a factor (src/dmvnorm.cpp:9-11, 23-25)
template <int MP>
__device__ __forceinline__ double dmvnorm_lwr_small(const double (&x)[MP], const double (&mean)[MP], const double (&L)[MP * MP]) {
  if constexpr (MP == 1) {
    const double q = (x[0] - mean[0]) / L[0];
    return -0.5 * (1.8378770664093453 + 2.0 * tl_log(L[0]) + q * q);
  } else {
    double S[MP * MP];
#pragma unroll
    for (int j = 0; j < MP; j++)
#pragma unroll
      for (int i = 0; i < MP; i++) {
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < MP; q++) acc += L[i + MP * q] * L[j + MP * q];
        S[i + MP * j] = acc;
      }
    return dmvnorm_cov_small<MP>(x, mean, S);
  }
}
// the EKF proposal of one particle: at = T(anc); (att, Lf) = proposal mean and factor for the step to time tn
template <int MP, int FID = 0>
__device__ __forceinline__ void ekpf_proposal(const double* __restrict__ fp, const double* __restrict__ row, bool last,
                                              const double (&anc)[MP], double (&at)[MP], double (&att)[MP], double (&Lf)[MP * MP],
                                              int t = 0 /* time of the ancestor: T at t, Z and its Jacobian at t + 1 */) {
  const int id = FID > 0 ? FID : (int)fp[FP_ID];
  nlg_T_fn<MP>(id, fp, anc, at, t);
  double Pt[MP * MP];
  nlg_RR_at<MP>(id, fp, anc, t, Pt);
  if (last) {  // the final, never-weighted slot: the reference uses P_t itself as the factor (:1223-1226)
#pragma unroll
    for (int j = 0; j < MP; j++) att[j] = at[j];
#pragma unroll
    for (int q = 0; q < MP * MP; q++) Lf[q] = Pt[q];
    return;
  }
  double Ptt[MP * MP];
  if (row[RH_FLAG] != 0.0) {  // ekf_update_step (:1264-1293)
    const double y = row[ROW_HDR + OB_Y], H = nlg_H_at<MP>(id, fp, at, t + 1);  // H_fn at the predicted state (:1272)
    double Zg[MP], PZ[MP], K[MP];
    nlg_Z_gn<MP>(id, fp, at, Zg, t + 1);
    double F = H * H;
#pragma unroll
    for (int i = 0; i < MP; i++) {
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < MP; j++) acc += Pt[i + MP * j] * Zg[j];
      PZ[i] = acc;
    }
#pragma unroll
    for (int j = 0; j < MP; j++) F += Zg[j] * PZ[j];
    const double icf = 1.0 / sqrt(F), v = y - nlg_Z_fn<MP>(id, fp, at, t + 1);
#pragma unroll
    for (int i = 0; i < MP; i++) { K[i] = PZ[i] * icf * icf; att[i] = at[i] + K[i] * v; }
    double IKZ[MP * MP], tmp[MP * MP];
#pragma unroll
    for (int j = 0; j < MP; j++)
#pragma unroll
      for (int i = 0; i < MP; i++) IKZ[i + MP * j] = (i == j ? 1.0 : 0.0) - K[i] * Zg[j];
#pragma unroll
    for (int j = 0; j < MP; j++)
#pragma unroll
      for (int i = 0; i < MP; i++) {
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < MP; q++) acc += IKZ[i + MP * q] * Pt[q + MP * j];
        tmp[i + MP * j] = acc;
      }
#pragma unroll
    for (int j = 0; j < MP; j++)
#pragma unroll
      for (int i = 0; i < MP; i++) {
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < MP; q++) acc += tmp[i + MP * q] * IKZ[j + MP * q];
        Ptt[i + MP * j] = acc + K[i] * H * H * K[j];
      }
  } else {
#pragma unroll
    for (int j = 0; j < MP; j++) att[j] = at[j];
#pragma unroll
    for (int q = 0; q < MP * MP; q++) Ptt[q] = Pt[q];
  }
  psd_chol_small<MP>(Ptt, Lf, MP > 2 ? (int)fp[FP_M] : MP);
}
template <int MP, int FID = 0>
__device__ __forceinline__ void functor_ekpf_propagate(const double* __restrict__ fp, const double* __restrict__ row_next, bool last,
                                                       RngKey key, const u32x4& blk0, uint32_t t, uint32_t i,
                                                       const double (&anc)[MP], double (&x)[MP]) {
  double at[MP], att[MP], Lf[MP * MP], z[MP];
  ekpf_proposal<MP, FID>(fp, row_next, last, anc, at, att, Lf, (int)t);
  rng_normals_blk<MP, false>(blk0, key, RNG_TRANSITION, t, i, z);
#pragma unroll
  for (int j = 0; j < MP; j++) {
    double acc = 0.0;
#pragma unroll
    for (int c = 0; c < MP; c++) acc += Lf[j + MP * c] * z[c];
    x[j] = att[j] + acc;
  }
}
// EKPF weight at time t (:1178-1182 for t = 0, :1236-1244 after): `init` = [a1 | P1 | att1 | L1] of the filter
template <int MP, int FID = 0>
__device__ __forceinline__ double functor_ekpf_weight(const double* __restrict__ fp, const double* __restrict__ row,
                                                      const double* __restrict__ init, int t, const double (&x)[MP],
                                                      const double (&anc)[MP]) {
  if (row[RH_FLAG] == 0.0) return 0.0;
  double w = functor_log_weight<MP, FID>(fp, row, x, t);
  if (t == 0) {
    double a1[MP], P1[MP * MP], att1[MP], L1[MP * MP];
#pragma unroll
    for (int j = 0; j < MP; j++) { a1[j] = init[j]; att1[j] = init[MP + MP * MP + j]; }
#pragma unroll
    for (int q = 0; q < MP * MP; q++) { P1[q] = init[MP + q]; L1[q] = init[2 * MP + MP * MP + q]; }
    return w + dmvnorm_cov_small<MP>(x, a1, P1) - dmvnorm_lwr_small<MP>(x, att1, L1);
  }
  double at[MP], att[MP], Lf[MP * MP], RR[MP * MP];
  ekpf_proposal<MP, FID>(fp, row, false, anc, at, att, Lf, t - 1);
  nlg_RR_at<MP>((int)(FID > 0 ? FID : (int)fp[FP_ID]), fp, anc, t - 1, RR);
  return w + dmvnorm_cov_small<MP>(x, at, RR) - dmvnorm_lwr_small<MP>(x, att, Lf);
}

}  // namespace bssm
