// pf_functors.h -- device-side registry of the built-in non-linear / SDE model functors.
//
// bssm's ssm_nlg and ssm_sde models are arbitrary user C++ behind function pointers
// (src/model_ssm_nlg.h:10-21, src/model_ssm_sde.h:8-11) compiled at run time from the snippets in
// R/cpp_example_models.R; device code cannot call those, so the shipped examples are provided as
// device functors selected by id.  theta / known_params keep the snippets' meaning.
#pragma once
#include "pf_device.h"

namespace bssm {

enum { FN_NLG_GROWTH = 1, FN_NLG_AR_EXP = 2, FN_NLG_SIN_EXP = 3, FN_NLG_LINEAR_GAUSSIAN = 4, FN_SDE_GBM = 5, FN_SDE_POISSON_OU = 6 };
// per-filter parameter block (doubles): id, L, positive, x0, theta[8], known[8], H (sd), R diagonal[2], log-normaliser
enum { FP_ID = 0, FP_L = 1, FP_POS = 2, FP_X0 = 3, FP_THETA = 4, FP_KNOWN = 12, FP_H = 20, FP_R = 21, FP_LOGC = 23, FP_SIZE = 24 };

// ---- ssm_nlg: alpha_{t+1} = T(alpha_t) + R eta,  y_t ~ N(Z(alpha_t), H^2)   (src/model_ssm_nlg.cpp:1118-1126, 943-956)
template <int MP>
__device__ __forceinline__ void nlg_T_fn(int id, const double* __restrict__ fp, const double (&a)[MP], double (&x)[MP]) {
  switch (id) {
    case FN_NLG_GROWTH: {  // R/cpp_example_models.R:415-428
      const double dT = fp[FP_KNOWN + 0], K = fp[FP_KNOWN + 1];
      const double ea = fast_exp(a[0]);
      const double r = ea / (1.0 + ea);
      const double e = fast_exp(r * dT);
      x[0] = a[0];
      if constexpr (MP > 1) x[1] = K * a[1] * e / (K + a[1] * (e - 1.0));
    } break;
    case FN_NLG_AR_EXP: x[0] = fp[FP_THETA + 0] * (1.0 - fp[FP_THETA + 1]) + fp[FP_THETA + 1] * a[0]; break;  // :296-300
    case FN_NLG_SIN_EXP: x[0] = sin(a[0]); break;                                                            // :726-730
    default: x[0] = a[0]; break;                                                                             // linear_gaussian
  }
}
template <int MP>
__device__ __forceinline__ double nlg_Z_fn(int id, const double (&x)[MP]) {
  switch (id) {
    case FN_NLG_GROWTH: return MP > 1 ? x[MP > 1 ? 1 : 0] : x[0];  // second state (:393-400)
    case FN_NLG_AR_EXP:
    case FN_NLG_SIN_EXP: return fast_exp(x[0]);
    default: return x[0];
  }
}

// ---- ssm_sde: Milstein scheme with 2^L sub-steps per unit time (src/milstein.cpp:18-48)
__device__ __forceinline__ double sde_drift(int id, const double* __restrict__ fp, double x) {
  return id == FN_SDE_GBM ? fp[FP_THETA + 0] * x : fast_exp(fp[FP_THETA + 0]) * (fp[FP_THETA + 1] - x);
}
__device__ __forceinline__ double sde_diffusion(int id, const double* __restrict__ fp, double x) {
  return id == FN_SDE_GBM ? fmax(0.0, fp[FP_THETA + 1] * x) : fp[FP_R];  // poisson_ou: exp(theta2), precomputed
}
__device__ __forceinline__ double sde_ddiffusion(int id, const double* __restrict__ fp, double x) {
  return id == FN_SDE_GBM ? (x > 0.0 ? fp[FP_THETA + 1] : 0.0) : 0.0;
}
__device__ __forceinline__ double milstein(int id, const double* __restrict__ fp, RngKey key, uint32_t purpose, uint32_t t,
                                           uint32_t i, double x) {
  const int L = (int)fp[FP_L];
  const int nsub = 1 << L;
  const double dt = 1.0 / (double)nsub, sdt = sqrt(dt);
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
template <int MP>
__device__ __forceinline__ void functor_initial(const double* __restrict__ fp, const double* __restrict__ ini, RngKey key,
                                                uint32_t i, double (&x)[MP]) {
  const int id = (int)fp[FP_ID];
  if (id >= FN_SDE_GBM) {  // alpha_0 = milstein(x0) (src/model_ssm_sde.cpp:25-28)
    x[0] = milstein(id, fp, key, RNG_INIT, 0u, i, fp[FP_X0]);
#pragma unroll
    for (int c = 1; c < MP; c++) x[c] = 0.0;
  } else {  // a1 + psd_chol(P1) u (src/model_ssm_nlg.cpp:1069-1081)
    double z[MP];
    rng_normals<MP>(key, RNG_INIT, 0u, i, z);
#pragma unroll
    for (int j = 0; j < MP; j++) {
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < MP; c++) s += ini[MP + j + MP * c] * z[c];
      x[j] = ini[j] + s;
    }
  }
}

template <int MP, int KP>
__device__ __forceinline__ void functor_propagate(const double* __restrict__ fp, RngKey key, const u32x4& blk0, uint32_t t,
                                                  uint32_t i, const double (&a)[MP], double (&x)[MP]) {
  const int id = (int)fp[FP_ID];
  if (id >= FN_SDE_GBM) {
    x[0] = milstein(id, fp, key, RNG_SDE, t, i, a[0]);
#pragma unroll
    for (int c = 1; c < MP; c++) x[c] = 0.0;
  } else {
    double z[KP];
    rng_normals_blk<KP>(blk0, key, RNG_TRANSITION, t, i, z);  // blk0 = the step block
    nlg_T_fn<MP>(id, fp, a, x);
#pragma unroll
    for (int j = 0; j < MP; j++) if (j < KP) x[j] += fp[FP_R + j] * z[j];  // R is diagonal in every shipped model
  }
}

template <int MP>
__device__ __forceinline__ double functor_log_weight(const double* __restrict__ fp, const double* __restrict__ row,
                                                     const double (&x)[MP]) {
  if (row[RH_FLAG] == 0.0) return 0.0;
  const double* __restrict__ ob = row + ROW_HDR;
  const int id = (int)fp[FP_ID];
  const double y = ob[OB_Y];
  if (id == FN_SDE_POISSON_OU) {  // R::dpois(y, exp(x), log): y x - e^x - lgamma(y + 1) (host puts lgamma in OB_AUX)
    return y * x[0] - fast_exp(x[0]) - ob[OB_AUX];
  }
  // normalised Gaussian log-density (dmvnorm(..., lwr = true, logd = true), src/dmvnorm.cpp:10-13)
  const double mu = id == FN_SDE_GBM ? fast_log(x[0]) : nlg_Z_fn<MP>(id, x);
  const double q = (y - mu) / fp[FP_H];
  return fp[FP_LOGC] - 0.5 * q * q;
}

// psi-APF weight of ssm_nlg (src/model_ssm_nlg.cpp:880-935 minus scales_t): observation-density ratio between
// the model and its mode-linearised Gaussian approximation, and for t > 0 the transition-density ratio evaluated
// at the ANCESTOR.  H and R are constant in every shipped model, so the Gaussian normalisers cancel exactly.
//   row  : observation row of time t (y, approximating D_t in OB_D, approximating Z_t after the header, 1/H in OB_PSIIH)
//   amat : approximating T_{t-1} (MP x MP) and C_{t-1} (MP) -- only read when t > 0
template <int MP>
__device__ __forceinline__ double functor_psi_weight(const double* __restrict__ fp, const double* __restrict__ row,
                                                     const double* __restrict__ amat, int t, const double (&x)[MP],
                                                     const double (&aprev)[MP]) {
  if (row[RH_FLAG] == 0.0) return 0.0;  // y_t missing: weights are ones, transition ratio included (:1040-1056)
  const int id = (int)fp[FP_ID];
  double w = 0.0;
  {
    const double* __restrict__ ob = row + ROW_HDR;
    const double y = ob[OB_Y], ih = ob[OB_PSIIH];
    const double q1 = (y - nlg_Z_fn<MP>(id, x)) * ih;
    double lin = ob[OB_D];
#pragma unroll
    for (int c = 0; c < MP; c++) lin += ob[SER_HDR + c] * x[c];
    const double q2 = (y - lin) * ih;
    w = 0.5 * (q2 * q2 - q1 * q1);
  }
  if (t > 0) {
    double mean[MP];
    nlg_T_fn<MP>(id, fp, aprev, mean);
#pragma unroll
    for (int j = 0; j < MP; j++) {
      double am = amat[MP * MP + j];
#pragma unroll
      for (int c = 0; c < MP; c++) am += amat[j + MP * c] * aprev[c];
      const double ir = 1.0 / fp[FP_R + j];
      const double da = (x[j] - am) * ir, dm = (x[j] - mean[j]) * ir;
      w += 0.5 * (da * da - dm * dm);
    }
  }
  return w - row[RH_SCALE];
}

}  // namespace bssm
