// host_model.h -- host-side model handling: ingest of the C-ABI model, the Gaussian
// approximation needed by the psi-APF, and the padded per-filter tables the kernels read.
#pragma once
#include <string>
#include <vector>

#include "../../include/bssm_amd.h"

namespace bssm {

// owning copy of a bssm_ung_model (the reference's constructor also copies: SURVEY 8b "ownership")
struct HostModel {
  int n = 0, m = 0, k = 0, p = 1;     // p observation series; arrays below are held p-major:
                                      // y, u, initial_mode p x n; Z p x m x (1|n); D p x (1|n)
  std::vector<int> dist;              // per-series distribution code
  std::vector<double> phis;           // per-series dispersion
  int Ztv = 0, Ttv = 0, Rtv = 0, Dtv = 0, Ctv = 0, Htv = 0;
  int distribution = 0;
  int max_iter = 100, local_approx = 1;
  double conv_tol = 1e-8, phi = 1.0;
  std::vector<double> y, Z, T, R, a1, P1, D, C, xbeta, u, H, initial_mode;
  bool gaussian() const { return p == 1 && distribution == BSSM_GAUSSIAN; }  // ssm_ulg family (sd = H_t)
};

// returns empty string on success, otherwise the validation error
std::string ingest_model(const bssm_ung_model* src, bool need_mode, HostModel* out);

// state-independent log-likelihood constants over finite y:
// src/model_ssm_ung.cpp:690-710, src/distr_consts.cpp:7-31; gaussian: norm_log_const(H_t) per
// observed t (src/model_ssm_ulg.cpp:595,632-633)
double const_term(const HostModel& hm);

// Result of approximate()/approximate_for_is() + smoother_ccov + conditional_cov
// (src/model_ssm_ung.cpp:87-161, 514-534; src/model_ssm_ulg.cpp:419-473; src/conditional_dist.cpp:4-31)
struct PsiSetup {
  std::vector<double> approx_y, approx_H, mode;  // p x n
  std::vector<double> scales;                     // n
  double approx_loglik = 0.0;
  std::vector<double> alphahat;  // m x (n+1)
  std::vector<double> Vt, Ct;    // m x m x (n+1)
  int iterations = 0;
};
// is_mode == nullptr: iterate from hm.initial_mode.  Returns false on a numerical failure.
bool psi_setup(const HostModel& hm, const double* is_mode, PsiSetup* out);
// log-likelihood of the Gaussian approximation only (nsim = 0 path, src/model_ssm_ung.cpp:214-234)
double approx_loglik_only(const HostModel& hm);

int pad_dim(int d);  // 1,2,4,8,16

// Padded tables for ONE filter, laid out as pf_kernels.h describes.
struct FilterTables {
  int MP = 0, KP = 0;
  int mats_tstride = 0;       // 0 when T,R,C are all time-invariant (bootstrap only)
  std::vector<double> mats;   // [(1|n)][MP*MP + MP*KP + MP]
  std::vector<double> obs;    // [n][OBS_HDR + MP]
  std::vector<double> init;   // [MP + MP*MP]
};
// bootstrap tables; KPmin lets a batch force a common padding
void build_bsf_tables(const HostModel& hm, int MP, int KP, FilterTables* out);
// psi-APF tables: T' = Ct(t+1), R' = Vt(t+1), C' = ahat(t+1) - Ct(t+1) ahat(t)
void build_psi_tables(const HostModel& hm, const PsiSetup& su, int MP, FilterTables* out);

// built-in update_model maps (see bssm_update_model_builtin in bssm_amd.h)
std::string update_model_builtin(int model_type, const int32_t* flags, const double* theta, int n_par,
                                 bssm_ung_model* model, double* T, double* R, double* P1, double* a1, double* C,
                                 double* beta);

}  // namespace bssm
