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
// Late round 6: the psi set-up of ONE model with 9 ... 64 states does its smoother and conditional factors on the device (the host's share --
// ~0.6 s of a 0.68 s single psi-APF call at 53 states, 87 ms at 25 -- was the conditional factors' decompositions): psi_setup() finds the mode and
// the approximating y / H on the host as before and then asks this hook (set by api_postcorrect.cpp at load; nullptr in a build without it).
// Returns false when it did not handle the model (the host routines run); *finite = the result's verdict.
struct PsiSetup;
typedef bool (*DeviceFactorsFn)(const HostModel& hm, const std::vector<double>& approx_y, const std::vector<double>& approx_H, PsiSetup* out, bool* finite);
DeviceFactorsFn& device_factors_hook();
// is_mode == nullptr: iterate from hm.initial_mode.  Returns false on a numerical failure.
// local_approx = FALSE (src/model_ssm_ung.cpp:104-116): the pseudo-observations (approx y, H) of the FIRST approximation
// of a chain are kept; later parameter values only re-run the smoother against them.  `set` is false until then.
struct GlobalApprox {
  std::vector<double> y, H;
  bool set = false;
};
bool psi_setup(const HostModel& hm, const double* is_mode, PsiSetup* out, const GlobalApprox* global = nullptr);
// exact smoothing factors of a linear-Gaussian (ssm_ulg family) model for gaussian_psi_smoother:
// smoother_ccov + conditional_cov on the model itself (src/model_ssm_ulg.cpp:646-652)
bool gaussian_smoothing_setup(const HostModel& hm, PsiSetup* out);
// the same for a multivariate Gaussian model (ssm_mlg: `src` carries y n x p, Z p x m x (1|n), D p x (1|n) and H p x p x
// (1|n), the lower factor of the observation noise; src/model_ssm_mlg.cpp:310-396, 473-480).  Returns "" or the error.
std::string mlg_smoothing_setup(const bssm_ung_model* src, PsiSetup* out);
// smoothed means m x (n+1) and covariances m x m x (n+1) of a linear-Gaussian model (host Kalman smoother)
bool gaussian_smoother(const HostModel& hm, std::vector<double>* alphahat, std::vector<double>* Vt);
// ---- two-phase psi set-up, so that the O(n m^3) smoother of MANY draws can run as one device batch
// (lg_smoother.h; one GPU lane per draw).  *_begin does everything up to the smoother (approximation at the stored
// mode, scales) and packs the approximating linear-Gaussian model; the caller runs the smoother (device batch or
// lg_packed_run on the host) into su->alphahat / Vt / Ct; build_*_tables then proceeds as after psi_setup().
// They return false when the fixed-size smoother does not apply (m > 8): use psi_setup().  Multivariate observations
// (ssm_mng: independent series) go through the same smoother one series at a time (LgModel::sub); psi_setup_compact then
// keeps the entries of the time points.
struct LgPacked {
  int n = 0, m = 0;   // n: (virtual) steps of the smoother = time points x sub
  int sub = 1;        // series per time point treated one at a time (LgModel::sub); > 1: ssm_mng's approximating model
  int Zs = 0, Ts = 0, Rs = 0, Ds = 0, Cs = 0, Hs = 0;            // strides in doubles, 0 = time-invariant
  size_t oZ = 0, oT = 0, oR = 0, oD = 0, oC = 0, oy = 0, oH = 0, oa = 0, oP = 0;  // offsets into buf
  size_t ou = 0, ox = 0, om = 0, oh = 0;                          // ung_pack only: u, x'beta, stored mode, (distribution, phi)
  std::vector<double> buf;                                        // [Z | T | R R' | D | C | y - x'beta | H | a1 | P1]
  bool same_layout(const LgPacked& o) const {
    return n == o.n && m == o.m && sub == o.sub && Zs == o.Zs && Ts == o.Ts && Rs == o.Rs && Ds == o.Ds && Cs == o.Cs && Hs == o.Hs && buf.size() == o.buf.size();
  }
};
bool psi_setup_begin(const HostModel& hm, const double* is_mode, PsiSetup* out, LgPacked* lg);
void psi_setup_compact(const LgPacked& lg, PsiSetup* su);
// The WHOLE psi set-up of an ssm_ung draw on the device (ung_setup_kernel: Laplace step at the stored mode, scales,
// smoother, conditional factors, the filter's tables): the host only packs the draw -- [Z | T | R R' | D | C | y (raw) |
// a1 | P1 | u | x'beta | mode | distribution, phi].  False when the fixed-size smoother does not apply.
bool ung_pack(const HostModel& hm, const double* is_mode, LgPacked* lg);
// the same for an ssm_mng draw (p > 1 independent series per time point, m <= 8): mng_setup_kernel
bool mng_pack(const HostModel& hm, const double* is_mode, LgPacked* lg);
// host execution of the packed smoother (what the device kernel does per lane); false = non-finite result
bool lg_packed_run(const LgPacked& lg, double* alphahat, double* Vt, double* Ct);

// log-likelihood of the Gaussian approximation only (nsim = 0 path, src/model_ssm_ung.cpp:214-234)
double approx_loglik_only(const HostModel& hm);
// same, also returning the converged mode (p x n) -- what approx_mcmc::amcmc stores per accepted value
// `global` (may be NULL): filled by the first call, used by the later ones -- the chain's state for local_approx = FALSE
double approx_loglik_mode(const HostModel& hm, std::vector<double>* mode, GlobalApprox* global = nullptr);
// records the message bssm_amd_last_error() returns on this thread; returns `code`
int set_error(int code, const std::string& msg);

int pad_dim(int d);  // 1,2,4,8,16
int pad_noise_dim(int MP, int k);  // padded noise dimension with a kernel instantiation for this MP

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

// ---- built-in ssm_nlg model functors on the host: psi-APF set-up (EKF, mode search, linearised model,
// smoother_ccov + conditional_cov), src/model_ssm_nlg.cpp:62-222, 284-381, 854-879, 974-991, 1295-1319
// host entry points of a run-time compiled user model (exported by its library, user_model_tu.hip): the psi-APF / EKF
// set-up evaluates T, Z and their Jacobians on the host
// (t, known_tv_params as a host pointer, its rows and columns) come first: src/model_ssm_nlg.h:10-21
struct UserHostFns {
  void (*T)(int t, const double* ktv, int rows, int cols, const double* alpha, const double* theta, const double* known, double* out) = nullptr;
  double (*Z)(int t, const double* ktv, int rows, int cols, const double* alpha, const double* theta, const double* known) = nullptr;
  void (*Tg)(int t, const double* ktv, int rows, int cols, const double* alpha, const double* theta, const double* known, double* out) = nullptr;  // m x m column-major
  void (*Zg)(int t, const double* ktv, int rows, int cols, const double* alpha, const double* theta, const double* known, double* out) = nullptr;  // m
  // the reference's state-dependent H_fn / R_fn (src/model_ssm_nlg.h:12-13), exported by snippets with BSSM_USER_HFN / BSSM_USER_RFN:
  // the observation sd at a state, and the m x k matrix R at a state (column-major, leading dimension m)
  double (*H)(int t, const double* ktv, int rows, int cols, const double* alpha, const double* theta, const double* known) = nullptr;
  void (*R)(int t, const double* ktv, int rows, int cols, const double* alpha, const double* theta, const double* known, double* out) = nullptr;
  int has_tv = 0;  // the snippet was compiled with BSSM_USER_TV
  int jit = 0;     // compiled in process (bssm_user_model_jit): device code only, no host entry points
};
struct FnModel {
  int id = 0, n = 0, m = 1;
  int k = 1;  // noise terms (columns of R); read for user models with an R_fn only
  std::vector<double> y;
  double theta[64] = {0}, known[64] = {0};  // (FP_LONG_MAX entries, pf_functors.h)
  double x0 = 0;
  int positive = 0, L = 0;
  // user-defined ssm_nlg (id 7): host functions and the state-independent numbers the caller supplied
  const UserHostFns* user = nullptr;
  double uH = 1, uR[16] = {0}, ua1[16] = {0}, uP1[256] = {0};  // (user models: up to 16 states in the psi-APF, round 6)
  double uRm[256] = {0};  // the full constant R (m x m, column-major, unpadded) of a user model; uR is its diagonal
  const double* known_tv = nullptr;  // known_tv_params (host), tv_rows x tv_cols column-major: passed to BSSM_USER_TV snippets with t
  int tv_rows = 0, tv_cols = 0;
  // H or R depends on the state: the approximating model then has H_t = H_fn(mode_t), R_t = R_fn(mode_t) per time point
  bool state_noise() const { return id == 7 && user && (user->H || user->R); }
};
struct FnSetup {
  std::vector<double> Z, T, D, C;   // m x n, m x m x n, n, m x n : the mode-linearised model
  double H = 1, R[256] = {0};
  std::vector<double> Ht, Rt;       // models with state-dependent noise: H_t (n) and R_t (m x m x n, zero columns behind k); else empty
  std::vector<double> mode, scales;
  double approx_loglik = 0;
  std::vector<double> alphahat, Vt, Ct;
  int iters = 0;
  bool ok = false;
};
// loglik_only: stop after the mode search and the approximate log-likelihood (no smoothing covariances / conditional factors)
bool fn_psi_setup(const FnModel& mod, const double* is_mode, int max_iter, double conv_tol, FnSetup* out, bool loglik_only = false);
// two-phase variant for approximate_for_is (is_mode must be given): see psi_setup_begin
bool fn_psi_setup_begin(const FnModel& mod, const double* is_mode, FnSetup* out, LgPacked* lg);
// psi proposal operands (as build_psi_tables), observation rows carrying the approximating D_t / Z_t / 1/H / scales,
// and the approximating (T_t, C_t) rows used by the transition-density ratio
void build_fn_psi_tables(const FnModel& mod, const FnSetup& su, FilterTables* ft, std::vector<double>* amats);
// doubles per time point of the approximating rows: T_t (MP x MP), C_t (MP), and for a user snippet with a state-dependent R_fn
// the approximating R_t R_t' (MP x MP) -- must match fn_amat_row<MP>() of the kernels compiled with that snippet
inline int fn_amats_row(int MP, bool user_rfn) { return MP * MP + MP + (user_rfn ? MP * MP : 0); }
void fn_a1P1(const FnModel& mod, double* a1, double* P1 /* m x m */);
// EKF particle filter: prior (a1, P1) and the initial proposal (att1, lower factor L1), src/model_ssm_nlg.cpp:1154-1161
void fn_ekpf_init(const FnModel& mod, double* a1, double* P1, double* att1, double* L1);

// built-in update_model maps (see bssm_update_model_builtin in bssm_amd.h)
std::string update_model_builtin(int model_type, const int32_t* flags, const double* theta, int n_par,
                                 bssm_ung_model* model, double* T, double* R, double* P1, double* a1, double* C,
                                 double* beta);

}  // namespace bssm
