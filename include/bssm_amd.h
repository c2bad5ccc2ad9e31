/*
 * bssm_amd.h -- C ABI of the MI355X particle-filter engine that stands in for the
 * particle-filter / IS post-correction entry points of helske/bssm's shared library.
 *
 * Every entry point names the reference interface it replaces (file:line under the
 * reference tree).  The reference boundary is R's .Call into Rcpp-generated
 * `extern "C" SEXP _bssm_<name>(SEXP...)` stubs (src/RcppExports.cpp:57-69,1287-1351);
 * the stubs only unpack the R list into the C++ model constructor
 * (src/model_ssm_ung.cpp:9-32).  `bssm_ung_model` below is that constructor's field list
 * as plain pointers, so an R-side shim is a field-by-field REAL()/INTEGER() unpack
 * (INTEGRATION.md shows it).
 *
 * Conventions
 *  - All arrays are column-major doubles exactly as R / Armadillo hold them.
 *  - Inputs are read during the call only; outputs are caller-allocated; a NULL
 *    output pointer skips that output (and the work behind it where possible).
 *  - Return value: 0 = ok, BSSM_WARN_DEGENERATE = filter stopped on a non-finite
 *    log-likelihood (the reference issues Rcpp::warning and returns -Inf,
 *    src/R_bootstrap_filter.cpp:140-141), negative = error (message via
 *    bssm_amd_last_error()).  There is NO CPU fallback: without a usable HIP device
 *    every compute entry point returns BSSM_ERR_NO_DEVICE.
 *  - Random numbers: counter-based Philox4x32-10 keyed by (seed, filter index); see
 *    DESIGN.md.  Same (model, nsim, seed) => bit-identical output on every call,
 *    for any number of GPUs (reference property: tests/testthat/test_bootstrap_filter.R:5-17).
 *    ("Any number of GPUs" is verified on logical shards of ONE physical GPU and with two gloo ranks: no multi-GPU
 *    node has been available to this build.)  Results for a given seed are not reproducible ACROSS library versions
 *    whose bssm_amd_version() names another random-stream format or normalisation.
 *  - Limits (round 6): state dimension m <= 64, observation series p <= 16; particles up to 262 144 for m <= 16 and up to 102 400 for
 *    17 <= m <= 64 (the reference's front-end stops at 1e5, R/check_arguments.R:407-418); user-defined models <= 16 states (bootstrap
 *    paths; psi-APF with Jacobians and a constant R), <= 8 (EKF particle filter; psi-APF with a state-dependent R_fn), <= 64 theta and
 *    <= 64 known_params entries.  Beyond them the entry points refuse with a message.
 */
#ifndef BSSM_AMD_H
#define BSSM_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BSSM_OK 0
#define BSSM_WARN_DEGENERATE 1
#define BSSM_ERR_NO_DEVICE (-1)
#define BSSM_ERR_INVALID (-2)
#define BSSM_ERR_UNSUPPORTED (-3)
#define BSSM_ERR_HIP (-4)
#define BSSM_ERR_NUMERIC (-5)

/* distribution codes after `pmatch(...) - 1`, R/bootstrap_filter.R:106-109;
 * 5 = gaussian observation (ssm_ulg / bsm_lg / ar1_lg with `gaussian = TRUE`). */
enum { BSSM_SVM = 0, BSSM_POISSON = 1, BSSM_BINOMIAL = 2, BSSM_NEGBIN = 3, BSSM_GAMMA = 4, BSSM_GAUSSIAN = 5 };
/* model_type(): R/model_type.R:1-16 (non-gaussian branch; gaussian branch uses 1..3 too) */
enum { BSSM_SSM_MNG = 0, BSSM_SSM_UNG = 1, BSSM_BSM_NG = 2, BSSM_SVM_MODEL = 3, BSSM_AR1_NG = 4 };
/* sampling_method: src/model_ssm_ung.cpp:163 */
enum { BSSM_PSI = 1, BSSM_BSF = 2, BSSM_EKPF = 4 };  /* 4: EKF-proposal particle filter, ssm_nlg only (src/model_ssm_nlg.cpp:237-238) */

/* Fields read by ssm_ung::ssm_ung (src/model_ssm_ung.cpp:9-32) and, for gaussian = TRUE,
 * ssm_ulg::ssm_ulg (src/model_ssm_ulg.cpp:8-38).  "(1|n)": 1 when time-invariant, n
 * otherwise -- the *_tv flags carry what the reference infers from trailing extents
 * (src/model_ssm_ung.cpp:15-16). */
typedef struct bssm_ung_model {
  int32_t n, m, k;
  int32_t Ztv, Ttv, Rtv, Dtv, Ctv, Htv;
  int32_t distribution;
  int32_t n_xreg;           /* columns of xreg (0 = none) */
  int32_t max_iter;         /* Laplace approximation: model_ssm_ung.cpp:21 */
  int32_t local_approx;
  double conv_tol;
  double phi;
  const double* y;            /* n, NaN = missing */
  const double* Z;            /* m x (1|n) */
  const double* T;            /* m x m x (1|n) */
  const double* R;            /* m x k x (1|n) */
  const double* a1;           /* m */
  const double* P1;           /* m x m */
  const double* D;            /* (1|n) */
  const double* C;            /* m x (1|n) */
  const double* xreg;         /* n x n_xreg or NULL */
  const double* beta;         /* n_xreg or NULL */
  const double* u;            /* n */
  const double* H;            /* (1|n) observation sd; gaussian only */
  const double* initial_mode; /* n; psi-APF only */
  /* ---- multivariate observations: ssm_mng (model_type 0), fields of src/model_ssm_mng.cpp:7-31.
   * p <= 1 means univariate and the three fields below are ignored.  With p > 1: y is n x p and u is n x p as R
   * holds them (the reference transposes on entry, :8,:17), Z is p x m x (1|n), D is p x (1|n), initial_mode is
   * n x p; `distribution` / `phi` above are ignored in favour of the per-series vectors. */
  int32_t p;
  const int32_t* distribution_p; /* p codes in 1..5 */
  const double* phi_p;           /* p */
} bssm_ung_model;

/* Outputs of the filter / smoother wrappers, shaped as the reference returns them to R:
 *   bsf           (src/R_bootstrap_filter.cpp:151-155): at (n+1) x m, att n x m, Pt m x m x (n+1),
 *                 Ptt m x m x n, weights N x (n+1), logLik, alpha m x (n+1) x N
 *   bsf_smoother / psi_smoother (src/R_bootstrap_filter.cpp:374-377, src/R_psi.cpp:97-101):
 *                 alphahat (n+1) x m, Vt m x m x (n+1), weights, logLik, alpha (back-traced)
 * `indices` (N x n ancestor table, uint32) is an extension for testing; the reference keeps
 * it internal. */
typedef struct bssm_filter_out {
  double* at;
  double* att;
  double* Pt;
  double* Ptt;
  double* alphahat;
  double* Vt;
  double* weights;
  double* alpha;
  uint32_t* indices;
  double logLik;
} bssm_filter_out;

/* ---- library / device ---- */
int bssm_amd_device_count(void);
const char* bssm_amd_last_error(void);
const char* bssm_amd_version(void);
/* The library keeps freed device buffers in an arena (default cap 96 GiB, env BSSM_AMD_CACHE_GB) because
 * hipMalloc/hipFree of multi-GB scratch dominate repeated calls; this returns all cached blocks to the driver. */
/* diagnostic (round 6): whether the stream that steps the tail of a post-correction chunk was MEASURED to run beside the current
 * device's null stream (csrc/options.cpp: prewarm_streams).  0 not calibrated, 1 yes, 2 no (tails then go behind their rounds) */
int bssm_amd_tail_overlap_state(int32_t* candidates_tried);
void bssm_amd_trim_cache(void);
/* The library reads its BSSM_AMD_* environment switches once, when it is loaded (bssm_amd/csrc/options.h lists them); a process
 * that changes one afterwards calls this to have them read again.  Calls already in flight keep the snapshot they started with. */
int bssm_amd_reload_env(void);
/* wall-clock split of the last bssm_is_correction / bssm_is_correction_nlg call on this thread:
 * out[0] = per-draw host set-up (ms), out[1] = device phase incl. copies (ms) */
void bssm_amd_last_is_timing(double out[2]);
/* the same call split four ways (ms): out[0] = per-draw host packing (prepare / table building on the caller's threads),
 * out[1] = the per-chunk device set-up call (uploads, set-up kernel launch and -- on calls of more than 4 096 draws --
 * the wait for that kernel), out[2] = busy time of the thread that runs the filters (overlaps out[0] and out[1]),
 * out[3] = wall time of the pipeline */
void bssm_amd_last_is_timing4(double out[4]);

/* ---- drop-in entry points (host buffers in, host buffers out) ---- */

/* replaces _bssm_bsf: `Rcpp::List bsf(model_, nsim, seed, gaussian, model_type)`
 * src/R_bootstrap_filter.cpp:15-241, registered src/RcppExports.cpp:57-69 */
int bssm_bsf(const bssm_ung_model* model, uint32_t nsim, uint32_t seed, int gaussian, int model_type,
             bssm_filter_out* out);

/* replaces _bssm_bsf_smoother: src/R_bootstrap_filter.cpp:245-457 */
int bssm_bsf_smoother(const bssm_ung_model* model, uint32_t nsim, uint32_t seed, int gaussian, int model_type,
                      bssm_filter_out* out);

/* replaces _bssm_psi_smoother: `psi_smoother(model_, nsim, seed, model_type)` src/R_psi.cpp:52-171 */
int bssm_psi_smoother(const bssm_ung_model* model, uint32_t nsim, uint32_t seed, int model_type,
                      bssm_filter_out* out);

/* replaces _bssm_gaussian_psi_smoother: `arma::cube gaussian_psi_smoother(model_, nsim, seed, model_type)`
 * src/R_psi.cpp:14-50 -> ssm_ulg::psi_filter src/model_ssm_ulg.cpp:644-672 (model_type 1..3) or ssm_mlg::psi_filter
 * src/model_ssm_mlg.cpp:473-501 (model_type 0): exact draws from the joint smoothing distribution.  alpha: m x (n+1) x N.
 * ssm_mlg: p > 1, y n x p, Z p x m x (1|n), D p x (1|n), H p x p x (1|n) = lower factor of the observation noise
 * (Htv says whether it varies); distribution_p / phi_p / u are not read. */
int bssm_gaussian_psi_smoother(const bssm_ung_model* model, uint32_t nsim, uint32_t seed, int model_type, double* alpha);

/* host Kalman smoother of a linear-Gaussian model (ssm_ulg::smoother, src/model_ssm_ulg.cpp:307-417): alphahat m x (n+1),
 * Vt m x m x (n+1).  Used by the state summary of the purely approximate chain (approx_mcmc::approx_state_summary,
 * src/approx_mcmc.cpp:1494-1520), which smooths the Gaussian approximation at every stored value. */
int bssm_gaussian_smoother(const bssm_ung_model* model, double* alphahat, double* Vt);

/* replaces _bssm_nongaussian_loglik: `double nongaussian_loglik(model_, nsim, sampling_method, seed,
 * model_type)` src/R_loglik.cpp:41-87.  nsim = 0 returns the Gaussian-approximation log-likelihood
 * (src/model_ssm_ung.cpp:214-234). */
int bssm_nongaussian_loglik(const bssm_ung_model* model, uint32_t nsim, uint32_t sampling_method, uint32_t seed,
                            int model_type, double* loglik);

/* Batched log-likelihoods: `n_rep` independent filters of the same model with Philox streams
 * stream0 .. stream0+n_rep-1 (what suggest_N / repeated logLik calls do one at a time,
 * src/R_postcorrection.cpp:24-35).  loglik has n_rep entries. */
int bssm_nongaussian_loglik_batch(const bssm_ung_model* model, uint32_t nsim, uint32_t sampling_method,
                                  uint32_t seed, uint32_t stream0, uint32_t n_rep, double* loglik);

/* ---- non-linear Gaussian (ssm_nlg) and SDE (ssm_sde) models ----
 * The reference takes user C++ compiled at run time behind function pointers (src/model_ssm_nlg.h:10-21,
 * src/model_ssm_sde.h:8-11); device code cannot call those, so the examples shipped in
 * R/cpp_example_models.R are built in and selected by id (theta / known_params keep the snippets' meaning).
 * Bootstrap filter for all six; psi-APF for the four ssm_nlg models (the reference has no psi-APF for ssm_sde). */
enum { BSSM_FN_NLG_GROWTH = 1, BSSM_FN_NLG_AR_EXP = 2, BSSM_FN_NLG_SIN_EXP = 3, BSSM_FN_NLG_LINEAR_GAUSSIAN = 4,
       BSSM_FN_SDE_GBM = 5, BSSM_FN_SDE_POISSON_OU = 6,
       BSSM_FN_USER_NLG = 7, BSSM_FN_USER_SDE = 8 /* user-defined, compiled at run time: bssm_user_model_build / _load */ };
typedef struct bssm_fn_model {
  int32_t model_id;
  int32_t n;                  /* time points */
  const double* y;            /* n (every built-in model is univariate), NaN = missing */
  const double* theta;   int32_t n_theta;
  const double* known_params; int32_t n_known;   /* nlg_growth: (dT, K, a11, a12, P11, P12) */
  double x0;                  /* ssm_sde: initial value */
  int32_t positive;           /* ssm_sde: reflect to stay positive (milstein.cpp:44) */
  int32_t L;                  /* ssm_sde: 2^L Milstein sub-steps per unit time */
  /* ---- user-defined models only (model_id BSSM_FN_USER_NLG / BSSM_FN_USER_SDE); ignored for the built-in ids */
  int32_t user_slot;          /* from bssm_user_model_load */
  int32_t n_states;           /* ssm_nlg: 1 ... 16 for the bootstrap filter / smoother / bootstrap IS correction and (round 6: Jacobians, constant R) the psi-APF, 1 ... 8 for the EKF particle filter */
  double H;                   /* ssm_nlg: observation sd (the value H_fn returns; unused when the snippet defines BSSM_USER_HFN) */
  double R_diag[2];           /* ssm_nlg: diagonal of R_fn's value (n_states <= 2, diagonal R) */
  double a1[2], P1[4];        /* ssm_nlg: a1_fn, P1_fn values (P1 column-major, n_states x n_states), n_states <= 2 */
  /* ---- wider user-defined ssm_nlg models (src/model_ssm_nlg.h:10-21 allows any m, k and matrix R): read instead of
   * R_diag / a1 / P1 when use_ext != 0; required for n_states > 2 */
  int32_t use_ext;
  int32_t n_etas;             /* columns of R, 1 ... n_states */
  double R_ext[256];          /* R_fn's value, n_states x n_etas column-major (the bootstrap kernels call the snippet's
                                 bssm_user_R instead when it defines BSSM_USER_RFN) */
  double a1_ext[16], P1_ext[256]; /* a1_fn, P1_fn values; P1 any positive semi-definite n_states x n_states matrix */
  /* ---- the reference's `known_tv_params` (R/models.R:1507, 1546; every model function receives it together with the
   * time index, src/model_ssm_nlg.h:10-21): any matrix, column-major, tv_rows x tv_cols.  Read by user models whose
   * snippet defines BSSM_USER_TV (its functions then take a leading `const bssm_user_tv&`); NULL / 0 / 0 otherwise */
  const double* known_tv_params;
  int32_t tv_rows, tv_cols;
} bssm_fn_model;

/* User-defined ssm_nlg / ssm_sde models.  The reference compiles the user's C++ at run time (Rcpp::sourceCpp) and
 * calls it through function pointers (src/model_ssm_nlg.h:10-21, src/model_ssm_sde.h:8-11); device code cannot call host
 * pointers, so the user's __device__ functions (contract: bssm_amd/csrc/pf_functors.h) are compiled by hipcc INTO the
 * filter kernels, one shared library per snippet, cached by content hash next to libbssm_amd.so (or in
 * $BSSM_AMD_USER_CACHE).  bssm_user_model_build compiles `source` for `n_states` (1 ... 16) and writes the library path;
 * bssm_user_model_load maps it and returns a slot >= 1 for bssm_fn_model.user_slot (negative = error).
 * Built for user models: every single-model entry point of the built-in models -- bootstrap, psi-APF (bssm_psi_smoother_nlg,
 * bssm_fn_psi_setup) and EKF particle filter (bssm_ekpf_*) with their smoothers and bssm_fn_loglik_batch; the psi-APF and
 * the EKF filter need the Jacobians in the snippet (BSSM_USER_JACOBIANS).  The psi-APF takes up to 16 states (round 6; 8 with a state-dependent R_fn), the EKF particle filter up to 8, any constant
 * n_states x n_etas matrix R whose R R' is positive definite on its non-zero diagonal, and any positive semi-definite P1
 * (round 3); so does the EKF particle filter (1- and 2-state models: constant diagonal R).  Snippets that define
 * BSSM_USER_TV receive the time index and known_tv_params in every function.  The bootstrap filter / smoother / bootstrap IS correction take up to 16 states (round 5), any n_states x n_etas matrix
 * R and the reference's state-dependent H_fn / R_fn (snippet functions bssm_user_H / bssm_user_R); so do the psi-APF and the
 * EKF particle filter (the approximating model then has H_t = H_fn(mode_t), R_t = R_fn(mode_t), src/model_ssm_nlg.cpp:100-107;
 * R(a) R(a)' must be positive definite on its non-zero diagonal, since the psi weight contains the transition density).  IS post-correction
 * (bssm_is_correction_nlg / _multi) and the chains (bssm_run_mcmc_fn; theta-dependent H / R / a1 / P1 through
 * bssm_mcmc_job.fn_update_fn) work for user models too. */
int bssm_user_model_build(const char* source, int32_t n_states, char* lib_path, int32_t lib_path_capacity);
/* The same snippet compiled IN PROCESS by hiprtc (the compiler as a library: no hipcc on the machine, no process, no shared library) --
 * what Rcpp::sourceCpp is to the reference's session (R/cpp_example_models.R:762).  Returns the registry slot (>= 1) directly.  Only the
 * kernel variant a call needs is compiled, the first time it is needed (1-2 s; the smallest workgroup-resident variant at once, so a
 * snippet with an error fails here with the compiler's log); code objects are cached on disk by content hash.  Such a slot runs EVERY
 * sampler since round 6: the bootstrap paths bit-identical to the hipcc-built library of the same snippet (same kernel templates, same
 * arithmetic); the psi-APF and the EKF particle filter evaluate the snippet during their host set-up through a one-thread evaluation kernel
 * compiled from it (~15 us per call: the device's libm instead of the host's, so the last place of the mode may differ from the hipcc route).
 * The same (source, n_states) registers once: a repeated call returns the slot it already has. */
int bssm_user_model_jit(const char* source, int32_t n_states);
int bssm_user_model_load(const char* lib_path);
/* replaces _bssm_bsf_nlg (18 args, src/R_bootstrap_filter.cpp:461-504) / _bssm_bsf_smoother_nlg (:508-551):
 * outputs as bssm_bsf / bssm_bsf_smoother */
int bssm_bsf_nlg(const bssm_fn_model* model, uint32_t nsim, uint32_t seed, bssm_filter_out* out);
int bssm_bsf_smoother_nlg(const bssm_fn_model* model, uint32_t nsim, uint32_t seed, bssm_filter_out* out);
/* replaces _bssm_bsf_sde / _bssm_bsf_smoother_sde (12 args, src/R_sde.cpp:35-130) */
int bssm_bsf_sde(const bssm_fn_model* model, uint32_t nsim, uint32_t seed, bssm_filter_out* out);
int bssm_bsf_smoother_sde(const bssm_fn_model* model, uint32_t nsim, uint32_t seed, bssm_filter_out* out);
/* replaces _bssm_R_milstein (9 args, src/R_milstein.cpp:4-22; src/RcppExports.cpp:1313): terminal value of one Milstein path from
 * model->x0 over [0, t_end] in 2^(model->L) steps -- the prior draw of particle 0 of a bootstrap filter with `seed` (t_end = 1 there) */
int bssm_milstein(const bssm_fn_model* model, double t_end, uint32_t seed, double* x_out);
/* replaces _bssm_nonlinear_loglik (method = 2, bsf) and _bssm_loglik_sde (src/R_loglik.cpp:91-124, src/R_sde.cpp:11-31),
 * batched over n_rep Philox streams */
int bssm_fn_loglik_batch(const bssm_fn_model* model, uint32_t nsim, uint32_t sampling_method, uint32_t seed, uint32_t stream0,
                         uint32_t n_rep, double* loglik);
/* the same with the mode search's max_iter / conv_tol (R: the model's `max_iter`, `conv_tol`) and, for the psi-APF, the
 * approximate log-likelihood of the very Gaussian approximation the particles were drawn from (*approx_loglik, may be
 * NULL): ll(1) of ssm_nlg::log_likelihood, src/model_ssm_nlg.cpp:253-254 */
int bssm_fn_loglik_batch_ex(const bssm_fn_model* model, uint32_t nsim, uint32_t sampling_method, uint32_t seed,
                            uint32_t stream0, uint32_t n_rep, int32_t max_iter, double conv_tol, double* loglik,
                            double* approx_loglik);
/* one smoother call (psi / bsf / ekf particle filter, ssm_nlg or ssm_sde) on Philox stream `stream0` instead of 0 */
int bssm_fn_smoother_stream(const bssm_fn_model* model, uint32_t sampling_method, uint32_t nsim, uint32_t seed,
                            uint32_t stream0, int32_t max_iter, double conv_tol, bssm_filter_out* out);
/* replaces _bssm_psi_smoother_nlg (21 args, src/R_psi.cpp:175-222): psi-APF + back-trace + summary of a built-in
 * ssm_nlg model.  max_iter / conv_tol are the mode-search controls of ssm_nlg (src/model_ssm_nlg.cpp:19-20). */
int bssm_psi_smoother_nlg(const bssm_fn_model* model, uint32_t nsim, uint32_t seed, int32_t max_iter, double conv_tol,
                          bssm_filter_out* out);
/* replaces _bssm_ekpf / _bssm_ekpf_smoother (18 args each, src/R_ekpf.cpp:11-100): the EKF-proposal particle filter
 * ssm_nlg::ekf_filter (src/model_ssm_nlg.cpp:1148-1293) of a built-in model; outputs as bssm_bsf / bssm_bsf_smoother.
 * Its log-likelihood alone: bssm_fn_loglik_batch with sampling_method = BSSM_EKPF. */
int bssm_ekpf_nlg(const bssm_fn_model* model, uint32_t nsim, uint32_t seed, bssm_filter_out* out);
int bssm_ekpf_smoother_nlg(const bssm_fn_model* model, uint32_t nsim, uint32_t seed, bssm_filter_out* out);
/* host-side psi set-up of ssm_nlg on its own (EKF, mode search, linearised model, conditional factors); is_mode m x n
 * selects approximate_for_is.  Outputs (any may be NULL): mode m x n, scales n, approx_loglik, alphahat m x (n+1),
 * Vt / Ct m x m x (n+1), linearised Z_t m x n, T_t m x m x n, D_t n, C_t m x n. */
int bssm_fn_psi_setup(const bssm_fn_model* model, const double* is_mode, int32_t max_iter, double conv_tol, double* mode,
                      double* scales, double* approx_loglik, double* alphahat, double* Vt, double* Ct, double* Zt, double* Tt,
                      double* Dt, double* Cc);
/* replaces _bssm_postcorrection_nonlinear (24 args, src/R_postcorrection.cpp:252-317): one model per retained draw
 * (only theta differs), modes m x n x count (psi) or approx_loglik[count] (bsf); other arguments and outputs as
 * bssm_is_correction.  Draws [first, first + count) of the global job. */
int bssm_is_correction_nlg(const bssm_fn_model* models, int32_t first, int32_t count, const double* modes,
                           const double* approx_loglik, const uint32_t* counts, int32_t sampling_method, int32_t is_type,
                           int32_t output_type, uint32_t nsim, uint32_t seed, int32_t n_threads, double* weights,
                           double* loglik, double* alpha, double* alphahat_i, double* Vt_i);

/* replaces _bssm_stratified_sample: `arma::uvec stratified_sample(arma::vec& p, const arma::vec& r, const unsigned int N)`
 * src/stratified_sample.cpp:9-28 (registered src/RcppExports.cpp:1275,1344).  p [n_elem] normalised weights, OVERWRITTEN
 * with their cumulative sum (last entry forced to 1) exactly as the reference does through its reference argument;
 * r [N] U[0,1) draws; out [N] zero-based ancestor indices.  The cumulative sum is evaluated in the reference's
 * sequential order, so the indices are bit-for-bit the reference's. */
int bssm_stratified_sample(double* p, const double* r, uint32_t n_elem, uint32_t N, uint32_t* out);

/* replaces _bssm_suggest_n_nongaussian (6 args, src/R_postcorrection.cpp:13-140) and _bssm_suggest_n_nonlinear
 * (20 args, src/R_postcorrection.cpp:107-148): standard deviation (N-1 normalisation) of the psi-APF
 * log-likelihood over `replications` runs for every candidate particle count.  `model` carries the theta-updated
 * components (what model.update_model(theta) leaves, :23,:41).  The replications of one candidate run as one device
 * batch on Philox streams i * replications + j.  sds [n_candidates]. */
int bssm_suggest_n_nongaussian(const bssm_ung_model* model, const uint32_t* candidates, int32_t n_candidates,
                               uint32_t replications, uint32_t seed, int model_type, double* sds);
int bssm_suggest_n_nonlinear(const bssm_fn_model* model, const uint32_t* candidates, int32_t n_candidates,
                             uint32_t replications, uint32_t seed, double* sds);

/* ---- IS post-correction ----
 * replaces _bssm_postcorrection_nongaussian: `postcorrection_nongaussian(model_, model_type, output_type,
 * nsim, seed, n_threads, is_type, counts, theta, modes)` src/R_postcorrection.cpp:150-248 and the drivers
 * approx_mcmc::is_correction_psi / _bsf (src/approx_mcmc.cpp:353-427, 447-518, 685-921).
 *
 * The theta-dependent model components are supplied pre-expanded, one bssm_ung_model per retained
 * draw -- exactly what parset_ung materialises before the parallel region (src/parset_ng.cpp:5-111);
 * bssm_update_model_builtin() below reproduces the built-in update_model() maps for
 * bsm_ng / svm / ar1_ng.  `first`/`count` select the contiguous block of draws this process owns
 * (theta-sharding across GPUs); random streams are keyed by the GLOBAL draw index, so the result
 * does not depend on how the draws are sharded. */
typedef struct bssm_is_job {
  int32_t n_stored;              /* global number of retained draws */
  int32_t first, count;          /* this process handles draws [first, first+count) */
  const bssm_ung_model* models;  /* `count` models (local) */
  const double* modes;           /* p x n x count signal modes (psi; the `modes` cube of R_postcorrection.cpp:158) or NULL */
  const double* approx_loglik;   /* count (bsf: src/approx_mcmc.cpp:479) or NULL */
  const uint32_t* counts;        /* count jump-chain multiplicities */
  int32_t sampling_method;       /* BSSM_PSI / BSSM_BSF */
  int32_t is_type;               /* 1..3 (is3 = caller pre-expanded, src/approx_mcmc.cpp:49-88) */
  int32_t output_type;           /* 1 full, 2 summary, 3 theta */
  uint32_t nsim;
  uint32_t seed;
  int32_t n_threads;             /* host threads for the per-draw psi set-up */
} bssm_is_job;

/* outputs (local block): weights[count], loglik[count] (log of the filter estimate),
 * output 1: alpha (n+1) x m x count; output 2: per-draw alphahat_i m x (n+1) x count and
 * Vt_i m x m x (n+1) x count -- the caller (or bssm_is_merge) folds them with weights*counts. */
int bssm_is_correction(const bssm_is_job* job, double* weights, double* loglik, double* alpha,
                       double* alphahat_i, double* Vt_i);

/* ---- one post-correction call over several GPUs of the node (single process) ----
 * The reference's postcorrection_nongaussian / postcorrection_nonlinear are ONE call whose `n_threads` argument spreads
 * the retained draws over OpenMP threads (src/approx_mcmc.cpp:364-369, `omp parallel for schedule(static)`;
 * src/R_postcorrection.cpp:150-158, 252-262).  The _multi entry points are that call for a node of GPUs: the draws are
 * cut into contiguous static blocks (balanced by sum(count_i) for is1), one host thread per block sets its device
 * (`devices[r]`, HIP ordinals; NULL / 0 = every visible device) and runs the single-device driver, and the per-draw
 * weights / log-likelihoods are exchanged by ONE ncclAllGather (RCCL over xGMI, communicators from ncclCommInitAll,
 * librccl opened lazily with dlopen).  The same ordinal may be listed more than once (logical shards on one GPU; the
 * exchange is then plain device copies).  Streams are keyed by the global draw index and every kernel family that can run a
 * draw evaluates the same arithmetic (csrc/pf_canon.h), so the outputs are bit-identical to the single-device call for any
 * device list -- also when a sparsely filled last round of a shard goes through the row family (csrc/pf_row_step.h; default
 * since round 5, BSSM_AMD_TAIL_SPLIT=0 keeps every draw workgroup-resident).  Verified on logical shards of one physical GPU
 * (tests/test_gpu_postcorrect.py, tests/test_gpu_fullsize.py).  `job->n_threads` host threads are divided between the shards.
 * Arguments and outputs otherwise as bssm_is_correction / bssm_is_correction_nlg. */
int bssm_is_correction_multi(const bssm_is_job* job, const int32_t* devices, int32_t n_devices, double* weights,
                             double* loglik, double* alpha, double* alphahat_i, double* Vt_i);
int bssm_is_correction_nlg_multi(const bssm_fn_model* models, int32_t first, int32_t count, const double* modes,
                                 const double* approx_loglik, const uint32_t* counts, int32_t sampling_method,
                                 int32_t is_type, int32_t output_type, uint32_t nsim, uint32_t seed, int32_t n_threads,
                                 const int32_t* devices, int32_t n_devices, double* weights, double* loglik, double* alpha,
                                 double* alphahat_i, double* Vt_i);
/* suggest_N sharded over several devices INSIDE one call (the reference runs candidates x replications on one thread,
 * src/R_postcorrection.cpp:24-35, 118-129): the replications of every candidate are cut into contiguous blocks, one per
 * device (same `devices` convention as above), each shard runs its replications of a candidate as one device batch on the
 * streams the single-device entry point uses, ONE exchange of the log-likelihood blocks at the end (RCCL all-gather when
 * the devices are distinct) -- results bit-identical for any device list.  loglik (n_candidates x replications,
 * replication fastest) may be NULL. */
int bssm_suggest_n_nongaussian_multi(const bssm_ung_model* model, const uint32_t* candidates, int32_t n_candidates,
                                     uint32_t replications, uint32_t seed, int model_type, const int32_t* devices,
                                     int32_t n_devices, double* sds, double* loglik);
int bssm_suggest_n_nonlinear_multi(const bssm_fn_model* model, const uint32_t* candidates, int32_t n_candidates,
                                   uint32_t replications, uint32_t seed, const int32_t* devices, int32_t n_devices,
                                   double* sds, double* loglik);
/* shards of the last _multi call on this thread: out[6 * r + ...] = device, first draw, draws, host set-up ms, device
 * phase ms, total ms of shard r (at most `capacity` shards written); *collective_ms = the exchange;
 * *gather_kind = 1 RCCL all-gather, 2 device copies, 3 host blocks (no device exchange
 * was possible: RCCL is an optimisation, the call does not fail after the filters have run), 0 nothing exchanged.  Returns the number of shards. */
int bssm_amd_last_is_shards(double* out, int32_t capacity, double* collective_ms, int32_t* gather_kind);
/* 1 when librccl could be opened and resolved (bssm_amd_last_error() says why not otherwise) */
int bssm_amd_rccl_available(void);

/* West/Chan merge of per-draw summaries in draw order (src/approx_mcmc.cpp:397-414, 422-424):
 * alphahat m x (n+1), Vt m x m x (n+1). Host-side, O(n_stored * n * m^2). */
int bssm_is_merge(int32_t n, int32_t m, int32_t n_stored, const double* weights, const uint32_t* counts,
                  const double* alphahat_i, const double* Vt_i, double* alphahat, double* Vt);

/* Built-in parameter maps: bsm_ng::update_model (src/model_bsm_ng.cpp:17-51), svm::update_model
 * (src/model_svm.cpp:12-29), ar1_ng::update_model (src/model_ar1_ng.cpp:12-36).  Writes into the
 * caller-owned (mutable) arrays the model points at. `flags` carries the model-type specific
 * switches: bsm_ng {slope, seasonal, noise, fixed[0..2], phi_est}, svm {svm_type},
 * ar1_ng {mu_est, phi_est}. */
int bssm_update_model_builtin(int model_type, const int32_t* flags, const double* theta, int32_t n_par,
                              bssm_ung_model* model, double* T, double* R, double* P1, double* a1, double* C,
                              double* beta);
/* The same for every stored draw at once -- what parset_ung does with update_fn before the parallel region of the
 * post-correction (src/parset_ng.cpp:5-72).  theta: n_par x count, one draw per column.  T, R, P1, a1, C, beta: room for
 * `count` consecutive copies of the model's arrays (their sizes follow from n, m, k, n_xreg and the time-variation flags
 * of `model`); each slot is filled with the model's array and then rewritten by the map of its draw; beta may be NULL when
 * n_xreg = 0.  phi[count] receives the dispersion of every draw.  `model` itself is not modified. */
int bssm_update_model_builtin_batch(int model_type, const int32_t* flags, const double* theta, int32_t n_par, int32_t count,
                                    const bssm_ung_model* model, double* T, double* R, double* P1, double* a1, double* C,
                                    double* beta, double* phi);

/* ---- psi-APF set-up on its own (host): approximate() + smoother_ccov + conditional_cov,
 * src/model_ssm_ung.cpp:87-161, 514-534.  is_mode (p x n) NULL = iterate from initial_mode.
 * Outputs: approx_y, approx_H, mode (p x n each), scales[n], approx_loglik, alphahat m x (n+1),
 * Vt / Ct m x m x (n+1). Any may be NULL. */
int bssm_psi_setup(const bssm_ung_model* model, const double* is_mode, double* approx_y, double* approx_H,
                   double* mode, double* scales, double* approx_loglik, double* alphahat, double* Vt, double* Ct);

/* ---- device-resident plan: repeated filtering of fixed models with nothing but the seed
 * changing (pseudo-marginal MCMC inner loop, benchmarks).  Outputs stay in HBM. ---- */
typedef struct bssm_plan bssm_plan;
#define BSSM_PLAN_KEEP_HISTORY 1u  /* alpha, indices, weights kept (full-output filter) */
#define BSSM_PLAN_FORCE_GRID 2u
#define BSSM_PLAN_FORCE_WG 4u
/* Opt-in (device-resident callers): keep the weights output in FACTORED form where the step kernel supports it --
 * column t of `weights` holds exp(w_i - max over i's particle block), written once, and one factor per (filter, time point,
 * particle block) takes it to the reference's exp(w_i - max_t) (src/model_ssm_ung.cpp:675-677):
 *   reference weight (t, i) = weights[f][t][i] * scales[f][t][i / block_particles]     for t < n; column n is final.
 * The launch that computes the weights then writes them once and nothing reads them back to rescale them: 16 of the 104
 * bytes a particle-step moves disappear.  bssm_plan_weights_layout() says whether the last run used it (the layout is
 * granted per run: only the weights-exchange step kernels produce it); bssm_plan_fetch() always returns the reference's
 * form.  The drop-in entry points (bssm_bsf, ...) never use it. */
#define BSSM_PLAN_FACTORED_WEIGHTS 8u
/* n_models filters x n_rep replicas each; model i, replica r uses stream stream0 + i*n_rep + r */
bssm_plan* bssm_plan_create(const bssm_ung_model* models, int32_t n_models, int32_t n_rep, uint32_t nsim,
                            uint32_t sampling_method, uint32_t flags);
void bssm_plan_destroy(bssm_plan* plan);
/* new parameter values for the same model structure, copied into the plan's existing device tables (no allocation):
 * the per-iteration step of pseudo-marginal / delayed-acceptance MCMC (src/mcmc.cpp:337-491, 530-685) */
int bssm_plan_update(bssm_plan* plan, const bssm_ung_model* models, void* hip_stream);
/* sample_or_summarise (src/summary.cpp:82-103) on the kept particle system of filter `filter` after bssm_plan_run:
 * alpha (n+1) x m = one uniformly drawn back-traced trajectory; alphahat m x (n+1), Vt m x m x (n+1) = smoothed summary.
 * Needs BSSM_PLAN_KEEP_HISTORY; any output may be NULL. */
int bssm_plan_sample_or_summarise(bssm_plan* plan, int32_t filter, uint32_t seed, void* hip_stream, double* alpha,
                                  double* alphahat, double* Vt);
/* enqueue one pass on `hip_stream` (a hipStream_t, may be NULL for the default stream); async */
int bssm_plan_run(bssm_plan* plan, uint32_t seed, uint32_t stream0, void* hip_stream);
/* wait for the stream and copy the n_models*n_rep log-likelihoods (constants included) */
int bssm_plan_loglik(bssm_plan* plan, void* hip_stream, double* loglik);
/* introspection for benchmarks: which kernel shape, its grid, algorithmic bytes per particle-step */
int bssm_plan_info(const bssm_plan* plan, int32_t* path /*0 wg, 1 grid*/, int32_t* n_filters, int32_t* blocks,
                   int32_t* threads, double* bytes_per_particle_step);
/* device pointers of the kept outputs (NULL when not kept): alpha_hist [f][n+1][m][N],
 * indices [f][n][N], weights [f][n+1][N] */
int bssm_plan_device_outputs(const bssm_plan* plan, void** alpha_hist, void** indices, void** weights);
/* layout of the weights the last bssm_plan_run left in device memory: *factored = 1 when they are in the factored form of
 * BSSM_PLAN_FACTORED_WEIGHTS, then *block_particles / *n_blocks give the particle blocks and *scales the device pointer of
 * the factors [f][n+1][n_blocks] (column n unused); otherwise 0 / 0 / 0 / NULL and `weights` holds the reference's form */
int bssm_plan_weights_layout(const bssm_plan* plan, int32_t* factored, int32_t* block_particles, int32_t* n_blocks, void** scales);
/* waits for `hip_stream` and copies the kept outputs of ONE filter of the plan to host memory, in the device layout:
 * alpha_hist (n+1) x m x N with the particle index fastest, indices n x N, weights (n+1) x N; any may be NULL
 * (parity tests and bench.py's accuracy leg read the very batch that was timed) */
int bssm_plan_fetch(const bssm_plan* plan, int32_t filter, void* hip_stream, double* alpha_hist, uint32_t* indices,
                    double* weights);
/* average device time per bssm_plan_run over `reps` runs, measured with HIP events on `hip_stream` */
int bssm_plan_time(bssm_plan* plan, uint32_t seed, int32_t reps, void* hip_stream, double* ms_per_run);

/* average duration of the DOMINANT kernel's launches during one run, measured with HIP events on
 * `hip_stream` around sampled launches (grid shape: every `sample_every`-th time-step launch;
 * workgroup shape: the single launch).  Used for the roofline line of bench.py. */
int bssm_plan_time_kernel(bssm_plan* plan, uint32_t seed, int32_t sample_every, void* hip_stream,
                          double* avg_kernel_ms, int32_t* n_samples);

/* test / diagnostic hooks: device elementary functions (fn 0 exp, 1 log, 2 sincos(2*pi*x) -> 2n outputs,
 * 3 Box-Muller pairs of the stream (seed 1, stream 0, purpose 2, t 0, particle i) -> 2n outputs) and the
 * cycle-stamp timeline of the grid step kernel (12 averaged phase durations). */
int bssm_debug_math(int fn, const double* in, double* out, int32_t n);
/* how many step-launch sequences the last bssm_plan_run / bssm_plan_time_kernel kept in flight (a batch that fills the chip more than
 * once is stepped as two interleaved sequences over the halves of its filters; each timed launch then covers half of them) */
int bssm_plan_launch_sequences(const bssm_plan* plan);
/* name, with template arguments, of the filter kernel variant the calling thread launched last (as the launcher chose it) */
int bssm_amd_last_kernel(char* buf, int32_t len);
/* filter batches launched so far, process-wide (any host thread), by kernel family: workgroup-resident, step kernels, row family
 * (pf_row_step.h: the tail round of a post-correction chunk) -- tests use it to see which family a call really took */
int bssm_amd_family_launches(uint64_t* wg, uint64_t* step, uint64_t* row);
/* runs one kernel that reads n_doubles and writes n_doubles (8 B per lane, coalesced): a known byte count for
 * calibrating the profiler's FETCH_SIZE / WRITE_SIZE counters on this library's access pattern */
int bssm_debug_stream_copy(uint64_t n_doubles);
/* diagnostic: workgroups of the conditional-factor kernels (time points >= 1, 16 and more padded states) that took the full decomposition / all of
 * them, since the last call (csrc/lg_factors_rows.h: the thin path) */
int bssm_debug_thin_counts(uint64_t out[2]);
int bssm_plan_debug_timeline(bssm_plan* plan, uint32_t seed, double* avg_delta_cycles);

/* approximate log-likelihood of each of the plan's filters (psi plans: the Gaussian-approximation term ll(1) of
 * log_likelihood(), src/model_ssm_ung.cpp:196-208; bsf plans have none and the call fails) */
int bssm_plan_approx_loglik(const bssm_plan* plan, double* approx_loglik);

/* ---- Markov chain Monte Carlo drivers: the callers of the path ----
 * replaces _bssm_nongaussian_pm_mcmc / _bssm_nongaussian_da_mcmc / _bssm_nongaussian_is_mcmc's chain stage:
 *   mcmc::pm_mcmc (src/mcmc.cpp:337-491), mcmc::da_mcmc (src/mcmc.cpp:530-685), approx_mcmc::amcmc
 *   (src/approx_mcmc.cpp:110-232), with ramcmc::adapt_S (CRAN ramcmc, Vihola 2012: S S' <- S (I + eta (a - a*) u u' / |u|^2) S',
 *   eta = min(1, d i^-gamma)) as the proposal adaptation.
 * The per-iteration likelihood of pm / da is one pass of a device-resident plan (bssm_plan_update + bssm_plan_run);
 * the approximate chain (mcmc_type 0) is host-only and is followed by bssm_is_correction() for the `is` types
 * (src/R_mcmc.cpp:331-470).
 *
 * Built-in models (bsm_ng / svm / ar1_ng) use the library's update_model maps and the reference's prior tables:
 * prior_distributions[n_par] (0 uniform, 1 half-normal, 2 normal, 3 truncated normal, 4 gamma) and prior_parameters
 * 4 x n_par column-major, as R/priors.R:246-261 builds them; log_prior_pdf follows src/model_bsm_ng.cpp:90-136,
 * src/model_svm.cpp:50-88, src/model_ar1_ng.cpp:63-118.  General ssm_ung / ssm_mng models supply the two callbacks the
 * reference takes as R closures (src/model_ssm_ung.cpp:41-82): update_fn rewrites the arrays the model points at (and
 * model->phi), prior_fn returns log p(theta). */
typedef double (*bssm_prior_fn)(const double* theta, int32_t n_par, void* user);
typedef int (*bssm_update_fn)(const double* theta, int32_t n_par, bssm_ung_model* model, void* user);
enum { BSSM_MCMC_APPROX = 0, BSSM_MCMC_PM = 1, BSSM_MCMC_DA = 2 };
typedef struct bssm_mcmc_job {
  int32_t model_type;                 /* BSSM_SSM_MNG .. BSSM_AR1_NG */
  const int32_t* flags;               /* built-in models: as bssm_update_model_builtin */
  int32_t n_par;
  const int32_t* prior_distributions; /* built-in models */
  const double* prior_parameters;
  bssm_update_fn update_fn;           /* general models (NULL for built-ins) */
  bssm_prior_fn prior_fn;
  void* user;
  int32_t mcmc_type;                  /* BSSM_MCMC_* */
  uint32_t sampling_method;           /* BSSM_PSI / BSSM_BSF (pm, da) */
  uint32_t nsim;
  uint64_t iter, burnin;
  uint32_t thin;
  double gamma, target_acceptance;
  int32_t end_ram;                    /* stop adapting after the burn-in */
  uint32_t seed;
  int32_t output_type;                /* pm / da: 1 one trajectory per stored value, 2 running summary, 3 theta only */
  int32_t max_iter;                   /* ssm_nlg mode search (src/model_ssm_nlg.cpp:19-20); ssm_ung models carry their own */
  double conv_tol;
  int32_t L_c, L_f;                   /* ssm_sde delayed acceptance (sde_da_mcmc, src/mcmc.cpp:840-960): coarse / fine levels; 0 otherwise */
  /* user-defined ssm_nlg models whose H, diag(R), a1, P1 depend on theta (the reference's H_fn / R_fn / a1_fn / P1_fn take
   * theta): called after every theta change to rewrite those fields of the chain's model copy; NULL = they are constants */
  int (*fn_update_fn)(const double* theta, int32_t n_par, bssm_fn_model* model, void* user);
} bssm_mcmc_job;
typedef struct bssm_mcmc_out {
  int32_t capacity;        /* in: room for this many stored values (floor((iter - burnin) / thin) + 1 always suffices) */
  int32_t n_stored;        /* out */
  double acceptance_rate;  /* out */
  double* S;               /* n_par x n_par lower triangular, in: initial proposal factor, out: adapted */
  double* theta;           /* n_par x capacity */
  uint32_t* counts;        /* capacity */
  double* posterior;       /* capacity: log prior + log-likelihood (approximate chain: + approximate log-likelihood) */
  double* approx_loglik;   /* capacity, approximate chain only (may be NULL) */
  double* prior;           /* capacity, approximate chain only (may be NULL) */
  double* modes;           /* (p n) x capacity, approximate chain only (may be NULL) */
  double* alpha;           /* (n+1) x m x capacity, pm / da with output_type 1 */
  double* alphahat;        /* m x (n+1), pm / da with output_type 2 */
  double* Vt;              /* m x m x (n+1), pm / da with output_type 2 */
} bssm_mcmc_out;
/* `model` carries the structure and the initial values; the chain works on its own copies of the theta-dependent
 * arrays (T, R, P1, a1, C, beta) for built-in models.  With callbacks the arrays `model` points at are the caller's and
 * update_fn rewrites them in place.  Errors: BSSM_ERR_INVALID for a non-finite initial prior / log-likelihood or when no
 * proposal was accepted (the reference stops with the same conditions, src/mcmc.cpp:352,363,485). */
int bssm_run_mcmc(bssm_ung_model* model, const double* theta0, const bssm_mcmc_job* job, bssm_mcmc_out* out);
/* The same chains for ssm_nlg / ssm_sde models (run_mcmc.ssm_nlg, R/run_mcmc.R:561-700 -> nonlinear_pm_mcmc / _da_mcmc /
 * _is_mcmc, src/R_mcmc.cpp:534-830; run_mcmc.ssm_sde -> sde_pm_mcmc, src/R_sde.cpp:134-170): theta IS the model's
 * parameter vector, prior_fn its log_prior_pdf; sampling_method BSSM_PSI / BSSM_BSF / BSSM_EKPF.  The approximate
 * log-likelihood is the mode-based one (ssm_nlg::log_likelihood with nsim = 0, src/model_ssm_nlg.cpp:258-277).
 * ssm_sde: pseudo-marginal chain with the bootstrap filter at the model's L, or the two-level delayed-acceptance chain
 * (first stage a bootstrap filter at 2^L_c sub-steps, second stage at 2^L_f; job->L_c, job->L_f).  The IS types of run_mcmc.ssm_sde are this
 * chain at the coarse level L_c (approx_mcmc::amcmc(ssm_sde), src/approx_mcmc.cpp:236-330; out->approx_loglik / prior are
 * filled for filtered chains too) followed by bssm_is_correction_nlg at the fine level L_f (src/approx_mcmc.cpp:523-600). */
int bssm_run_mcmc_fn(const bssm_fn_model* model, const double* theta0, const bssm_mcmc_job* job, bssm_mcmc_out* out);
/* the chain's own random numbers (tests): standard normals of iteration `iter` and its acceptance uniforms */
void bssm_mcmc_draws(uint32_t seed, uint64_t iter, int32_t n_par, double* normals, double* uniforms2);
/* log_prior_pdf of a built-in model (tests, host-only) */
double bssm_log_prior_builtin(int model_type, const int32_t* flags, int32_t n_xreg, const double* theta, int32_t n_par,
                              const int32_t* prior_distributions, const double* prior_parameters);

/* exposed for tests: the exact Philox block and uniform / normal pair the kernels use */
void bssm_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

#ifdef __cplusplus
}
#endif
#endif
