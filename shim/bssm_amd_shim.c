/* bssm_amd_shim.c -- the R-side binding: bssm's `.Call` entry points of the particle-filter / IS-MCMC path, re-pointed
 * at libbssm_amd.so (include/bssm_amd.h).
 *
 * STATUS: never met real R (neither this image nor the GPU box has R), but EXECUTED since round 4 against a mock of the R C
 * API (shim/mockr/: SEXP vectors / lists / attributes, closures, a PROTECT stack checked gctorture-style, Rf_error as a
 * non-local exit): tests/test_gpu_shim.py runs the registered entry points with model lists built field for field as the
 * reference's R constructors build them and requires the outputs of the ctypes path bit for bit, the protect stack back at
 * zero, R errors for hand-edited lists (tests/test_host_logic.py: registry and argument counts against the reference's
 * CallEntries, error paths without a GPU).  It is the complete source a maintainer drops into bssm's src/ in place of the Rcpp
 * stubs it names; R code does not change (R/RcppExports.R keeps calling `.Call('_bssm_bsf', PACKAGE = 'bssm', ...)`).
 *
 * What it replaces: the rows of `CallEntries` in /root/reference/src/RcppExports.cpp:1287-1351 that belong to the
 * path, each stub citing the wrapper whose argument order and return list it reproduces.  Everything is
 * plain R C API (Rinternals.h): no Rcpp, no Armadillo.  Inputs are read in place (R holds column-major doubles, which
 * is what the C ABI takes); outputs are allocated as R arrays and filled by the library.
 *
 * ssm_nlg / ssm_sde: the reference receives external pointers to run-time compiled HOST functions
 * (src/model_ssm_nlg.h:10-21); device code cannot call those.  The shim keeps the .Call signatures and reads the model
 * identity from attributes of the first pointer argument, which the R constructor attaches:
 *   attr(Z, "bssm_amd_model_id")     integer, BSSM_FN_* (the shipped example models are built in), or
 *   attr(Z, "bssm_amd_device_code")  character, the user's __device__ functions (contract: bssm_amd/csrc/pf_functors.h),
 *                                    with numeric attributes "bssm_amd_H", "bssm_amd_R_diag" (or the matrix "bssm_amd_R"), "bssm_amd_a1", "bssm_amd_P1".
 * Link: PKG_LIBS = -L$(BSSM_AMD) -lbssm_amd -Wl,-rpath,$(BSSM_AMD)
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "bssm_amd.h"

/* ------------------------------------------------------------------ small helpers */
static SEXP elt(SEXP list, const char* name) {
  SEXP names = Rf_getAttrib(list, R_NamesSymbol);
  for (R_xlen_t i = 0; i < Rf_xlength(list); i++)
    if (!strcmp(CHAR(STRING_ELT(names, i)), name)) return VECTOR_ELT(list, i);
  return R_NilValue;
}
static int has_elt(SEXP list, const char* name) { return elt(list, name) != R_NilValue; }
static int dim_at(SEXP x, int which) { /* extent `which` (0-based), 1 if absent */
  SEXP d = Rf_getAttrib(x, R_DimSymbol);
  return (d == R_NilValue || Rf_length(d) <= which) ? 1 : INTEGER(d)[which];
}
static double* dup_real(SEXP x) { /* private, writable copy (freed by R at the end of the .Call) */
  const R_xlen_t len = Rf_xlength(x);
  double* p = (double*)R_alloc((size_t)(len > 0 ? len : 1), sizeof(double));
  if (len > 0) memcpy(p, REAL(x), (size_t)len * sizeof(double));
  return p;
}
static void check_rc(int rc) { /* BEGIN_RCPP / END_RCPP turn exceptions into R errors, src/RcppExports.cpp:58,68 */
  if (rc < 0) Rf_error("%s", bssm_amd_last_error());
}
static const char* kDegenerate = "Particle filtering stopped prematurely due to nonfinite log-likelihood."; /* src/R_bootstrap_filter.cpp:140-141 */

/* ------------------------------------------------------------------ model list -> bssm_ung_model
 * field by field the constructors src/model_ssm_ung.cpp:9-32, src/model_ssm_mng.cpp:7-31 and (gaussian) src/model_ssm_ulg.cpp:8-38 */
typedef struct {
  bssm_ung_model mod;
  int32_t* dist_p; /* ssm_mng: per-series distribution codes as int32 */
} shim_model;

static void check_model_extents(SEXP m, const shim_model* o, int gaussian);
static void unpack_model(SEXP m, int model_type, int gaussian, shim_model* o) {
  SEXP Z = elt(m, "Z"), T = elt(m, "T"), R = elt(m, "R"), D = elt(m, "D"), C = elt(m, "C"), xreg = elt(m, "xreg");
  bssm_ung_model* q = &o->mod;
  memset(o, 0, sizeof(*o));
  q->a1 = REAL(elt(m, "a1"));
  q->m = Rf_length(elt(m, "a1"));
  q->T = REAL(T);  q->Ttv = dim_at(T, 2) > 1;
  q->R = REAL(R);  q->Rtv = dim_at(R, 2) > 1;  q->k = dim_at(R, 1);
  q->P1 = REAL(elt(m, "P1"));
  q->C = REAL(C);  q->Ctv = dim_at(C, 1) > 1;
  q->y = REAL(elt(m, "y"));
  if (!gaussian && model_type == BSSM_SSM_MNG) { /* y n x p, Z p x m x (1|n), D p x (1|n), u n x p, phi[p], distribution[p] */
    SEXP dist = elt(m, "distribution");
    q->n = dim_at(elt(m, "y"), 0);
    q->p = dim_at(elt(m, "y"), 1);
    q->Z = REAL(Z);  q->Ztv = dim_at(Z, 2) > 1;
    q->D = REAL(D);  q->Dtv = dim_at(D, 1) > 1;
    q->u = REAL(elt(m, "u"));
    q->phi_p = REAL(elt(m, "phi"));
    o->dist_p = (int32_t*)R_alloc((size_t)q->p, sizeof(int32_t));
    for (int j = 0; j < q->p; j++) o->dist_p[j] = TYPEOF(dist) == INTSXP ? INTEGER(dist)[j] : (int32_t)REAL(dist)[j];
    q->distribution_p = o->dist_p;
  } else {
    q->n = Rf_length(elt(m, "y"));
    q->p = 1;
    q->Z = REAL(Z);  q->Ztv = dim_at(Z, 1) > 1;        /* Z.n_cols > 1, src/model_ssm_ung.cpp:15 */
    q->D = REAL(D);  q->Dtv = Rf_length(D) > 1;
    q->n_xreg = Rf_length(xreg) > 0 ? dim_at(xreg, 1) : 0;
    q->xreg = q->n_xreg ? REAL(xreg) : NULL;
    q->beta = q->n_xreg ? REAL(elt(m, "beta")) : NULL;
    if (gaussian) {
      q->distribution = BSSM_GAUSSIAN;
      q->H = REAL(elt(m, "H"));  q->Htv = Rf_length(elt(m, "H")) > 1;
    } else {
      q->u = REAL(elt(m, "u"));
      q->phi = Rf_asReal(elt(m, "phi"));
      q->distribution = Rf_asInteger(elt(m, "distribution")); /* already pmatch(...) - 1 on the R side */
      if (q->distribution == BSSM_SVM) q->u = NULL; /* svm() stores u = 1 (R/models.R:1231), the density never reads it: the library then takes ones */
    }
  }
  if (!gaussian) {
    q->max_iter = Rf_asInteger(elt(m, "max_iter"));
    q->conv_tol = Rf_asReal(elt(m, "conv_tol"));
    q->local_approx = Rf_asLogical(elt(m, "local_approx"));
    q->initial_mode = REAL(elt(m, "initial_mode"));
  }
  check_model_extents(m, o, gaussian);
}

/* The C ABI takes bare pointers, so the extents the library will read are checked HERE against the lengths R holds (the Rcpp
 * constructors get them from the arma objects; a list edited by hand must fail with an R error, not read out of bounds) */
static void need_len(SEXP m, const char* name, R_xlen_t a, R_xlen_t b) {
  const R_xlen_t got = Rf_xlength(elt(m, name));
  if (got != a && got != b) {
    if (a == b) Rf_error("model$%s has length %ld, expected %ld", name, (long)got, (long)a);
    Rf_error("model$%s has length %ld, expected %ld or %ld", name, (long)got, (long)a, (long)b);
  }
}
static void check_model_extents(SEXP m, const shim_model* o, int gaussian) {
  const bssm_ung_model* q = &o->mod;
  const R_xlen_t n = q->n, ms = q->m, k = q->k, p = q->p;
  if (n < 1 || ms < 1 || k < 1) Rf_error("model with an empty series, state or noise dimension (n = %ld, m = %ld, k = %ld)", (long)n, (long)ms, (long)k);
  need_len(m, "T", ms * ms, ms * ms * n);
  need_len(m, "R", ms * k, ms * k * n);
  need_len(m, "P1", ms * ms, ms * ms);
  need_len(m, "C", ms, ms * n);
  need_len(m, "y", n * p, n * p);
  need_len(m, "Z", p * ms, p * ms * n);
  need_len(m, "D", p, p * n);
  if (gaussian) { need_len(m, "H", 1, n); return; }
  if (q->u) need_len(m, "u", n * p, n * p);   /* (svm carries u = 1, R/models.R:1231: not handed over) */
  need_len(m, "initial_mode", n * p, n * p);
  if (p > 1) { need_len(m, "phi", p, p); need_len(m, "distribution", p, p); }
  if (q->n_xreg) { need_len(m, "xreg", n * q->n_xreg, n * q->n_xreg); need_len(m, "beta", q->n_xreg, q->n_xreg); }
}

/* theta-dependent components get private copies so that update_model can rewrite them */
static void own_theta_dependent(SEXP m, shim_model* o) {
  bssm_ung_model* q = &o->mod;
  q->T = dup_real(elt(m, "T"));   q->R = dup_real(elt(m, "R"));   q->P1 = dup_real(elt(m, "P1"));
  q->a1 = dup_real(elt(m, "a1")); q->C = dup_real(elt(m, "C"));   q->Z = dup_real(elt(m, "Z"));
  q->D = dup_real(elt(m, "D"));
  if (q->n_xreg) q->beta = dup_real(elt(m, "beta"));
  if (q->p > 1) q->phi_p = dup_real(elt(m, "phi"));
}

/* flags of the built-in update_model maps: src/model_bsm_ng.cpp:4-14, src/model_svm.cpp:5-10, src/model_ar1_ng.cpp:4-10 */
static int builtin_flags(SEXP m, int model_type, int32_t flags[8]) {
  memset(flags, 0, 8 * sizeof(int32_t));
  if (model_type == BSSM_BSM_NG) {
    SEXP fx = PROTECT(Rf_coerceVector(elt(m, "fixed"), INTSXP));
    flags[0] = Rf_asLogical(elt(m, "slope")); flags[1] = Rf_asLogical(elt(m, "seasonal")); flags[2] = Rf_asLogical(elt(m, "noise"));
    flags[3] = INTEGER(fx)[0]; flags[4] = INTEGER(fx)[1]; flags[5] = INTEGER(fx)[2];
    flags[6] = Rf_asLogical(elt(m, "phi_est"));
    UNPROTECT(1);
    return 1;
  }
  if (model_type == BSSM_SVM_MODEL) { flags[0] = Rf_asInteger(elt(m, "svm_type")); return 1; }
  if (model_type == BSSM_AR1_NG) { flags[0] = Rf_asLogical(elt(m, "mu_est")); flags[1] = Rf_asLogical(elt(m, "phi_est")); return 1; }
  return 0;
}

/* ssm_ung::update_model / ssm_mng::update_model through the R closure (src/model_ssm_ung.cpp:41-78): the closure returns a
 * list of the components that depend on theta; each one present overwrites the model's array of the same shape */
static void copy_component(SEXP upd, const char* name, const double* dst_const, R_xlen_t len) {
  SEXP v = elt(upd, name);
  if (v == R_NilValue) return;
  SEXP r = PROTECT(Rf_coerceVector(v, REALSXP));
  if (Rf_xlength(r) != len) Rf_error("update_fn returned '%s' of length %ld, expected %ld", name, (long)Rf_xlength(r), (long)len);
  memcpy((double*)dst_const, REAL(r), (size_t)len * sizeof(double));
  UNPROTECT(1);
}
static void apply_closure_update(SEXP update_fn, const double* theta, int n_par, SEXP model_, bssm_ung_model* q) {
  SEXP th = PROTECT(Rf_allocVector(REALSXP, n_par));
  memcpy(REAL(th), theta, (size_t)n_par * sizeof(double));
  SEXP call = PROTECT(Rf_lang2(update_fn, th));
  SEXP upd = PROTECT(Rf_eval(call, R_GlobalEnv));
  copy_component(upd, "Z", q->Z, Rf_xlength(elt(model_, "Z")));
  copy_component(upd, "T", q->T, Rf_xlength(elt(model_, "T")));
  copy_component(upd, "R", q->R, Rf_xlength(elt(model_, "R")));
  copy_component(upd, "a1", q->a1, Rf_xlength(elt(model_, "a1")));
  copy_component(upd, "P1", q->P1, Rf_xlength(elt(model_, "P1")));
  copy_component(upd, "D", q->D, Rf_xlength(elt(model_, "D")));
  copy_component(upd, "C", q->C, Rf_xlength(elt(model_, "C")));
  if (q->n_xreg) copy_component(upd, "beta", q->beta, q->n_xreg);
  if (has_elt(upd, "phi")) {
    if (q->p > 1) copy_component(upd, "phi", q->phi_p, q->p);
    else q->phi = Rf_asReal(elt(upd, "phi"));
  }
  UNPROTECT(3);
}
static void apply_theta(SEXP model_, int model_type, const double* theta, int n_par, shim_model* o) {
  int32_t flags[8];
  if (builtin_flags(model_, model_type, flags)) {
    bssm_ung_model* q = &o->mod;
    check_rc(bssm_update_model_builtin(model_type, flags, theta, n_par, q, (double*)q->T, (double*)q->R, (double*)q->P1,
                                       (double*)q->a1, (double*)q->C, (double*)q->beta));
  } else {
    apply_closure_update(elt(model_, "update_fn"), theta, n_par, model_, &o->mod);
  }
}

/* ------------------------------------------------------------------ result lists */
static SEXP filter_result(SEXP at, SEXP att, SEXP Pt, SEXP Ptt, SEXP w, double loglik, SEXP alpha) {
  const char* nm[] = {"at", "att", "Pt", "Ptt", "weights", "logLik", "alpha", ""};  /* src/R_bootstrap_filter.cpp:151-155 */
  SEXP res = PROTECT(Rf_mkNamed(VECSXP, nm));
  SET_VECTOR_ELT(res, 0, at); SET_VECTOR_ELT(res, 1, att); SET_VECTOR_ELT(res, 2, Pt); SET_VECTOR_ELT(res, 3, Ptt);
  SET_VECTOR_ELT(res, 4, w); SET_VECTOR_ELT(res, 5, Rf_ScalarReal(loglik)); SET_VECTOR_ELT(res, 6, alpha);
  UNPROTECT(1);
  return res;
}
static SEXP smoother_result(SEXP alphahat, SEXP Vt, SEXP w, double loglik, SEXP alpha) {
  const char* nm[] = {"alphahat", "Vt", "weights", "logLik", "alpha", ""};  /* src/R_bootstrap_filter.cpp:374-377 */
  SEXP res = PROTECT(Rf_mkNamed(VECSXP, nm));
  SET_VECTOR_ELT(res, 0, alphahat); SET_VECTOR_ELT(res, 1, Vt); SET_VECTOR_ELT(res, 2, w);
  SET_VECTOR_ELT(res, 3, Rf_ScalarReal(loglik)); SET_VECTOR_ELT(res, 4, alpha);
  UNPROTECT(1);
  return res;
}
typedef int (*filter_call)(const void* model, uint32_t nsim, uint32_t seed, bssm_filter_out* out, const void* extra);

/* allocate the outputs of a filter (`smoother` = 0) or smoother call, run it, wrap the list */
static SEXP run_filter_like(int n, int m, uint32_t N, uint32_t seed, int smoother, filter_call fn, const void* model, const void* extra) {
  bssm_filter_out out;
  memset(&out, 0, sizeof(out));
  SEXP w = PROTECT(Rf_allocMatrix(REALSXP, (int)N, n + 1)), alpha = PROTECT(Rf_alloc3DArray(REALSXP, m, n + 1, (int)N));
  SEXP a = R_NilValue, b = R_NilValue, c = R_NilValue, d = R_NilValue, res;
  out.weights = REAL(w); out.alpha = REAL(alpha);
  if (smoother) {
    a = PROTECT(Rf_allocMatrix(REALSXP, n + 1, m)); b = PROTECT(Rf_alloc3DArray(REALSXP, m, m, n + 1));
    out.alphahat = REAL(a); out.Vt = REAL(b);
  } else {
    a = PROTECT(Rf_allocMatrix(REALSXP, n + 1, m)); b = PROTECT(Rf_allocMatrix(REALSXP, n, m));
    c = PROTECT(Rf_alloc3DArray(REALSXP, m, m, n + 1)); d = PROTECT(Rf_alloc3DArray(REALSXP, m, m, n));
    out.at = REAL(a); out.att = REAL(b); out.Pt = REAL(c); out.Ptt = REAL(d);
  }
  const int rc = fn(model, N, seed, &out, extra);
  check_rc(rc);
  if (rc == BSSM_WARN_DEGENERATE) Rf_warning("%s", kDegenerate);
  res = smoother ? smoother_result(a, b, w, out.logLik, alpha) : filter_result(a, b, c, d, w, out.logLik, alpha);
  UNPROTECT(smoother ? 4 : 6);
  return res;
}

/* ================================================================== non-Gaussian / Gaussian state space models */
typedef struct { int gaussian, model_type; } ung_extra;
static int call_bsf(const void* mod, uint32_t N, uint32_t seed, bssm_filter_out* out, const void* ex) {
  const ung_extra* e = (const ung_extra*)ex;
  return bssm_bsf((const bssm_ung_model*)mod, N, seed, e->gaussian, e->model_type, out);
}
static int call_bsf_smoother(const void* mod, uint32_t N, uint32_t seed, bssm_filter_out* out, const void* ex) {
  const ung_extra* e = (const ung_extra*)ex;
  return bssm_bsf_smoother((const bssm_ung_model*)mod, N, seed, e->gaussian, e->model_type, out);
}
static int call_psi_smoother(const void* mod, uint32_t N, uint32_t seed, bssm_filter_out* out, const void* ex) {
  return bssm_psi_smoother((const bssm_ung_model*)mod, N, seed, ((const ung_extra*)ex)->model_type, out);
}

/* .Call('_bssm_bsf', model_, nsim, seed, gaussian, model_type): src/R_bootstrap_filter.cpp:15-241 */
SEXP _bssm_bsf(SEXP model_, SEXP nsim, SEXP seed, SEXP gaussian, SEXP model_type) {
  shim_model sm;
  ung_extra ex = {Rf_asLogical(gaussian), Rf_asInteger(model_type)};
  unpack_model(model_, ex.model_type, ex.gaussian, &sm);
  return run_filter_like(sm.mod.n, sm.mod.m, (uint32_t)Rf_asInteger(nsim), (uint32_t)Rf_asInteger(seed), 0, call_bsf, &sm.mod, &ex);
}
/* .Call('_bssm_bsf_smoother', ...): src/R_bootstrap_filter.cpp:245-457 */
SEXP _bssm_bsf_smoother(SEXP model_, SEXP nsim, SEXP seed, SEXP gaussian, SEXP model_type) {
  shim_model sm;
  ung_extra ex = {Rf_asLogical(gaussian), Rf_asInteger(model_type)};
  unpack_model(model_, ex.model_type, ex.gaussian, &sm);
  return run_filter_like(sm.mod.n, sm.mod.m, (uint32_t)Rf_asInteger(nsim), (uint32_t)Rf_asInteger(seed), 1, call_bsf_smoother, &sm.mod, &ex);
}
/* .Call('_bssm_psi_smoother', model_, nsim, seed, model_type): src/R_psi.cpp:52-171 */
SEXP _bssm_psi_smoother(SEXP model_, SEXP nsim, SEXP seed, SEXP model_type) {
  shim_model sm;
  ung_extra ex = {0, Rf_asInteger(model_type)};
  unpack_model(model_, ex.model_type, 0, &sm);
  return run_filter_like(sm.mod.n, sm.mod.m, (uint32_t)Rf_asInteger(nsim), (uint32_t)Rf_asInteger(seed), 1, call_psi_smoother, &sm.mod, &ex);
}
/* .Call('_bssm_gaussian_psi_smoother', model_, nsim, seed, model_type): src/R_psi.cpp:14-50; returns the cube m x (n+1) x nsim */
SEXP _bssm_gaussian_psi_smoother(SEXP model_, SEXP nsim, SEXP seed, SEXP model_type) {
  shim_model sm;
  const int mt = Rf_asInteger(model_type), N = Rf_asInteger(nsim);
  unpack_model(model_, mt, 1, &sm);
  if (mt == 0) { /* ssm_mlg: y n x p, Z p x m x (1|n), H p x p x (1|n) */
    sm.mod.n = dim_at(elt(model_, "y"), 0); sm.mod.p = dim_at(elt(model_, "y"), 1);
    sm.mod.Ztv = dim_at(elt(model_, "Z"), 2) > 1; sm.mod.Dtv = dim_at(elt(model_, "D"), 1) > 1; sm.mod.Htv = dim_at(elt(model_, "H"), 2) > 1;
    sm.mod.n_xreg = 0; sm.mod.xreg = NULL; sm.mod.beta = NULL;
  }
  SEXP alpha = PROTECT(Rf_alloc3DArray(REALSXP, sm.mod.m, sm.mod.n + 1, N));
  check_rc(bssm_gaussian_psi_smoother(&sm.mod, (uint32_t)N, (uint32_t)Rf_asInteger(seed), mt, REAL(alpha)));
  UNPROTECT(1);
  return alpha;
}
/* .Call('_bssm_nongaussian_loglik', model_, nsim, sampling_method, seed, model_type): src/R_loglik.cpp:41-87 */
SEXP _bssm_nongaussian_loglik(SEXP model_, SEXP nsim, SEXP sampling_method, SEXP seed, SEXP model_type) {
  shim_model sm;
  double ll = R_NegInf;
  unpack_model(model_, Rf_asInteger(model_type), 0, &sm);
  check_rc(bssm_nongaussian_loglik(&sm.mod, (uint32_t)Rf_asInteger(nsim), (uint32_t)Rf_asInteger(sampling_method),
                                   (uint32_t)Rf_asInteger(seed), Rf_asInteger(model_type), &ll));
  return Rf_ScalarReal(ll);
}

/* .Call('_bssm_stratified_sample', p, r, N): src/stratified_sample.cpp:9-28 (registered src/RcppExports.cpp:1275-1285).
 * The reference overwrites its argument with the cumulative sum; Rcpp hands it a copy of the R vector, so does the shim. */
SEXP _bssm_stratified_sample(SEXP pS, SEXP rS, SEXP NS) {
  const unsigned N = (unsigned)Rf_asInteger(NS);
  SEXP p = PROTECT(Rf_duplicate(pS)), out = PROTECT(Rf_allocVector(REALSXP, N)); /* arma::uvec wraps to a numeric vector */
  uint32_t* idx = (uint32_t*)R_alloc(N ? N : 1, sizeof(uint32_t));
  check_rc(bssm_stratified_sample(REAL(p), REAL(rS), (uint32_t)Rf_length(p), N, idx));
  for (unsigned j = 0; j < N; j++) REAL(out)[j] = idx[j];
  UNPROTECT(2);
  return out;
}

/* .Call('_bssm_suggest_n_nongaussian', model_, theta, candidates, replications, seed, model_type): src/R_postcorrection.cpp:13-105 */
SEXP _bssm_suggest_n_nongaussian(SEXP model_, SEXP theta, SEXP candidates, SEXP replications, SEXP seed, SEXP model_type) {
  shim_model sm;
  const int mt = Rf_asInteger(model_type), nc = Rf_length(candidates);
  unpack_model(model_, mt, 0, &sm);
  own_theta_dependent(model_, &sm);
  apply_theta(model_, mt, REAL(theta), Rf_length(theta), &sm);  /* model.update_model(theta), :23,:41,:59 */
  uint32_t* cand = (uint32_t*)R_alloc((size_t)(nc ? nc : 1), sizeof(uint32_t));
  for (int i = 0; i < nc; i++) cand[i] = (uint32_t)REAL(candidates)[i];
  SEXP sds = PROTECT(Rf_allocVector(REALSXP, nc));
  /* replications spread over the GPUs of the node inside the call (devices NULL = all visible), one exchange at the end */
  check_rc(bssm_suggest_n_nongaussian_multi(&sm.mod, cand, nc, (uint32_t)Rf_asInteger(replications), (uint32_t)Rf_asInteger(seed), mt, NULL, 0,
                                            REAL(sds), NULL));
  UNPROTECT(1);
  return sds;
}

/* ------------------------------------------------------------------ per-draw models (parset_ung / parset_mng, src/parset_ng.cpp:5-111)
 * The R closure is evaluated serially for every stored theta BEFORE any parallel work, exactly where the reference
 * builds its parset; built-in classes use the library's update_model maps.  theta: n_par x n_stored. */
static bssm_ung_model* expand_models(SEXP model_, int model_type, const double* theta, int n_par, int n_stored) {
  bssm_ung_model* models = (bssm_ung_model*)R_alloc((size_t)(n_stored ? n_stored : 1), sizeof(bssm_ung_model));
  for (int i = 0; i < n_stored; i++) {
    shim_model sm;
    unpack_model(model_, model_type, 0, &sm);
    own_theta_dependent(model_, &sm);
    apply_theta(model_, model_type, theta + (size_t)n_par * i, n_par, &sm);
    models[i] = sm.mod;
  }
  return models;
}

/* expand(): one row per jump-chain repetition (src/approx_mcmc.cpp:49-88), for is_type 3 */
typedef struct { int n; double* theta; double* modes; double* approx_loglik; double* prior; uint32_t* counts; } expanded;
static expanded expand_is3(const double* theta, int n_par, const double* modes, size_t mode_len, const double* al, const double* pr,
                           const uint32_t* counts, int n_stored) {
  expanded e;
  size_t tot = 0;
  for (int i = 0; i < n_stored; i++) tot += counts[i];
  e.n = (int)tot;
  e.theta = (double*)R_alloc(tot * (size_t)n_par + 1, sizeof(double));
  e.modes = (double*)R_alloc(tot * mode_len + 1, sizeof(double));
  e.approx_loglik = (double*)R_alloc(tot + 1, sizeof(double));
  e.prior = (double*)R_alloc(tot + 1, sizeof(double));
  e.counts = (uint32_t*)R_alloc(tot + 1, sizeof(uint32_t));
  size_t q = 0;
  for (int i = 0; i < n_stored; i++)
    for (uint32_t r = 0; r < counts[i]; r++, q++) {
      memcpy(e.theta + q * n_par, theta + (size_t)i * n_par, (size_t)n_par * sizeof(double));
      memcpy(e.modes + q * mode_len, modes + (size_t)i * mode_len, mode_len * sizeof(double));
      e.approx_loglik[q] = al ? al[i] : 0.0;
      e.prior[q] = pr ? pr[i] : 0.0;
      e.counts[q] = 1u;
    }
  return e;
}

/* IS correction of prepared draws on every GPU of the node (the reference spreads the same loop over `n_threads` OpenMP
 * threads, src/approx_mcmc.cpp:364-369), outputs wrapped as the reference returns them */
typedef struct { SEXP weights, alpha, alphahat, Vt; double* loglik; } is_out;
static is_out run_is_ung(const bssm_ung_model* models, int n_stored, const double* modes, const double* approx_loglik,
                         const uint32_t* counts, int sampling_method, int is_type, int output_type, uint32_t nsim, uint32_t seed,
                         int n_threads, int* n_protect) {
  is_out o = {R_NilValue, R_NilValue, R_NilValue, R_NilValue, NULL};
  const int n = models[0].n, m = models[0].m;
  bssm_is_job job;
  memset(&job, 0, sizeof(job));
  job.n_stored = n_stored; job.first = 0; job.count = n_stored; job.models = models; job.modes = modes;
  job.approx_loglik = approx_loglik; job.counts = counts; job.sampling_method = sampling_method; job.is_type = is_type;
  job.output_type = output_type; job.nsim = nsim; job.seed = seed; job.n_threads = n_threads;
  o.weights = PROTECT(Rf_allocVector(REALSXP, n_stored)); (*n_protect)++;
  o.loglik = (double*)R_alloc((size_t)n_stored + 1, sizeof(double));
  double *alpha = NULL, *ah_i = NULL, *vt_i = NULL;
  if (output_type == 1) { o.alpha = PROTECT(Rf_alloc3DArray(REALSXP, n + 1, m, n_stored)); (*n_protect)++; alpha = REAL(o.alpha); }
  if (output_type == 2) {
    ah_i = (double*)R_alloc((size_t)m * (n + 1) * n_stored + 1, sizeof(double));
    vt_i = (double*)R_alloc((size_t)m * m * (n + 1) * n_stored + 1, sizeof(double));
  }
  const int rc = bssm_is_correction_multi(&job, NULL, 0, REAL(o.weights), o.loglik, alpha, ah_i, vt_i);
  check_rc(rc);
  if (output_type == 2) { /* weighted running mean / covariance in draw order, src/approx_mcmc.cpp:397-414, 422-424 */
    SEXP ah = PROTECT(Rf_allocMatrix(REALSXP, m, n + 1)); (*n_protect)++;
    o.Vt = PROTECT(Rf_alloc3DArray(REALSXP, m, m, n + 1)); (*n_protect)++;
    check_rc(bssm_is_merge(n, m, n_stored, REAL(o.weights), counts, ah_i, vt_i, REAL(ah), REAL(o.Vt)));
    o.alphahat = PROTECT(Rf_allocMatrix(REALSXP, n + 1, m)); (*n_protect)++;  /* mcmc_run.alphahat.t() */
    for (int t = 0; t <= n; t++) for (int c = 0; c < m; c++) REAL(o.alphahat)[t + (size_t)(n + 1) * c] = REAL(ah)[c + (size_t)m * t];
  }
  return o;
}

static uint32_t* counts_u32(SEXP counts) {
  const int ns = Rf_length(counts);
  uint32_t* c = (uint32_t*)R_alloc((size_t)ns + 1, sizeof(uint32_t));
  for (int i = 0; i < ns; i++) c[i] = TYPEOF(counts) == INTSXP ? (uint32_t)INTEGER(counts)[i] : (uint32_t)REAL(counts)[i];
  return c;
}

/* .Call('_bssm_postcorrection_nongaussian', model_, model_type, output_type, nsim, seed, n_threads, is_type, counts,
 *       theta, modes): src/R_postcorrection.cpp:150-248.  theta n_par x n_stored, modes p x n x n_stored. */
SEXP _bssm_postcorrection_nongaussian(SEXP model_, SEXP model_type, SEXP output_type, SEXP nsim, SEXP seed, SEXP n_threads,
                                      SEXP is_type, SEXP counts, SEXP theta, SEXP modes) {
  const int mt = Rf_asInteger(model_type), ot = Rf_asInteger(output_type), it = Rf_asInteger(is_type);
  const int n_par = dim_at(theta, 0);
  int ns = dim_at(theta, 1), np = 0;
  const double *th = REAL(theta), *md = REAL(modes);
  uint32_t* cn = counts_u32(counts);
  const size_t mode_len = (size_t)dim_at(modes, 0) * dim_at(modes, 1);
  if (it == 3) { /* :197-199 */
    expanded e = expand_is3(th, n_par, md, mode_len, NULL, NULL, cn, ns);
    th = e.theta; md = e.modes; cn = e.counts; ns = e.n;
  }
  const bssm_ung_model* models = expand_models(model_, mt, th, n_par, ns);
  is_out o = run_is_ung(models, ns, md, NULL, cn, BSSM_PSI, it, ot, (uint32_t)Rf_asInteger(nsim), (uint32_t)Rf_asInteger(seed),
                        Rf_asInteger(n_threads), &np);
  SEXP post = PROTECT(Rf_allocVector(REALSXP, ns)); np++;
  for (int i = 0; i < ns; i++) REAL(post)[i] = log(REAL(o.weights)[i]);  /* prior and approx_loglik storage are zero here, :178-179 */
  SEXP res;
  if (ot == 1) {
    const char* nm[] = {"alpha", "weights", "posterior", ""};
    res = PROTECT(Rf_mkNamed(VECSXP, nm)); np++;
    SET_VECTOR_ELT(res, 0, o.alpha); SET_VECTOR_ELT(res, 1, o.weights); SET_VECTOR_ELT(res, 2, post);
  } else if (ot == 2) {
    const char* nm[] = {"alphahat", "Vt", "weights", "posterior", ""};
    res = PROTECT(Rf_mkNamed(VECSXP, nm)); np++;
    SET_VECTOR_ELT(res, 0, o.alphahat); SET_VECTOR_ELT(res, 1, o.Vt); SET_VECTOR_ELT(res, 2, o.weights); SET_VECTOR_ELT(res, 3, post);
  } else {
    const char* nm[] = {"weights", "posterior", ""};
    res = PROTECT(Rf_mkNamed(VECSXP, nm)); np++;
    SET_VECTOR_ELT(res, 0, o.weights); SET_VECTOR_ELT(res, 1, post);
  }
  UNPROTECT(np);
  return res;
}

/* ------------------------------------------------------------------ chains: closures as C callbacks */
typedef struct { SEXP model_, update_fn, prior_fn; } closure_ctx;
static int cb_update(const double* theta, int32_t n_par, bssm_ung_model* model, void* user) {
  closure_ctx* c = (closure_ctx*)user;
  apply_closure_update(c->update_fn, theta, n_par, c->model_, model);
  return 0;
}
static double cb_prior(const double* theta, int32_t n_par, void* user) { /* ssm_ung::log_prior_pdf, src/model_ssm_ung.cpp:80-82 */
  closure_ctx* c = (closure_ctx*)user;
  SEXP th = PROTECT(Rf_allocVector(REALSXP, n_par));
  memcpy(REAL(th), theta, (size_t)n_par * sizeof(double));
  SEXP call = PROTECT(Rf_lang2(c->prior_fn, th));
  const double v = Rf_asReal(Rf_eval(call, R_GlobalEnv));
  UNPROTECT(2);
  return v;
}

static SEXP transpose_theta(const double* theta, int n_par, int ns) { /* theta_storage.t(): n_stored x n_par */
  SEXP t = PROTECT(Rf_allocMatrix(REALSXP, ns, n_par));
  for (int i = 0; i < ns; i++) for (int j = 0; j < n_par; j++) REAL(t)[i + (size_t)ns * j] = theta[j + (size_t)n_par * i];
  UNPROTECT(1);
  return t;
}
static SEXP counts_sexp(const uint32_t* c, int ns) {
  SEXP v = PROTECT(Rf_allocVector(REALSXP, ns));
  for (int i = 0; i < ns; i++) REAL(v)[i] = c[i];
  UNPROTECT(1);
  return v;
}
static SEXP real_vec(const double* x, int ns) {
  SEXP v = PROTECT(Rf_allocVector(REALSXP, ns));
  if (ns) memcpy(REAL(v), x, (size_t)ns * sizeof(double));
  UNPROTECT(1);
  return v;
}
/* the list every *_is_mcmc stub returns, by output_type (src/R_mcmc.cpp:440-469 and the nonlinear / sde twins) */
static SEXP mcmc_result(int output_type, SEXP alpha, SEXP alphahat, SEXP Vt, SEXP theta, SEXP weights, SEXP counts, double acc, SEXP S,
                        SEXP posterior, SEXP modes) {
  const char* n1[] = {"alpha", "theta", "weights", "counts", "acceptance_rate", "S", "posterior", "modes", ""};
  const char* n2[] = {"alphahat", "Vt", "theta", "weights", "counts", "acceptance_rate", "S", "posterior", "modes", ""};
  const char* n3[] = {"theta", "weights", "counts", "acceptance_rate", "S", "posterior", "modes", ""};
  SEXP res = PROTECT(Rf_mkNamed(VECSXP, output_type == 1 ? n1 : (output_type == 2 ? n2 : n3)));
  int k = 0;
  if (output_type == 1) SET_VECTOR_ELT(res, k++, alpha);
  if (output_type == 2) { SET_VECTOR_ELT(res, k++, alphahat); SET_VECTOR_ELT(res, k++, Vt); }
  SET_VECTOR_ELT(res, k++, theta); SET_VECTOR_ELT(res, k++, weights); SET_VECTOR_ELT(res, k++, counts);
  SET_VECTOR_ELT(res, k++, Rf_ScalarReal(acc)); SET_VECTOR_ELT(res, k++, S); SET_VECTOR_ELT(res, k++, posterior);
  SET_VECTOR_ELT(res, k++, modes);
  UNPROTECT(1);
  return res;
}

typedef struct {
  int cap, ns;
  double acc;
  double *theta, *posterior, *approx_loglik, *prior, *modes;
  uint32_t* counts;
  SEXP S;
} chain_store;
static void chain_alloc(chain_store* cs, bssm_mcmc_out* out, int n_par, size_t mode_len, int iter, int burnin, int thin, SEXP S0, int* np) {
  memset(out, 0, sizeof(*out));
  cs->cap = (iter - burnin) / (thin > 0 ? thin : 1) + 1;
  cs->theta = (double*)R_alloc((size_t)n_par * cs->cap + 1, sizeof(double));
  cs->posterior = (double*)R_alloc((size_t)cs->cap + 1, sizeof(double));
  cs->approx_loglik = (double*)R_alloc((size_t)cs->cap + 1, sizeof(double));
  cs->prior = (double*)R_alloc((size_t)cs->cap + 1, sizeof(double));
  cs->modes = (double*)R_alloc(mode_len * cs->cap + 1, sizeof(double));
  cs->counts = (uint32_t*)R_alloc((size_t)cs->cap + 1, sizeof(uint32_t));
  cs->S = PROTECT(Rf_duplicate(S0)); (*np)++;
  out->capacity = cs->cap; out->S = REAL(cs->S); out->theta = cs->theta; out->counts = cs->counts; out->posterior = cs->posterior;
  out->approx_loglik = cs->approx_loglik; out->prior = cs->prior; out->modes = cs->modes;
}

/* .Call('_bssm_nongaussian_is_mcmc', model_, output_type, nsim, iter, burnin, thin, gamma, target_acceptance, S, seed,
 *       end_ram, n_threads, sampling_method, is_type, model_type, approx, verbose): src/R_mcmc.cpp:331-470.
 * approximate chain (approx_mcmc::amcmc) on the host, then is_correction_psi / _bsf on the GPUs.  The state outputs of
 * the purely approximate chain (approx = TRUE with output_type 1 / 2) are outside the path and stay with the reference. */
SEXP _bssm_nongaussian_is_mcmc(SEXP model_, SEXP output_type, SEXP nsim, SEXP iter, SEXP burnin, SEXP thin, SEXP gamma,
                               SEXP target_acceptance, SEXP S, SEXP seed, SEXP end_ram, SEXP n_threads, SEXP sampling_method,
                               SEXP is_type, SEXP model_type, SEXP approx, SEXP verbose) {
  const int mt = Rf_asInteger(model_type), ot = Rf_asInteger(output_type), it = Rf_asInteger(is_type), sm_ = Rf_asInteger(sampling_method);
  int np = 0;
  (void)verbose;
  if (sm_ != BSSM_PSI && sm_ != BSSM_BSF) Rf_error("sampling method 'spdk' is outside the particle-filter path of this build");
  if (Rf_asLogical(approx) && ot != 3) Rf_error("state outputs of the purely approximate chain are not part of this build");
  shim_model sm;
  unpack_model(model_, mt, 0, &sm);
  own_theta_dependent(model_, &sm);
  SEXP theta0 = elt(model_, "theta");
  const int n_par = Rf_length(theta0), n = sm.mod.n, p = sm.mod.p > 1 ? sm.mod.p : 1;
  int32_t flags[8];
  bssm_mcmc_job job;
  closure_ctx ctx = {model_, elt(model_, "update_fn"), elt(model_, "prior_fn")};
  memset(&job, 0, sizeof(job));
  job.model_type = mt; job.n_par = n_par;
  SEXP pd = R_NilValue;
  if (builtin_flags(model_, mt, flags)) {
    pd = PROTECT(Rf_coerceVector(elt(model_, "prior_distributions"), INTSXP)); np++;
    job.flags = flags; job.prior_distributions = (const int32_t*)INTEGER(pd); job.prior_parameters = REAL(elt(model_, "prior_parameters"));
  } else {
    job.update_fn = cb_update; job.prior_fn = cb_prior; job.user = &ctx;
  }
  job.mcmc_type = BSSM_MCMC_APPROX; job.sampling_method = (uint32_t)sm_; job.nsim = (uint32_t)Rf_asInteger(nsim);
  job.iter = (uint64_t)Rf_asInteger(iter); job.burnin = (uint64_t)Rf_asInteger(burnin); job.thin = (uint32_t)Rf_asInteger(thin);
  job.gamma = Rf_asReal(gamma); job.target_acceptance = Rf_asReal(target_acceptance); job.end_ram = Rf_asLogical(end_ram);
  job.seed = (uint32_t)Rf_asInteger(seed); job.output_type = ot;
  chain_store cs;
  bssm_mcmc_out out;
  chain_alloc(&cs, &out, n_par, (size_t)p * n, Rf_asInteger(iter), Rf_asInteger(burnin), Rf_asInteger(thin), S, &np);
  check_rc(bssm_run_mcmc(&sm.mod, REAL(theta0), &job, &out));
  int ns = out.n_stored;
  const double *th = cs.theta, *md = cs.modes, *al = cs.approx_loglik, *pr = cs.prior;
  uint32_t* cn = cs.counts;
  SEXP weights, alpha = R_NilValue, alphahat = R_NilValue, Vt = R_NilValue;
  double* loglik = NULL;
  if (Rf_asLogical(approx) || job.nsim <= 1) {
    weights = PROTECT(Rf_allocVector(REALSXP, ns)); np++;
    for (int i = 0; i < ns; i++) REAL(weights)[i] = 1.0;
  } else {
    if (it == 3) { /* src/R_mcmc.cpp:376-378 */
      expanded e = expand_is3(th, n_par, md, (size_t)p * n, al, pr, cn, ns);
      th = e.theta; md = e.modes; al = e.approx_loglik; pr = e.prior; cn = e.counts; ns = e.n;
    }
    const bssm_ung_model* models = expand_models(model_, mt, th, n_par, ns);
    is_out o = run_is_ung(models, ns, md, al, cn, sm_, it, ot, job.nsim, job.seed, Rf_asInteger(n_threads), &np);
    weights = o.weights; alpha = o.alpha; alphahat = o.alphahat; Vt = o.Vt; loglik = o.loglik;
  }
  (void)loglik;
  SEXP post = PROTECT(Rf_allocVector(REALSXP, ns)); np++;  /* prior + approx_loglik + log(w), src/approx_mcmc.cpp:425-426 */
  for (int i = 0; i < ns; i++) REAL(post)[i] = pr[i] + al[i] + log(REAL(weights)[i]);
  SEXP modes = PROTECT(Rf_alloc3DArray(REALSXP, p, n, ns)); np++;
  memcpy(REAL(modes), md, (size_t)p * n * ns * sizeof(double));
  SEXP tht = PROTECT(transpose_theta(th, n_par, ns)); np++;
  SEXP cnt = PROTECT(counts_sexp(cn, ns)); np++;
  SEXP res = mcmc_result(ot, alpha, alphahat, Vt, tht, weights, cnt, out.acceptance_rate, cs.S, post, modes);
  UNPROTECT(np);
  return res;
}


/* ------------------------------------------------------------------ pseudo-marginal / delayed-acceptance chains
 * run_mcmc(model, mcmc_type = "pm" | "da") (R/run_mcmc.R:497-522): one device filter pass per iteration through a
 * device-resident plan (bssm_run_mcmc, csrc/mcmc.cpp).  The lists carry no weights / modes (src/R_mcmc.cpp:207-232). */
static SEXP chain_result(int output_type, SEXP alpha, SEXP alphahat, SEXP Vt, SEXP theta, SEXP counts, double acc, SEXP S, SEXP posterior) {
  const char* n1[] = {"alpha", "theta", "counts", "acceptance_rate", "S", "posterior", ""};
  const char* n2[] = {"alphahat", "Vt", "theta", "counts", "acceptance_rate", "S", "posterior", ""};
  const char* n3[] = {"theta", "counts", "acceptance_rate", "S", "posterior", ""};
  SEXP res = PROTECT(Rf_mkNamed(VECSXP, output_type == 1 ? n1 : (output_type == 2 ? n2 : n3)));
  int k = 0;
  if (output_type == 1) SET_VECTOR_ELT(res, k++, alpha);
  if (output_type == 2) { SET_VECTOR_ELT(res, k++, alphahat); SET_VECTOR_ELT(res, k++, Vt); }
  SET_VECTOR_ELT(res, k++, theta); SET_VECTOR_ELT(res, k++, counts);
  SET_VECTOR_ELT(res, k++, Rf_ScalarReal(acc)); SET_VECTOR_ELT(res, k++, S); SET_VECTOR_ELT(res, k++, posterior);
  UNPROTECT(1);
  return res;
}
/* state storage of a filtered chain: alpha (n+1) x m x capacity (trimmed to n_stored afterwards), or the running summaries */
typedef struct { double* alpha_cap; SEXP ah_mn, Vt; } state_store;
static void state_alloc(state_store* st, bssm_mcmc_out* out, int ot, int n, int m, int cap, int* np) {
  st->alpha_cap = NULL; st->ah_mn = R_NilValue; st->Vt = R_NilValue;
  if (ot == 1) { st->alpha_cap = (double*)R_alloc((size_t)(n + 1) * m * cap + 1, sizeof(double)); out->alpha = st->alpha_cap; }
  if (ot == 2) {
    st->ah_mn = PROTECT(Rf_allocMatrix(REALSXP, m, n + 1)); (*np)++;
    st->Vt = PROTECT(Rf_alloc3DArray(REALSXP, m, m, n + 1)); (*np)++;
    out->alphahat = REAL(st->ah_mn); out->Vt = REAL(st->Vt);
  }
}
static SEXP finish_chain(const chain_store* cs, const bssm_mcmc_out* out, const state_store* st, int ot, int n, int m, int n_par, int* np) {
  const int ns = out->n_stored;
  SEXP alpha = R_NilValue, alphahat = R_NilValue;
  if (ot == 1) {
    alpha = PROTECT(Rf_alloc3DArray(REALSXP, n + 1, m, ns)); (*np)++;
    memcpy(REAL(alpha), st->alpha_cap, (size_t)(n + 1) * m * ns * sizeof(double));
  }
  if (ot == 2) { /* mcmc_run.alphahat.t() */
    alphahat = PROTECT(Rf_allocMatrix(REALSXP, n + 1, m)); (*np)++;
    for (int t = 0; t <= n; t++) for (int c = 0; c < m; c++) REAL(alphahat)[t + (size_t)(n + 1) * c] = REAL(st->ah_mn)[c + (size_t)m * t];
  }
  SEXP tht = PROTECT(transpose_theta(cs->theta, n_par, ns)); (*np)++;
  SEXP cnt = PROTECT(counts_sexp(cs->counts, ns)); (*np)++;
  SEXP post = PROTECT(real_vec(cs->posterior, ns)); (*np)++;
  return chain_result(ot, alpha, alphahat, st->Vt, tht, cnt, out->acceptance_rate, cs->S, post);
}

static SEXP nongaussian_chain(int mcmc_type, SEXP model_, SEXP output_type, SEXP nsim, SEXP iter, SEXP burnin, SEXP thin, SEXP gamma,
                              SEXP target_acceptance, SEXP S, SEXP seed, SEXP end_ram, SEXP sampling_method, SEXP model_type) {
  const int mt = Rf_asInteger(model_type), ot = Rf_asInteger(output_type), sm_ = Rf_asInteger(sampling_method);
  int np = 0;
  if (sm_ != BSSM_PSI && sm_ != BSSM_BSF) Rf_error("sampling method 'spdk' is outside the particle-filter path of this build");
  shim_model sm;
  unpack_model(model_, mt, 0, &sm);
  own_theta_dependent(model_, &sm);
  SEXP theta0 = elt(model_, "theta");
  const int n_par = Rf_length(theta0), n = sm.mod.n, m = sm.mod.m;
  int32_t flags[8];
  bssm_mcmc_job job;
  closure_ctx ctx = {model_, elt(model_, "update_fn"), elt(model_, "prior_fn")};
  memset(&job, 0, sizeof(job));
  job.model_type = mt; job.n_par = n_par;
  if (builtin_flags(model_, mt, flags)) {
    SEXP pd = PROTECT(Rf_coerceVector(elt(model_, "prior_distributions"), INTSXP)); np++;
    job.flags = flags; job.prior_distributions = (const int32_t*)INTEGER(pd); job.prior_parameters = REAL(elt(model_, "prior_parameters"));
  } else {
    job.update_fn = cb_update; job.prior_fn = cb_prior; job.user = &ctx;
  }
  job.mcmc_type = mcmc_type; job.sampling_method = (uint32_t)sm_; job.nsim = (uint32_t)Rf_asInteger(nsim);
  job.iter = (uint64_t)Rf_asInteger(iter); job.burnin = (uint64_t)Rf_asInteger(burnin); job.thin = (uint32_t)Rf_asInteger(thin);
  job.gamma = Rf_asReal(gamma); job.target_acceptance = Rf_asReal(target_acceptance); job.end_ram = Rf_asLogical(end_ram);
  job.seed = (uint32_t)Rf_asInteger(seed); job.output_type = ot;
  chain_store cs;
  bssm_mcmc_out out;
  state_store st;
  chain_alloc(&cs, &out, n_par, 1, Rf_asInteger(iter), Rf_asInteger(burnin), Rf_asInteger(thin), S, &np);
  out.modes = NULL; out.approx_loglik = NULL; out.prior = NULL;  /* filtered chains keep theta / counts / posterior only */
  state_alloc(&st, &out, ot, n, m, cs.cap, &np);
  check_rc(bssm_run_mcmc(&sm.mod, REAL(theta0), &job, &out));
  SEXP res = finish_chain(&cs, &out, &st, ot, n, m, n_par, &np);
  UNPROTECT(np);
  return res;
}
/* .Call('_bssm_nongaussian_pm_mcmc', model_, output_type, nsim, iter, burnin, thin, gamma, target_acceptance, S, seed, end_ram,
 *       n_threads, sampling_method, model_type, verbose): src/R_mcmc.cpp:167-242 (mcmc::pm_mcmc, src/mcmc.cpp:337-491) */
SEXP _bssm_nongaussian_pm_mcmc(SEXP model_, SEXP output_type, SEXP nsim, SEXP iter, SEXP burnin, SEXP thin, SEXP gamma, SEXP target_acceptance,
                               SEXP S, SEXP seed, SEXP end_ram, SEXP n_threads, SEXP sampling_method, SEXP model_type, SEXP verbose) {
  (void)n_threads; (void)verbose;
  return nongaussian_chain(BSSM_MCMC_PM, model_, output_type, nsim, iter, burnin, thin, gamma, target_acceptance, S, seed, end_ram, sampling_method, model_type);
}
/* .Call('_bssm_nongaussian_da_mcmc', <the same 15>): src/R_mcmc.cpp:246-327 (mcmc::da_mcmc, src/mcmc.cpp:530-685) */
SEXP _bssm_nongaussian_da_mcmc(SEXP model_, SEXP output_type, SEXP nsim, SEXP iter, SEXP burnin, SEXP thin, SEXP gamma, SEXP target_acceptance,
                               SEXP S, SEXP seed, SEXP end_ram, SEXP n_threads, SEXP sampling_method, SEXP model_type, SEXP verbose) {
  (void)n_threads; (void)verbose;
  return nongaussian_chain(BSSM_MCMC_DA, model_, output_type, nsim, iter, burnin, thin, gamma, target_acceptance, S, seed, end_ram, sampling_method, model_type);
}

/* ================================================================== ssm_nlg / ssm_sde */
static void read_numbers(SEXP attr, double* dst, int cap) {
  if (attr == R_NilValue) return;
  SEXP r = PROTECT(Rf_coerceVector(attr, REALSXP));
  for (int i = 0; i < cap && i < Rf_length(r); i++) dst[i] = REAL(r)[i];
  UNPROTECT(1);
}
/* identity of the model behind the external pointers: see the header comment of this file */
static void fn_identity(SEXP first_ptr, int n_states, int sde, bssm_fn_model* f) {
  SEXP id = Rf_getAttrib(first_ptr, Rf_install("bssm_amd_model_id"));
  SEXP code = Rf_getAttrib(first_ptr, Rf_install("bssm_amd_device_code"));
  if (id != R_NilValue) { f->model_id = Rf_asInteger(id); return; }
  if (code == R_NilValue)
    Rf_error("this build of bssm runs ssm_nlg / ssm_sde models on the GPU: the model needs attr(., 'bssm_amd_model_id') or "
             "attr(., 'bssm_amd_device_code') on its first function pointer (see INTEGRATION.md)");
  /* attr(., "bssm_amd_jit") = TRUE: compile the snippet in process (hiprtc; no hipcc, 0.2-0.3 s per kernel variant) -- for models that
   * are only run through the bootstrap filter; the psi-APF / EKF particle filter need the hipcc-built library (host-side set-up) */
  SEXP jit = Rf_getAttrib(first_ptr, Rf_install("bssm_amd_jit"));
  int slot;
  if (jit != R_NilValue && Rf_asLogical(jit) == 1) {
    slot = bssm_user_model_jit(CHAR(STRING_ELT(code, 0)), n_states);
  } else {
    char path[4096];
    check_rc(bssm_user_model_build(CHAR(STRING_ELT(code, 0)), n_states, path, (int32_t)sizeof(path)));  /* hipcc once, cached by content */
    slot = bssm_user_model_load(path);
  }
  if (slot < 1) Rf_error("%s", bssm_amd_last_error());
  f->model_id = sde ? BSSM_FN_USER_SDE : BSSM_FN_USER_NLG;
  f->user_slot = slot; f->n_states = n_states;
  if (!sde) {
    double h = 0.0;
    read_numbers(Rf_getAttrib(first_ptr, Rf_install("bssm_amd_H")), &h, 1); f->H = h;
    read_numbers(Rf_getAttrib(first_ptr, Rf_install("bssm_amd_R_diag")), f->R_diag, 2);
    read_numbers(Rf_getAttrib(first_ptr, Rf_install("bssm_amd_a1")), f->a1, 2);
    read_numbers(Rf_getAttrib(first_ptr, Rf_install("bssm_amd_P1")), f->P1, 4);
    /* wider models (up to 8 states, any n_states x n_etas matrix R; bootstrap paths): attr "bssm_amd_R" = R_fn's value */
    SEXP Rm = Rf_getAttrib(first_ptr, Rf_install("bssm_amd_R"));
    if (Rm != R_NilValue || n_states > 2) {
      f->use_ext = 1;
      f->n_etas = Rm != R_NilValue ? Rf_length(Rm) / n_states : n_states;
      read_numbers(Rm, f->R_ext, 256);
      read_numbers(Rf_getAttrib(first_ptr, Rf_install("bssm_amd_a1")), f->a1_ext, 16);
      read_numbers(Rf_getAttrib(first_ptr, Rf_install("bssm_amd_P1")), f->P1_ext, 256);
    }
  }
}
/* the 16 leading arguments every ssm_nlg stub shares (src/R_bootstrap_filter.cpp:461-466) */
#define NLG_ARGS SEXP y, SEXP Z, SEXP H, SEXP T, SEXP R, SEXP Zg, SEXP Tg, SEXP a1, SEXP P1, SEXP theta, SEXP log_prior_pdf, \
                 SEXP known_params, SEXP known_tv_params, SEXP n_states, SEXP n_etas, SEXP time_varying
#define NLG_UNUSED (void)H; (void)T; (void)R; (void)Zg; (void)Tg; (void)a1; (void)P1; (void)log_prior_pdf; (void)n_etas; (void)time_varying
static bssm_fn_model nlg_model_tv(SEXP y, SEXP Z, SEXP theta, SEXP known_params, SEXP known_tv_params, SEXP n_states) {
  bssm_fn_model f;
  memset(&f, 0, sizeof(f));
  /* known_tv_params (R/models.R:1507; default matrix(NA)): handed to BSSM_USER_TV snippets with the time index */
  if (TYPEOF(known_tv_params) == REALSXP && Rf_length(known_tv_params) > 0 && !isnan(REAL(known_tv_params)[0])) {
    f.known_tv_params = REAL(known_tv_params);
    f.tv_rows = dim_at(known_tv_params, 0) > 1 || dim_at(known_tv_params, 1) > 1 ? dim_at(known_tv_params, 0) : Rf_length(known_tv_params);
    f.tv_cols = Rf_length(known_tv_params) / (f.tv_rows > 0 ? f.tv_rows : 1);
  }
  f.n = Rf_length(y);                /* every built-in ssm_nlg model is univariate: y is 1 x n after t(y) on the R side */
  f.y = REAL(y);
  f.theta = REAL(theta); f.n_theta = Rf_length(theta);
  f.known_params = Rf_length(known_params) ? REAL(known_params) : NULL; f.n_known = Rf_length(known_params);
  fn_identity(Z, Rf_asInteger(n_states), 0, &f);
  return f;
}
static int fn_states(const bssm_fn_model* f) { return f->model_id == BSSM_FN_NLG_GROWTH ? 2 : (f->model_id == BSSM_FN_USER_NLG ? f->n_states : 1); }

typedef struct { int method, max_iter; double conv_tol; } fn_extra;
static int call_fn_filter(const void* mod, uint32_t N, uint32_t seed, bssm_filter_out* out, const void* ex) {
  const bssm_fn_model* f = (const bssm_fn_model*)mod;
  (void)ex;
  return (f->model_id == BSSM_FN_SDE_GBM || f->model_id == BSSM_FN_SDE_POISSON_OU || f->model_id == BSSM_FN_USER_SDE)
             ? bssm_bsf_sde(f, N, seed, out) : bssm_bsf_nlg(f, N, seed, out);
}
static int call_fn_smoother(const void* mod, uint32_t N, uint32_t seed, bssm_filter_out* out, const void* ex) {
  const fn_extra* e = (const fn_extra*)ex;
  return bssm_fn_smoother_stream((const bssm_fn_model*)mod, (uint32_t)e->method, N, seed, 0u, e->max_iter, e->conv_tol, out);
}
static int call_ekpf(const void* mod, uint32_t N, uint32_t seed, bssm_filter_out* out, const void* ex) {
  (void)ex;
  return bssm_ekpf_nlg((const bssm_fn_model*)mod, N, seed, out);
}

/* .Call('_bssm_bsf_nlg', <16 model args>, nsim, seed): src/R_bootstrap_filter.cpp:461-504 */
SEXP _bssm_bsf_nlg(NLG_ARGS, SEXP nsim, SEXP seed) {
  NLG_UNUSED;
  bssm_fn_model f = nlg_model_tv(y, Z, theta, known_params, known_tv_params, n_states);
  return run_filter_like(f.n, fn_states(&f), (uint32_t)Rf_asInteger(nsim), (uint32_t)Rf_asInteger(seed), 0, call_fn_filter, &f, NULL);
}
/* .Call('_bssm_bsf_smoother_nlg', ...): src/R_bootstrap_filter.cpp:508-551 */
SEXP _bssm_bsf_smoother_nlg(NLG_ARGS, SEXP nsim, SEXP seed) {
  NLG_UNUSED;
  bssm_fn_model f = nlg_model_tv(y, Z, theta, known_params, known_tv_params, n_states);
  fn_extra ex = {BSSM_BSF, 0, 0.0};
  return run_filter_like(f.n, fn_states(&f), (uint32_t)Rf_asInteger(nsim), (uint32_t)Rf_asInteger(seed), 1, call_fn_smoother, &f, &ex);
}
/* .Call('_bssm_psi_smoother_nlg', <16>, nsim, seed, max_iter, conv_tol, iekf_iter): src/R_psi.cpp:173-222 */
SEXP _bssm_psi_smoother_nlg(NLG_ARGS, SEXP nsim, SEXP seed, SEXP max_iter, SEXP conv_tol, SEXP iekf_iter) {
  NLG_UNUSED; (void)iekf_iter;
  bssm_fn_model f = nlg_model_tv(y, Z, theta, known_params, known_tv_params, n_states);
  fn_extra ex = {BSSM_PSI, Rf_asInteger(max_iter), Rf_asReal(conv_tol)};
  return run_filter_like(f.n, fn_states(&f), (uint32_t)Rf_asInteger(nsim), (uint32_t)Rf_asInteger(seed), 1, call_fn_smoother, &f, &ex);
}
/* .Call('_bssm_ekpf', <16>, nsim, seed) / '_bssm_ekpf_smoother': src/R_ekpf.cpp:11-100 */
SEXP _bssm_ekpf(NLG_ARGS, SEXP nsim, SEXP seed) {
  NLG_UNUSED;
  bssm_fn_model f = nlg_model_tv(y, Z, theta, known_params, known_tv_params, n_states);
  return run_filter_like(f.n, fn_states(&f), (uint32_t)Rf_asInteger(nsim), (uint32_t)Rf_asInteger(seed), 0, call_ekpf, &f, NULL);
}
SEXP _bssm_ekpf_smoother(NLG_ARGS, SEXP nsim, SEXP seed) {
  NLG_UNUSED;
  bssm_fn_model f = nlg_model_tv(y, Z, theta, known_params, known_tv_params, n_states);
  fn_extra ex = {BSSM_EKPF, 0, 0.0};
  return run_filter_like(f.n, fn_states(&f), (uint32_t)Rf_asInteger(nsim), (uint32_t)Rf_asInteger(seed), 1, call_fn_smoother, &f, &ex);
}
/* .Call('_bssm_nonlinear_loglik', <16>, nsim, seed, max_iter, conv_tol, iekf_iter, method): src/R_loglik.cpp:91-124 */
SEXP _bssm_nonlinear_loglik(NLG_ARGS, SEXP nsim, SEXP seed, SEXP max_iter, SEXP conv_tol, SEXP iekf_iter, SEXP method) {
  NLG_UNUSED; (void)iekf_iter;
  bssm_fn_model f = nlg_model_tv(y, Z, theta, known_params, known_tv_params, n_states);
  double ll = R_NegInf;
  const uint32_t N = (uint32_t)Rf_asInteger(nsim);
  if (N == 0) { /* the mode-based approximate log-likelihood, src/model_ssm_nlg.cpp:258-277 */
    check_rc(bssm_fn_psi_setup(&f, NULL, Rf_asInteger(max_iter), Rf_asReal(conv_tol), NULL, NULL, &ll, NULL, NULL, NULL, NULL, NULL, NULL, NULL));
  } else {
    check_rc(bssm_fn_loglik_batch_ex(&f, N, (uint32_t)Rf_asInteger(method), (uint32_t)Rf_asInteger(seed), 0u, 1u, Rf_asInteger(max_iter),
                                     Rf_asReal(conv_tol), &ll, NULL));
  }
  return Rf_ScalarReal(ll);
}
/* .Call('_bssm_suggest_n_nonlinear', <16>, theta_map, candidates, replications, seed): src/R_postcorrection.cpp:107-148 */
SEXP _bssm_suggest_n_nonlinear(NLG_ARGS, SEXP theta_map, SEXP candidates, SEXP replications, SEXP seed) {
  NLG_UNUSED; (void)theta;
  bssm_fn_model f = nlg_model_tv(y, Z, theta_map, known_params, known_tv_params, n_states);  /* model.update_model(theta_map), :137 */
  const int nc = Rf_length(candidates);
  uint32_t* cand = (uint32_t*)R_alloc((size_t)(nc ? nc : 1), sizeof(uint32_t));
  for (int i = 0; i < nc; i++) cand[i] = (uint32_t)REAL(candidates)[i];
  SEXP sds = PROTECT(Rf_allocVector(REALSXP, nc));
  check_rc(bssm_suggest_n_nonlinear_multi(&f, cand, nc, (uint32_t)Rf_asInteger(replications), (uint32_t)Rf_asInteger(seed), NULL, 0, REAL(sds), NULL));
  UNPROTECT(1);
  return sds;
}

/* IS correction of functor-model draws over the GPUs of the node; theta n_par x n_stored */
static is_out run_is_fn(const bssm_fn_model* tmpl, const double* theta, int n_par, int ns, const double* modes, const double* approx_loglik,
                        const uint32_t* counts, int sampling_method, int is_type, int output_type, uint32_t nsim, uint32_t seed,
                        int n_threads, int* n_protect) {
  is_out o = {R_NilValue, R_NilValue, R_NilValue, R_NilValue, NULL};
  const int n = tmpl->n, m = fn_states(tmpl);
  bssm_fn_model* models = (bssm_fn_model*)R_alloc((size_t)ns + 1, sizeof(bssm_fn_model));
  for (int i = 0; i < ns; i++) { models[i] = *tmpl; models[i].theta = theta + (size_t)n_par * i; models[i].n_theta = n_par; }
  o.weights = PROTECT(Rf_allocVector(REALSXP, ns)); (*n_protect)++;
  o.loglik = (double*)R_alloc((size_t)ns + 1, sizeof(double));
  double *alpha = NULL, *ah_i = NULL, *vt_i = NULL;
  if (output_type == 1) { o.alpha = PROTECT(Rf_alloc3DArray(REALSXP, n + 1, m, ns)); (*n_protect)++; alpha = REAL(o.alpha); }
  if (output_type == 2) {
    ah_i = (double*)R_alloc((size_t)m * (n + 1) * ns + 1, sizeof(double));
    vt_i = (double*)R_alloc((size_t)m * m * (n + 1) * ns + 1, sizeof(double));
  }
  check_rc(bssm_is_correction_nlg_multi(models, 0, ns, modes, approx_loglik, counts, sampling_method, is_type, output_type, nsim, seed,
                                        n_threads, NULL, 0, REAL(o.weights), o.loglik, alpha, ah_i, vt_i));
  if (output_type == 2) {
    SEXP ah = PROTECT(Rf_allocMatrix(REALSXP, m, n + 1)); (*n_protect)++;
    o.Vt = PROTECT(Rf_alloc3DArray(REALSXP, m, m, n + 1)); (*n_protect)++;
    check_rc(bssm_is_merge(n, m, ns, REAL(o.weights), counts, ah_i, vt_i, REAL(ah), REAL(o.Vt)));
    o.alphahat = PROTECT(Rf_allocMatrix(REALSXP, n + 1, m)); (*n_protect)++;
    for (int t = 0; t <= n; t++) for (int c = 0; c < m; c++) REAL(o.alphahat)[t + (size_t)(n + 1) * c] = REAL(ah)[c + (size_t)m * t];
  }
  return o;
}

/* .Call('_bssm_postcorrection_nonlinear', <16 with theta_init>, output_type, nsim, seed, n_threads, is_type, counts, theta,
 *       modes): src/R_postcorrection.cpp:252-317 */
SEXP _bssm_postcorrection_nonlinear(NLG_ARGS, SEXP output_type, SEXP nsim, SEXP seed, SEXP n_threads, SEXP is_type, SEXP counts,
                                    SEXP theta_mat, SEXP modes) {
  NLG_UNUSED;
  bssm_fn_model f = nlg_model_tv(y, Z, theta, known_params, known_tv_params, n_states);
  const int ot = Rf_asInteger(output_type), it = Rf_asInteger(is_type), n_par = dim_at(theta_mat, 0);
  int ns = dim_at(theta_mat, 1), np = 0;
  const double *th = REAL(theta_mat), *md = REAL(modes);
  uint32_t* cn = counts_u32(counts);
  const size_t mode_len = (size_t)dim_at(modes, 0) * dim_at(modes, 1);
  if (it == 3) { expanded e = expand_is3(th, n_par, md, mode_len, NULL, NULL, cn, ns); th = e.theta; md = e.modes; cn = e.counts; ns = e.n; }
  is_out o = run_is_fn(&f, th, n_par, ns, md, NULL, cn, BSSM_PSI, it, ot, (uint32_t)Rf_asInteger(nsim), (uint32_t)Rf_asInteger(seed),
                       Rf_asInteger(n_threads), &np);
  SEXP post = PROTECT(Rf_allocVector(REALSXP, ns)); np++;
  for (int i = 0; i < ns; i++) REAL(post)[i] = log(REAL(o.weights)[i]);
  SEXP res;
  if (ot == 1) {
    const char* nm[] = {"alpha", "weights", "posterior", ""};
    res = PROTECT(Rf_mkNamed(VECSXP, nm)); np++;
    SET_VECTOR_ELT(res, 0, o.alpha); SET_VECTOR_ELT(res, 1, o.weights); SET_VECTOR_ELT(res, 2, post);
  } else if (ot == 2) {
    const char* nm[] = {"alphahat", "Vt", "weights", "posterior", ""};
    res = PROTECT(Rf_mkNamed(VECSXP, nm)); np++;
    SET_VECTOR_ELT(res, 0, o.alphahat); SET_VECTOR_ELT(res, 1, o.Vt); SET_VECTOR_ELT(res, 2, o.weights); SET_VECTOR_ELT(res, 3, post);
  } else {
    const char* nm[] = {"weights", "posterior", ""};
    res = PROTECT(Rf_mkNamed(VECSXP, nm)); np++;
    SET_VECTOR_ELT(res, 0, o.weights); SET_VECTOR_ELT(res, 1, post);
  }
  UNPROTECT(np);
  return res;
}

/* log_prior_pdf of a functor model: the reference calls the user's compiled pointer (src/model_ssm_nlg.h:21).  The shim
 * evaluates the R function the constructor keeps next to it, attr(log_prior_pdf, "bssm_amd_prior_fn") (function(theta)) */
static double cb_fn_prior(const double* theta, int32_t n_par, void* user) {
  SEXP fn = (SEXP)user;
  SEXP th = PROTECT(Rf_allocVector(REALSXP, n_par));
  memcpy(REAL(th), theta, (size_t)n_par * sizeof(double));
  SEXP call = PROTECT(Rf_lang2(fn, th));
  const double v = Rf_asReal(Rf_eval(call, R_GlobalEnv));
  UNPROTECT(2);
  return v;
}
static SEXP prior_closure(SEXP log_prior_pdf) {
  SEXP fn = Rf_getAttrib(log_prior_pdf, Rf_install("bssm_amd_prior_fn"));
  if (!Rf_isFunction(fn)) Rf_error("the chains need attr(log_prior_pdf, 'bssm_amd_prior_fn'): an R function(theta) returning the log prior");
  return fn;
}

/* .Call('_bssm_nonlinear_is_mcmc', y, Z, H, T, R, Zg, Tg, a1, P1, theta, log_prior_pdf, known_params, known_tv_params,
 *       time_varying, n_states, n_etas, seed, nsim, iter, burnin, thin, gamma, target_acceptance, S, end_ram, n_threads,
 *       is_type, sampling_method, max_iter, conv_tol, iekf_iter, output_type, approx, verbose): src/R_mcmc.cpp:736-832 */
SEXP _bssm_nonlinear_is_mcmc(SEXP y, SEXP Z, SEXP H, SEXP T, SEXP R, SEXP Zg, SEXP Tg, SEXP a1, SEXP P1, SEXP theta, SEXP log_prior_pdf,
                             SEXP known_params, SEXP known_tv_params, SEXP time_varying, SEXP n_states, SEXP n_etas, SEXP seed,
                             SEXP nsim, SEXP iter, SEXP burnin, SEXP thin, SEXP gamma, SEXP target_acceptance, SEXP S, SEXP end_ram,
                             SEXP n_threads, SEXP is_type, SEXP sampling_method, SEXP max_iter, SEXP conv_tol, SEXP iekf_iter,
                             SEXP output_type, SEXP approx, SEXP verbose) {
  (void)H; (void)T; (void)R; (void)Zg; (void)Tg; (void)a1; (void)P1; (void)known_tv_params; (void)time_varying; (void)n_etas;
  (void)iekf_iter; (void)verbose;
  bssm_fn_model f = nlg_model_tv(y, Z, theta, known_params, known_tv_params, n_states);
  const int ot = Rf_asInteger(output_type), it = Rf_asInteger(is_type), sm_ = Rf_asInteger(sampling_method), n_par = Rf_length(theta);
  const int n = f.n, m = fn_states(&f);
  int np = 0;
  if (Rf_asLogical(approx) && ot != 3) Rf_error("state outputs of the purely approximate chain are not part of this build");
  bssm_mcmc_job job;
  memset(&job, 0, sizeof(job));
  job.n_par = n_par; job.prior_fn = cb_fn_prior; job.user = (void*)prior_closure(log_prior_pdf);
  job.mcmc_type = BSSM_MCMC_APPROX; job.sampling_method = (uint32_t)sm_; job.nsim = (uint32_t)Rf_asInteger(nsim);
  job.iter = (uint64_t)Rf_asInteger(iter); job.burnin = (uint64_t)Rf_asInteger(burnin); job.thin = (uint32_t)Rf_asInteger(thin);
  job.gamma = Rf_asReal(gamma); job.target_acceptance = Rf_asReal(target_acceptance); job.end_ram = Rf_asLogical(end_ram);
  job.seed = (uint32_t)Rf_asInteger(seed); job.output_type = ot; job.max_iter = Rf_asInteger(max_iter); job.conv_tol = Rf_asReal(conv_tol);
  chain_store cs;
  bssm_mcmc_out out;
  chain_alloc(&cs, &out, n_par, (size_t)m * n, Rf_asInteger(iter), Rf_asInteger(burnin), Rf_asInteger(thin), S, &np);
  check_rc(bssm_run_mcmc_fn(&f, REAL(theta), &job, &out));
  int ns = out.n_stored;
  const double *th = cs.theta, *md = cs.modes, *al = cs.approx_loglik, *pr = cs.prior;
  uint32_t* cn = cs.counts;
  SEXP weights, alpha = R_NilValue, alphahat = R_NilValue, Vt = R_NilValue;
  if (Rf_asLogical(approx) || job.nsim <= 1) {
    weights = PROTECT(Rf_allocVector(REALSXP, ns)); np++;
    for (int i = 0; i < ns; i++) REAL(weights)[i] = 1.0;
  } else {
    if (it == 3) { expanded e = expand_is3(th, n_par, md, (size_t)m * n, al, pr, cn, ns); th = e.theta; md = e.modes; al = e.approx_loglik; pr = e.prior; cn = e.counts; ns = e.n; }
    is_out o = run_is_fn(&f, th, n_par, ns, md, al, cn, sm_ == BSSM_PSI ? BSSM_PSI : BSSM_BSF, it, ot, job.nsim, job.seed, Rf_asInteger(n_threads), &np);
    weights = o.weights; alpha = o.alpha; alphahat = o.alphahat; Vt = o.Vt;
  }
  SEXP post = PROTECT(Rf_allocVector(REALSXP, ns)); np++;
  for (int i = 0; i < ns; i++) REAL(post)[i] = pr[i] + al[i] + log(REAL(weights)[i]);
  SEXP modes = PROTECT(Rf_alloc3DArray(REALSXP, m, n, ns)); np++;
  memcpy(REAL(modes), md, (size_t)m * n * ns * sizeof(double));
  SEXP tht = PROTECT(transpose_theta(th, n_par, ns)); np++;
  SEXP cnt = PROTECT(counts_sexp(cn, ns)); np++;
  SEXP res = mcmc_result(ot, alpha, alphahat, Vt, tht, weights, cnt, out.acceptance_rate, cs.S, post, modes);
  UNPROTECT(np);
  return res;
}


/* pm / da chains of the functor models: bssm_run_mcmc_fn with the filter of each iteration on the device */
static SEXP fn_chain(const bssm_fn_model* f, int mcmc_type, int sampling_method, int L_c, int L_f, SEXP theta, SEXP log_prior_pdf, SEXP nsim,
                     SEXP iter, SEXP burnin, SEXP thin, SEXP gamma, SEXP target_acceptance, SEXP S, SEXP end_ram, SEXP seed, int max_iter,
                     double conv_tol, int ot) {
  const int n_par = Rf_length(theta), n = f->n, m = fn_states(f);
  int np = 0;
  bssm_mcmc_job job;
  memset(&job, 0, sizeof(job));
  job.n_par = n_par; job.prior_fn = cb_fn_prior; job.user = (void*)prior_closure(log_prior_pdf);
  job.mcmc_type = mcmc_type; job.sampling_method = (uint32_t)sampling_method; job.nsim = (uint32_t)Rf_asInteger(nsim);
  job.iter = (uint64_t)Rf_asInteger(iter); job.burnin = (uint64_t)Rf_asInteger(burnin); job.thin = (uint32_t)Rf_asInteger(thin);
  job.gamma = Rf_asReal(gamma); job.target_acceptance = Rf_asReal(target_acceptance); job.end_ram = Rf_asLogical(end_ram);
  job.seed = (uint32_t)Rf_asInteger(seed); job.output_type = ot; job.max_iter = max_iter; job.conv_tol = conv_tol;
  job.L_c = L_c; job.L_f = L_f;
  chain_store cs;
  bssm_mcmc_out out;
  state_store st;
  chain_alloc(&cs, &out, n_par, 1, Rf_asInteger(iter), Rf_asInteger(burnin), Rf_asInteger(thin), S, &np);
  out.modes = NULL; out.approx_loglik = NULL; out.prior = NULL;
  state_alloc(&st, &out, ot, n, m, cs.cap, &np);
  check_rc(bssm_run_mcmc_fn(f, REAL(theta), &job, &out));
  SEXP res = finish_chain(&cs, &out, &st, ot, n, m, n_par, &np);
  UNPROTECT(np);
  return res;
}
/* .Call('_bssm_nonlinear_pm_mcmc', y, Z, H, T, R, Zg, Tg, a1, P1, theta, log_prior_pdf, known_params, known_tv_params, time_varying,
 *       n_states, n_etas, seed, nsim, iter, burnin, thin, gamma, target_acceptance, S, end_ram, n_threads, max_iter, conv_tol,
 *       sampling_method, iekf_iter, output_type, verbose): src/R_mcmc.cpp:550-608; sampling_method 1 psi / 2 bsf / 4 ekf */
SEXP _bssm_nonlinear_pm_mcmc(SEXP y, SEXP Z, SEXP H, SEXP T, SEXP R, SEXP Zg, SEXP Tg, SEXP a1, SEXP P1, SEXP theta, SEXP log_prior_pdf,
                             SEXP known_params, SEXP known_tv_params, SEXP time_varying, SEXP n_states, SEXP n_etas, SEXP seed, SEXP nsim,
                             SEXP iter, SEXP burnin, SEXP thin, SEXP gamma, SEXP target_acceptance, SEXP S, SEXP end_ram, SEXP n_threads,
                             SEXP max_iter, SEXP conv_tol, SEXP sampling_method, SEXP iekf_iter, SEXP output_type, SEXP verbose) {
  (void)H; (void)T; (void)R; (void)Zg; (void)Tg; (void)a1; (void)P1; (void)time_varying; (void)n_etas; (void)n_threads; (void)iekf_iter; (void)verbose;
  bssm_fn_model f = nlg_model_tv(y, Z, theta, known_params, known_tv_params, n_states);
  return fn_chain(&f, BSSM_MCMC_PM, Rf_asInteger(sampling_method), 0, 0, theta, log_prior_pdf, nsim, iter, burnin, thin, gamma, target_acceptance, S,
                  end_ram, seed, Rf_asInteger(max_iter), Rf_asReal(conv_tol), Rf_asInteger(output_type));
}
/* .Call('_bssm_nonlinear_da_mcmc', <the same 32>): src/R_mcmc.cpp:610-668 */
SEXP _bssm_nonlinear_da_mcmc(SEXP y, SEXP Z, SEXP H, SEXP T, SEXP R, SEXP Zg, SEXP Tg, SEXP a1, SEXP P1, SEXP theta, SEXP log_prior_pdf,
                             SEXP known_params, SEXP known_tv_params, SEXP time_varying, SEXP n_states, SEXP n_etas, SEXP seed, SEXP nsim,
                             SEXP iter, SEXP burnin, SEXP thin, SEXP gamma, SEXP target_acceptance, SEXP S, SEXP end_ram, SEXP n_threads,
                             SEXP max_iter, SEXP conv_tol, SEXP sampling_method, SEXP iekf_iter, SEXP output_type, SEXP verbose) {
  (void)H; (void)T; (void)R; (void)Zg; (void)Tg; (void)a1; (void)P1; (void)time_varying; (void)n_etas; (void)n_threads; (void)iekf_iter; (void)verbose;
  bssm_fn_model f = nlg_model_tv(y, Z, theta, known_params, known_tv_params, n_states);
  return fn_chain(&f, BSSM_MCMC_DA, Rf_asInteger(sampling_method), 0, 0, theta, log_prior_pdf, nsim, iter, burnin, thin, gamma, target_acceptance, S,
                  end_ram, seed, Rf_asInteger(max_iter), Rf_asReal(conv_tol), Rf_asInteger(output_type));
}

/* ------------------------------------------------------------------ ssm_sde */
#define SDE_ARGS SEXP y, SEXP x0, SEXP positive, SEXP drift, SEXP diffusion, SEXP ddiffusion, SEXP log_prior_pdf, SEXP log_obs_density
#define SDE_UNUSED (void)diffusion; (void)ddiffusion; (void)log_prior_pdf; (void)log_obs_density
static bssm_fn_model sde_model(SEXP y, SEXP x0, SEXP positive, SEXP drift, SEXP theta, int L) {
  bssm_fn_model f;
  memset(&f, 0, sizeof(f));
  f.n = Rf_length(y); f.y = REAL(y);
  f.theta = REAL(theta); f.n_theta = Rf_length(theta);
  f.x0 = Rf_asReal(x0); f.positive = Rf_asLogical(positive); f.L = L;
  fn_identity(drift, 1, 1, &f);
  return f;
}
/* .Call('_bssm_loglik_sde', y, x0, positive, drift, diffusion, ddiffusion, log_prior_pdf, log_obs_density, theta, nsim, L, seed): src/R_sde.cpp:10-31 */
SEXP _bssm_loglik_sde(SDE_ARGS, SEXP theta, SEXP nsim, SEXP L, SEXP seed) {
  SDE_UNUSED;
  bssm_fn_model f = sde_model(y, x0, positive, drift, theta, Rf_asInteger(L));
  double ll = R_NegInf;
  check_rc(bssm_fn_loglik_batch(&f, (uint32_t)Rf_asInteger(nsim), BSSM_BSF, (uint32_t)Rf_asInteger(seed), 0u, 1u, &ll));
  return Rf_ScalarReal(ll);
}
/* .Call('_bssm_bsf_sde', ...): src/R_sde.cpp:35-71 */
SEXP _bssm_bsf_sde(SDE_ARGS, SEXP theta, SEXP nsim, SEXP L, SEXP seed) {
  SDE_UNUSED;
  bssm_fn_model f = sde_model(y, x0, positive, drift, theta, Rf_asInteger(L));
  return run_filter_like(f.n, 1, (uint32_t)Rf_asInteger(nsim), (uint32_t)Rf_asInteger(seed), 0, call_fn_filter, &f, NULL);
}
/* .Call('_bssm_R_milstein', x0, L, t, theta, drift_pntr, diffusion_pntr, ddiffusion_pntr, positive, seed): src/R_milstein.cpp:4-22 */
SEXP _bssm_R_milstein(SEXP x0, SEXP L, SEXP t, SEXP theta, SEXP drift, SEXP diffusion, SEXP ddiffusion, SEXP positive, SEXP seed) {
  (void)diffusion; (void)ddiffusion;
  bssm_fn_model f;
  double x = 0.0;
  memset(&f, 0, sizeof(f));
  f.theta = REAL(theta); f.n_theta = Rf_length(theta);
  f.x0 = Rf_asReal(x0); f.positive = Rf_asLogical(positive); f.L = Rf_asInteger(L);
  fn_identity(drift, 1, 1, &f);
  check_rc(bssm_milstein(&f, Rf_asReal(t), (uint32_t)Rf_asInteger(seed), &x));
  return Rf_ScalarReal(x);
}
/* .Call('_bssm_bsf_smoother_sde', ...): src/R_sde.cpp:75-130 */
SEXP _bssm_bsf_smoother_sde(SDE_ARGS, SEXP theta, SEXP nsim, SEXP L, SEXP seed) {
  SDE_UNUSED;
  bssm_fn_model f = sde_model(y, x0, positive, drift, theta, Rf_asInteger(L));
  fn_extra ex = {BSSM_BSF, 0, 0.0};
  return run_filter_like(f.n, 1, (uint32_t)Rf_asInteger(nsim), (uint32_t)Rf_asInteger(seed), 1, call_fn_smoother, &f, &ex);
}
/* .Call('_bssm_sde_state_sampler_bsf_is2', y, x0, positive, drift, diffusion, ddiffusion, log_prior_pdf, log_obs_density, nsim,
 *       L_f, seed, approx_loglik_storage, theta): src/R_sde_state_sampler.cpp:10-50; theta n_par x n_stored */
SEXP _bssm_sde_state_sampler_bsf_is2(SDE_ARGS, SEXP nsim, SEXP L_f, SEXP seed, SEXP approx_loglik_storage, SEXP theta) {
  SDE_UNUSED;
  const int n_par = dim_at(theta, 0), ns = dim_at(theta, 1);
  int np = 0;
  SEXP th0 = PROTECT(Rf_allocVector(REALSXP, n_par)); np++;
  memcpy(REAL(th0), REAL(theta), (size_t)n_par * sizeof(double));
  bssm_fn_model f = sde_model(y, x0, positive, drift, th0, Rf_asInteger(L_f));
  uint32_t* ones = (uint32_t*)R_alloc((size_t)ns + 1, sizeof(uint32_t));
  for (int i = 0; i < ns; i++) ones[i] = 1u;
  is_out o = run_is_fn(&f, REAL(theta), n_par, ns, NULL, REAL(approx_loglik_storage), ones, BSSM_BSF, 2, 1, (uint32_t)Rf_asInteger(nsim),
                       (uint32_t)Rf_asInteger(seed), 1, &np);
  const char* nm[] = {"alpha", "weights", ""};
  SEXP res = PROTECT(Rf_mkNamed(VECSXP, nm)); np++;
  SET_VECTOR_ELT(res, 0, o.alpha); SET_VECTOR_ELT(res, 1, o.weights);
  UNPROTECT(np);
  return res;
}
/* .Call('_bssm_sde_is_mcmc', y, x0, positive, drift, diffusion, ddiffusion, log_prior_pdf, log_obs_density, theta, nsim, L_c, L_f,
 *       seed, iter, burnin, thin, gamma, target_acceptance, S, end_ram, is_type, n_threads, type, verbose): src/R_sde.cpp:221-290.
 * Chain at the coarse discretisation L_c (approx_mcmc::amcmc(ssm_sde), src/approx_mcmc.cpp:236-330: a pseudo-marginal
 * chain whose log-likelihood estimates are kept as approx_loglik), bootstrap correction at L_f (:523-600). */
SEXP _bssm_sde_is_mcmc(SDE_ARGS, SEXP theta, SEXP nsim, SEXP L_c, SEXP L_f, SEXP seed, SEXP iter, SEXP burnin, SEXP thin, SEXP gamma,
                       SEXP target_acceptance, SEXP S, SEXP end_ram, SEXP is_type, SEXP n_threads, SEXP type, SEXP verbose) {
  (void)diffusion; (void)ddiffusion; (void)log_obs_density; (void)verbose;
  bssm_fn_model f = sde_model(y, x0, positive, drift, theta, Rf_asInteger(L_c));
  const int ot = Rf_asInteger(type), it = Rf_asInteger(is_type), n_par = Rf_length(theta), n = f.n;
  int np = 0;
  bssm_mcmc_job job;
  memset(&job, 0, sizeof(job));
  job.n_par = n_par; job.prior_fn = cb_fn_prior; job.user = (void*)prior_closure(log_prior_pdf);
  job.mcmc_type = BSSM_MCMC_PM; job.sampling_method = BSSM_BSF; job.nsim = (uint32_t)Rf_asInteger(nsim);
  job.iter = (uint64_t)Rf_asInteger(iter); job.burnin = (uint64_t)Rf_asInteger(burnin); job.thin = (uint32_t)Rf_asInteger(thin);
  job.gamma = Rf_asReal(gamma); job.target_acceptance = Rf_asReal(target_acceptance); job.end_ram = Rf_asLogical(end_ram);
  job.seed = (uint32_t)Rf_asInteger(seed); job.output_type = 3;
  chain_store cs;
  bssm_mcmc_out out;
  chain_alloc(&cs, &out, n_par, 1, Rf_asInteger(iter), Rf_asInteger(burnin), Rf_asInteger(thin), S, &np);
  out.modes = NULL;
  check_rc(bssm_run_mcmc_fn(&f, REAL(theta), &job, &out));
  int ns = out.n_stored;
  const double *th = cs.theta, *al = cs.approx_loglik, *pr = cs.prior;
  uint32_t* cn = cs.counts;
  double dummy_mode = 0.0;
  if (it == 3) { expanded e = expand_is3(th, n_par, &dummy_mode, 0, al, pr, cn, ns); th = e.theta; al = e.approx_loglik; pr = e.prior; cn = e.counts; ns = e.n; }
  f.L = Rf_asInteger(L_f);
  is_out o = run_is_fn(&f, th, n_par, ns, NULL, al, cn, BSSM_BSF, it, ot, job.nsim, job.seed, Rf_asInteger(n_threads), &np);
  SEXP post = PROTECT(Rf_allocVector(REALSXP, ns)); np++;
  for (int i = 0; i < ns; i++) REAL(post)[i] = pr[i] + al[i] + log(REAL(o.weights)[i]);
  SEXP tht = PROTECT(transpose_theta(th, n_par, ns)); np++;
  SEXP cnt = PROTECT(counts_sexp(cn, ns)); np++;
  (void)n;
  /* the sde list has no `modes` entry (src/R_sde.cpp:262-285) */
  const char* n1[] = {"alpha", "theta", "weights", "counts", "acceptance_rate", "S", "posterior", ""};
  const char* n2[] = {"alphahat", "Vt", "theta", "weights", "counts", "acceptance_rate", "S", "posterior", ""};
  const char* n3[] = {"theta", "weights", "counts", "acceptance_rate", "S", "posterior", ""};
  SEXP res = PROTECT(Rf_mkNamed(VECSXP, ot == 1 ? n1 : (ot == 2 ? n2 : n3))); np++;
  int k = 0;
  if (ot == 1) SET_VECTOR_ELT(res, k++, o.alpha);
  if (ot == 2) { SET_VECTOR_ELT(res, k++, o.alphahat); SET_VECTOR_ELT(res, k++, o.Vt); }
  SET_VECTOR_ELT(res, k++, tht); SET_VECTOR_ELT(res, k++, o.weights); SET_VECTOR_ELT(res, k++, cnt);
  SET_VECTOR_ELT(res, k++, Rf_ScalarReal(out.acceptance_rate)); SET_VECTOR_ELT(res, k++, cs.S); SET_VECTOR_ELT(res, k++, post);
  UNPROTECT(np);
  return res;
}


/* .Call('_bssm_sde_pm_mcmc', y, x0, positive, drift, diffusion, ddiffusion, log_prior_pdf, log_obs_density, theta, nsim, L, seed, iter,
 *       burnin, thin, gamma, target_acceptance, S, end_ram, type, verbose): src/R_sde.cpp:134-178 */
SEXP _bssm_sde_pm_mcmc(SDE_ARGS, SEXP theta, SEXP nsim, SEXP L, SEXP seed, SEXP iter, SEXP burnin, SEXP thin, SEXP gamma, SEXP target_acceptance,
                       SEXP S, SEXP end_ram, SEXP type, SEXP verbose) {
  (void)diffusion; (void)ddiffusion; (void)log_obs_density; (void)verbose;
  bssm_fn_model f = sde_model(y, x0, positive, drift, theta, Rf_asInteger(L));
  return fn_chain(&f, BSSM_MCMC_PM, BSSM_BSF, 0, 0, theta, log_prior_pdf, nsim, iter, burnin, thin, gamma, target_acceptance, S, end_ram, seed, 0, 0.0,
                  Rf_asInteger(type));
}
/* .Call('_bssm_sde_da_mcmc', <8>, theta, nsim, L_c, L_f, seed, iter, burnin, thin, gamma, target_acceptance, S, end_ram, type, verbose):
 * src/R_sde.cpp:180-219; coarse bootstrap filter first, the fine one for proposals that pass (src/mcmc.cpp:836-960) */
SEXP _bssm_sde_da_mcmc(SDE_ARGS, SEXP theta, SEXP nsim, SEXP L_c, SEXP L_f, SEXP seed, SEXP iter, SEXP burnin, SEXP thin, SEXP gamma,
                       SEXP target_acceptance, SEXP S, SEXP end_ram, SEXP type, SEXP verbose) {
  (void)diffusion; (void)ddiffusion; (void)log_obs_density; (void)verbose;
  bssm_fn_model f = sde_model(y, x0, positive, drift, theta, Rf_asInteger(L_f));
  return fn_chain(&f, BSSM_MCMC_DA, BSSM_BSF, Rf_asInteger(L_c), Rf_asInteger(L_f), theta, log_prior_pdf, nsim, iter, burnin, thin, gamma,
                  target_acceptance, S, end_ram, seed, 0, 0.0, Rf_asInteger(type));
}

/* ================================================================== registration
 * These rows replace the same-named rows of CallEntries in src/RcppExports.cpp:1287-1346; every other row of that table
 * (Kalman filters, prediction, ...) keeps pointing at the reference's own stubs. */
#define ENTRY(name, n) {#name, (DL_FUNC)&name, n}
static const R_CallMethodDef bssm_amd_entries[] = {
    ENTRY(_bssm_bsf, 5),
    ENTRY(_bssm_bsf_smoother, 5),
    ENTRY(_bssm_bsf_nlg, 18),
    ENTRY(_bssm_bsf_smoother_nlg, 18),
    ENTRY(_bssm_ekpf, 18),
    ENTRY(_bssm_ekpf_smoother, 18),
    ENTRY(_bssm_nongaussian_loglik, 5),
    ENTRY(_bssm_nonlinear_loglik, 22),
    ENTRY(_bssm_nongaussian_is_mcmc, 17),
    ENTRY(_bssm_nongaussian_pm_mcmc, 15),
    ENTRY(_bssm_nongaussian_da_mcmc, 15),
    ENTRY(_bssm_nonlinear_is_mcmc, 34),
    ENTRY(_bssm_nonlinear_pm_mcmc, 32),
    ENTRY(_bssm_nonlinear_da_mcmc, 32),
    ENTRY(_bssm_suggest_n_nongaussian, 6),
    ENTRY(_bssm_suggest_n_nonlinear, 20),
    ENTRY(_bssm_postcorrection_nongaussian, 10),
    ENTRY(_bssm_postcorrection_nonlinear, 24),
    ENTRY(_bssm_gaussian_psi_smoother, 4),
    ENTRY(_bssm_psi_smoother, 4),
    ENTRY(_bssm_psi_smoother_nlg, 21),
    ENTRY(_bssm_loglik_sde, 12),
    ENTRY(_bssm_bsf_sde, 12),
    ENTRY(_bssm_bsf_smoother_sde, 12),
    ENTRY(_bssm_R_milstein, 9),
    ENTRY(_bssm_sde_is_mcmc, 24),
    ENTRY(_bssm_sde_pm_mcmc, 21),
    ENTRY(_bssm_sde_da_mcmc, 22),
    ENTRY(_bssm_sde_state_sampler_bsf_is2, 13),
    ENTRY(_bssm_stratified_sample, 3),
    {NULL, NULL, 0}};

/* called from R_init_bssm next to (or instead of) the reference's own registration, src/RcppExports.cpp:1348-1351 */
void bssm_amd_register(DllInfo* dll) {
  R_registerRoutines(dll, NULL, bssm_amd_entries, NULL, NULL);
  R_useDynamicSymbols(dll, 0);
}
