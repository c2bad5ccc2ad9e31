// mcmc.cpp -- the chains that call the particle-filter path: pseudo-marginal and delayed-acceptance MCMC with one
// device filter pass per iteration, and the approximate chain whose stored values the IS post-correction re-weights.
//
// Follows mcmc::pm_mcmc (src/mcmc.cpp:337-491), mcmc::da_mcmc (src/mcmc.cpp:530-685) and approx_mcmc::amcmc
// (src/approx_mcmc.cpp:110-232).  The proposal adaptation is ramcmc::adapt_S of the CRAN package `ramcmc` (linked by
// DESCRIPTION:49, not vendored under the reference): Vihola (2012), robust adaptive Metropolis,
//   S S' <- S (I + eta_i (alpha_i - alpha*) u u' / |u|^2) S',   eta_i = min(1, d i^-gamma),
// applied as a rank-one Cholesky update / downdate of the lower factor.
//
// The chain's random numbers are counter based like the filters': block (iter_lo, iter_hi, d, RNG_MCMC) of the key
// (seed, 0); the filter of iteration i uses the stream i of the same seed, so a run is a pure function of
// (model, theta0, S, seed).
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "host_model.h"
#include "pf_device.h"

namespace {

enum { RNG_MCMC = 5 };

struct ChainRng {
  uint32_t seed;
  // standard normals of iteration `it` (libm Box-Muller on one Philox block per pair)
  void normals(uint64_t it, int cnt, double* out) const {
    for (int j = 0; j < cnt; j += 2) {
      const bssm::u32x4 w = bssm::philox4x32_10((uint32_t)it, (uint32_t)(it >> 32), (uint32_t)(j / 2), RNG_MCMC, seed, 0u);
      const double u1 = (double)(bssm::bits53(w.x, w.y) + 1) * 0x1.0p-53;  // (0,1]
      const double rad = std::sqrt(-2.0 * std::log(u1));
      const double ang = 2.0 * 3.14159265358979323846 * ((double)w.z * 0x1.0p-32);
      out[j] = rad * std::cos(ang);
      if (j + 1 < cnt) out[j + 1] = rad * std::sin(ang);
    }
  }
  // acceptance uniform `which` (0: first / only stage, 1: second stage of delayed acceptance)
  double uniform(uint64_t it, int which) const {
    const bssm::u32x4 w = bssm::philox4x32_10((uint32_t)it, (uint32_t)(it >> 32), 0x80000000u + (uint32_t)which, RNG_MCMC, seed, 0u);
    return (double)bssm::bits53(w.x, w.y) * 0x1.0p-53;
  }
};

// log_prior_pdf of the built-in models: src/model_bsm_ng.cpp:90-136, src/model_svm.cpp:50-88, src/model_ar1_ng.cpp:63-118
double log_prior_builtin(int model_type, const int32_t* fl, int n_xreg, const double* x, int n_par, const int32_t* pd,
                         const double* pp) {
  std::vector<double> pars(x, x + n_par);
  double lp = 0.0;
  if (model_type == BSSM_BSM_NG) {
    const int noise = fl[2], phi_est = fl[6];
    if (fl[3] + fl[4] + fl[5] < 3 || noise || phi_est)
      for (int i = 0; i < n_par - n_xreg; i++) { pars[i] = std::exp(x[i]); lp += x[i]; }  // log scale + Jacobian
  } else if (model_type == BSSM_AR1_NG) {
    pars[1] = std::exp(x[1]); lp += x[1];
    if (fl[1]) { pars[2 + fl[0]] = std::exp(x[2 + fl[0]]); lp += x[2 + fl[0]]; }
  }
  for (int i = 0; i < n_par; i++) {
    const double* q = pp + 4 * (size_t)i;
    const double v = pars[i];
    switch (pd[i]) {
      case 0: if (v < q[0] || v > q[1]) return -INFINITY; break;
      case 1: if (v < 0) return -INFINITY; lp -= 0.5 * (v / q[0]) * (v / q[0]); break;
      case 2: lp -= 0.5 * ((v - q[0]) / q[1]) * ((v - q[0]) / q[1]); break;
      case 3: if (v < q[2] || v > q[3]) return -INFINITY; lp -= 0.5 * ((v - q[0]) / q[1]) * ((v - q[0]) / q[1]); break;
      case 4: if (v < 0) return -INFINITY; lp += (q[0] - 1) * std::log(v) - q[1] * v; break;
      default: break;
    }
  }
  return lp;
}

// L L' <- L L' + sign * v v'  (lower triangular L, d x d column-major; v is consumed)
void chol_rank_one(double* L, double* v, int d, bool update) {
  const double sg = update ? 1.0 : -1.0;
  for (int i = 0; i < d; i++) {
    const double lii = L[i + (size_t)d * i];
    const double r = std::sqrt(lii * lii + sg * v[i] * v[i]);
    if (i == d - 1) { L[i + (size_t)d * i] = r; break; }
    const double c = r / lii, s = v[i] / lii;
    L[i + (size_t)d * i] = r;
    for (int j = i + 1; j < d; j++) {
      double& l = L[j + (size_t)d * i];
      l = (l + sg * s * v[j]) / c;
      v[j] = c * v[j] - s * l;
    }
  }
}

void adapt_S(double* S, const double* u, int d, double acc, double target, uint64_t i, double gamma) {
  const double change = acc - target;
  double nu = 0;
  for (int j = 0; j < d; j++) nu += u[j] * u[j];
  nu = std::sqrt(nu);
  if (!(nu > 0)) return;
  const double scale = std::sqrt(std::fmin(1.0, d * std::pow((double)i, -gamma)) * std::fabs(change)) / nu;
  std::vector<double> v(d, 0.0);
  for (int r = 0; r < d; r++) {
    double acc_r = 0;
    for (int c = 0; c <= r; c++) acc_r += S[r + (size_t)d * c] * u[c];
    v[r] = acc_r * scale;
  }
  chol_rank_one(S, v.data(), d, change > 0.0);
}

// the chain's working copy of the model: owns the theta-dependent arrays of a built-in model
struct WorkModel {
  bssm_ung_model mod;
  std::vector<double> T, R, P1, a1, C, beta;
  const bssm_mcmc_job* job = nullptr;
  bssm_ung_model* caller = nullptr;

  void init(bssm_ung_model* src, const bssm_mcmc_job* j) {
    job = j; caller = src; mod = *src;
    if (j->update_fn) return;  // callbacks rewrite the caller's arrays; `mod` is refreshed from *caller after each call
    const int n = mod.n, m = mod.m, k = mod.k;
    T.assign(src->T, src->T + (size_t)m * m * (mod.Ttv ? n : 1));
    R.assign(src->R, src->R + (size_t)m * k * (mod.Rtv ? n : 1));
    P1.assign(src->P1, src->P1 + (size_t)m * m);
    a1.assign(src->a1, src->a1 + m);
    C.assign(src->C, src->C + (size_t)m * (mod.Ctv ? n : 1));
    if (mod.n_xreg > 0) beta.assign(src->beta, src->beta + mod.n_xreg);
    mod.T = T.data(); mod.R = R.data(); mod.P1 = P1.data(); mod.a1 = a1.data(); mod.C = C.data();
    mod.beta = beta.empty() ? nullptr : beta.data();
  }
  std::string update(const double* theta) {
    if (job->update_fn) {
      if (job->update_fn(theta, job->n_par, caller, job->user) != 0) return "update_fn failed";
      mod = *caller;
      return "";
    }
    return bssm::update_model_builtin(job->model_type, job->flags, theta, job->n_par, &mod, T.data(), R.data(), P1.data(),
                                      a1.data(), C.data(), beta.empty() ? nullptr : beta.data());
  }
  double log_prior(const double* theta) const {
    if (job->prior_fn) return job->prior_fn(theta, job->n_par, job->user);
    return log_prior_builtin(job->model_type, job->flags, mod.n_xreg, theta, job->n_par, job->prior_distributions,
                             job->prior_parameters);
  }
};

struct PlanGuard {
  bssm_plan* p = nullptr;
  ~PlanGuard() { if (p) bssm_plan_destroy(p); }
};

// ---- what a chain needs from a model family
// ssm_ung / ssm_mng family: host Gaussian approximation, device filter on a resident plan
struct UngTarget {
  WorkModel wm;
  const bssm_mcmc_job* job;
  PlanGuard plan;
  int otype = 3;
  bssm::GlobalApprox global;  // local_approx = FALSE: the pseudo-observations of the chain's first approximation
  int n() const { return wm.mod.n; }
  int m() const { return wm.mod.m; }
  int mode_rows() const { return wm.mod.p > 1 ? wm.mod.p : 1; }
  std::string update(const double* theta) { return wm.update(theta); }
  double log_prior(const double* theta) const { return wm.log_prior(theta); }
  int approx_ll(uint64_t /*it*/, std::vector<double>* mode, double* val) {
    bssm::HostModel hm;
    const std::string e = bssm::ingest_model(&wm.mod, true, &hm);
    if (!e.empty()) return bssm::set_error(BSSM_ERR_INVALID, e);
    *val = bssm::approx_loglik_mode(hm, mode, wm.mod.local_approx ? nullptr : &global);
    return BSSM_OK;
  }
  // one device filter pass of the working model; ll[0] unbiased estimate, ll[1] approximate (psi) or the same (bsf),
  // src/model_ssm_ung.cpp:176-213
  int filter_ll(uint64_t it, double* ll) {
    int rc;
    const bool psi = job->sampling_method == BSSM_PSI;
    if (!plan.p) {
      plan.p = bssm_plan_create(&wm.mod, 1, 1, job->nsim, job->sampling_method, otype != 3 ? BSSM_PLAN_KEEP_HISTORY : 0u);
      if (!plan.p) return BSSM_ERR_INVALID;  // message already recorded
      // local_approx = FALSE: the psi filter of the initial value made the chain's first approximation
      // (src/model_ssm_ung.cpp:514-518); the approximate stage of delayed acceptance reuses its pseudo-observations
      if (psi && !wm.mod.local_approx && !global.set) { double tmp; if ((rc = approx_ll(it, nullptr, &tmp)) != BSSM_OK) return rc; }
    } else if ((rc = bssm_plan_update(plan.p, &wm.mod, nullptr)) != BSSM_OK) {
      if (rc == BSSM_ERR_NUMERIC) { ll[0] = ll[1] = -INFINITY; return BSSM_OK; }  // a failed approximation rejects the proposal
      return rc;
    }
    if ((rc = bssm_plan_run(plan.p, job->seed + (uint32_t)(it >> 32), (uint32_t)it, nullptr)) != BSSM_OK) return rc;
    if ((rc = bssm_plan_loglik(plan.p, nullptr, &ll[0])) != BSSM_OK) return rc;
    ll[1] = ll[0];
    if (psi && (rc = bssm_plan_approx_loglik(plan.p, &ll[1])) != BSSM_OK) return rc;
    if (!std::isfinite(ll[0])) ll[0] = -INFINITY;
    return BSSM_OK;
  }
  // src/summary.cpp:82-103 on the particle system just filtered
  int sample_or_summarise(uint64_t it, double* sampled, double* ah, double* vt) {
    const uint32_t sd = job->seed + (uint32_t)(it >> 32);
    if (otype == 1) return bssm_plan_sample_or_summarise(plan.p, 0, sd, nullptr, sampled, nullptr, nullptr);
    if (otype == 2) return bssm_plan_sample_or_summarise(plan.p, 0, sd, nullptr, nullptr, ah, vt);
    return BSSM_OK;
  }
};

// ssm_nlg / ssm_sde (model functors): theta is the model's parameter vector itself; approximate log-likelihood from the
// mode search (ssm_nlg::log_likelihood with nsim = 0, src/model_ssm_nlg.cpp:258-277); filters through the functor
// entry points.  Random streams: Philox key (chain_seed(), 2 * iteration) for the fine / only filter of an iteration and
// (chain_seed(), 2 * iteration + 1) for the coarse first stage of the two-level ssm_sde chain; chain_seed() is the job's
// seed in a domain of its own, so chains started with consecutive seeds share nothing and the IS correction that
// follows (key (seed, draw index)) never re-uses a chain stream.
struct FnTarget {
  bssm_fn_model mod;
  std::vector<double> theta;
  const bssm_mcmc_job* job;
  int otype = 3;
  std::vector<double> alpha_all, w_dummy;  // smoother outputs of the accepted proposal
  int n() const { return mod.n; }
  int m() const { return mod.model_id == BSSM_FN_NLG_GROWTH ? 2 : (mod.model_id == BSSM_FN_USER_NLG ? mod.n_states : 1); }
  int mode_rows() const { return m(); }
  bool is_sde() const { return mod.model_id == BSSM_FN_SDE_GBM || mod.model_id == BSSM_FN_SDE_POISSON_OU || mod.model_id == BSSM_FN_USER_SDE; }
  std::string update(const double* th) {
    std::memcpy(theta.data(), th, theta.size() * sizeof(double));
    if (job->fn_update_fn && job->fn_update_fn(th, job->n_par, &mod, job->user) != 0) return "fn_update_fn failed";
    mod.theta = theta.data();
    return "";
  }
  double log_prior(const double* th) const { return job->prior_fn(th, job->n_par, job->user); }
  int L_c = 0, L_f = 0;  // ssm_sde delayed acceptance: coarse / fine discretisation levels (0 = single level, mod.L)
  bool two_level() const { return L_c > 0 && L_f > L_c; }
  uint32_t chain_seed() const { return job->seed ^ 0x5bd1e995u; }
  double psi_approx_ll = 0.0;  // approximate log-likelihood of the set-up the last psi filter drew from
  int approx_ll(uint64_t it, std::vector<double>* mode, double* val) {
    if (two_level()) {  // first stage of sde_da_mcmc: a bootstrap filter at the coarse level (src/mcmc.cpp:859,908)
      mod.L = L_c;
      const int rc = bssm_fn_loglik_batch(&mod, job->nsim, BSSM_BSF, chain_seed(), 2u * (uint32_t)it + 1u, 1u, val);
      mod.L = L_f;
      if (rc < 0) return rc;
      if (!std::isfinite(*val)) *val = -INFINITY;
      return BSSM_OK;
    }
    std::vector<double> md((size_t)m() * n());
    const int rc = bssm_fn_psi_setup(&mod, nullptr, job->max_iter, job->conv_tol, md.data(), nullptr, val, nullptr, nullptr,
                                     nullptr, nullptr, nullptr, nullptr, nullptr);
    if (rc == BSSM_ERR_NUMERIC || rc == BSSM_WARN_DEGENERATE) { *val = -INFINITY; return BSSM_OK; }
    if (rc != BSSM_OK) return rc;
    if (mode) *mode = md;
    return BSSM_OK;
  }
  int filter_ll(uint64_t it, double* ll) {
    int rc = bssm_fn_loglik_batch_ex(&mod, job->nsim, job->sampling_method, chain_seed(), 2u * (uint32_t)it, 1u, job->max_iter,
                                     job->conv_tol, &ll[0], &psi_approx_ll);
    if (rc < 0) return rc;
    if (!std::isfinite(ll[0])) ll[0] = -INFINITY;
    if (two_level()) {  // ll[1] is the coarse-level estimate: the caller's first stage made it, except for the initial value
      if (it == 0 && (rc = approx_ll(0, nullptr, &ll[1])) != BSSM_OK) return rc;
      return BSSM_OK;
    }
    ll[1] = ll[0];
    // ll(1) = approx_loglik of the same Gaussian approximation (one mode search serves both), src/model_ssm_nlg.cpp:253-254
    if (job->sampling_method == BSSM_PSI && std::isfinite(ll[0])) ll[1] = psi_approx_ll;
    return BSSM_OK;
  }
  int sample_or_summarise(uint64_t it, double* sampled, double* ah, double* vt) {
    if (otype == 3) return BSSM_OK;
    const int nn = n(), mm = m();
    const uint32_t N = job->nsim, sd = chain_seed(), st = 2u * (uint32_t)it;
    alpha_all.resize((size_t)mm * (nn + 1) * N);
    w_dummy.resize((size_t)N * (nn + 1));
    std::vector<double> ahat((size_t)(nn + 1) * mm), v((size_t)mm * mm * (nn + 1));
    bssm_filter_out o;
    std::memset(&o, 0, sizeof o);
    o.alphahat = ahat.data(); o.Vt = v.data(); o.weights = w_dummy.data(); o.alpha = alpha_all.data();
    // the accepted particle system again: same set-up (max_iter, conv_tol), same (seed, stream)
    const int rc = bssm_fn_smoother_stream(&mod, job->sampling_method, N, sd, st, job->max_iter, job->conv_tol, &o);
    if (rc < 0) return rc;
    if (otype == 1) {  // one of the back-traced trajectories, uniformly (the weights of the extra time point are ones)
      const bssm::u32x4 w = bssm::philox4x32_10(0, 0, 0, bssm::RNG_PICK, sd, st);
      uint32_t pick = (uint32_t)((double)bssm::bits53(w.x, w.y) * 0x1.0p-53 * N);
      if (pick >= N) pick = N - 1;
      for (int t = 0; t <= nn; t++)
        for (int c = 0; c < mm; c++) sampled[t + (size_t)(nn + 1) * c] = alpha_all[c + (size_t)mm * t + (size_t)mm * (nn + 1) * pick];
    } else {
      for (int t = 0; t <= nn; t++)
        for (int c = 0; c < mm; c++) ah[c + (size_t)mm * t] = ahat[t + (size_t)(nn + 1) * c];
      std::memcpy(vt, v.data(), v.size() * sizeof(double));
    }
    return BSSM_OK;
  }
};

// ---- the chain: approx_mcmc::amcmc / mcmc::pm_mcmc / mcmc::da_mcmc over a target
template <class Target>
int run_chain(Target& tg, const double* theta0, const bssm_mcmc_job* job, bssm_mcmc_out* out) {
  using bssm::set_error;
  const int d = job->n_par, type = job->mcmc_type;
  const bool filtered = type != BSSM_MCMC_APPROX;
  const int otype = tg.otype;
  std::vector<double> theta(theta0, theta0 + d), theta_prop(d), u(d);
  std::string err = tg.update(theta.data());  // "just in case", src/mcmc.cpp:348
  if (!err.empty()) return set_error(BSSM_ERR_INVALID, err);
  double logprior = tg.log_prior(theta.data());
  if (!std::isfinite(logprior)) return set_error(BSSM_ERR_INVALID, "Initial prior probability is not finite.");

  const int n = tg.n(), m = tg.m(), prow = tg.mode_rows();
  const ChainRng rng{job->seed};
  const size_t sz_a = (size_t)(n + 1) * m, sz_v = (size_t)m * m * (n + 1);
  std::vector<double> sampled(otype == 1 ? sz_a : 0), ah_i(otype == 2 ? sz_a : 0), vt_i(otype == 2 ? sz_v : 0),
      valpha(otype == 2 ? sz_v : 0, 0.0), mode, mode_prop, diff(m), after(m);  // sized by m: states go to 64

  double ll[2] = {0, 0}, ll_prop[2] = {0, 0};
  int rc;
  if (filtered) {
    if ((rc = tg.filter_ll(0, ll)) != BSSM_OK) return rc;
    if (!std::isfinite(ll[0])) return set_error(BSSM_ERR_INVALID, "Initial log-likelihood is not finite.");
    if ((rc = tg.sample_or_summarise(0, sampled.data(), ah_i.data(), vt_i.data())) != BSSM_OK) return rc;
    if (otype == 2) {
      std::fill(out->alphahat, out->alphahat + sz_a, 0.0);
      std::fill(out->Vt, out->Vt + sz_v, 0.0);
    }
  } else {
    if ((rc = tg.approx_ll(0, &mode, &ll[0])) != BSSM_OK) return rc;
    ll[1] = ll[0];
    if (!std::isfinite(ll[0])) return set_error(BSSM_ERR_INVALID, "Initial log-likelihood is not finite.");
  }

  bool new_value = true;
  uint64_t n_values = 0, accepted = 0;
  int n_stored = 0;
  double* S = out->S;
  for (uint64_t i = 1; i <= job->iter; i++) {
    rng.normals(i, d, u.data());
    for (int r = 0; r < d; r++) {
      double acc = theta[r];
      for (int c = 0; c <= r; c++) acc += S[r + (size_t)d * c] * u[c];
      theta_prop[r] = acc;
    }
    double acceptance_prob = 0.0;
    const double logprior_prop = tg.log_prior(theta_prop.data());
    if (logprior_prop > -INFINITY && !std::isnan(logprior_prop)) {
      err = tg.update(theta_prop.data());
      if (!err.empty()) return set_error(BSSM_ERR_INVALID, err);
      bool accept = false;
      if (type == BSSM_MCMC_APPROX) {
        if ((rc = tg.approx_ll(i, &mode_prop, &ll_prop[0])) != BSSM_OK) return rc;
        ll_prop[1] = ll_prop[0];
        acceptance_prob = std::fmin(1.0, std::exp(ll_prop[0] - ll[0] + logprior_prop - logprior));
        accept = rng.uniform(i, 0) < acceptance_prob;
      } else if (type == BSSM_MCMC_PM) {
        if ((rc = tg.filter_ll(i, ll_prop)) != BSSM_OK) return rc;
        // the adaptation sees the approximate likelihoods, the chain the unbiased ones (src/mcmc.cpp:414-424)
        acceptance_prob = std::fmin(1.0, std::exp(ll_prop[1] - ll[1] + logprior_prop - logprior));
        const double log_alpha = ll_prop[0] - ll[0] + logprior_prop - logprior;
        accept = std::log(rng.uniform(i, 0)) < log_alpha;
      } else {  // delayed acceptance: the filter runs only for proposals that pass the approximate stage
        if ((rc = tg.approx_ll(i, nullptr, &ll_prop[1])) != BSSM_OK) return rc;
        acceptance_prob = std::fmin(1.0, std::exp(ll_prop[1] - ll[1] + logprior_prop - logprior));
        if (rng.uniform(i, 0) < acceptance_prob) {
          if ((rc = tg.filter_ll(i, ll_prop)) != BSSM_OK) return rc;
          const double log_alpha = ll_prop[0] + ll[1] - ll[0] - ll_prop[1];
          accept = std::log(rng.uniform(i, 1)) < log_alpha;
        }
      }
      if (std::isnan(acceptance_prob)) acceptance_prob = 0.0;
      if (accept) {
        if (i > job->burnin) { accepted++; n_values++; }
        if (filtered && (rc = tg.sample_or_summarise(i, sampled.data(), ah_i.data(), vt_i.data())) != BSSM_OK) return rc;
        ll[0] = ll_prop[0]; ll[1] = ll_prop[1];
        logprior = logprior_prop;
        theta = theta_prop;
        if (!filtered) mode = mode_prop;
        new_value = true;
      }
    }

    if (i > job->burnin && otype == 2) {  // running mean of the summaries, thinning does not affect this
      const double cnt = (double)(i - job->burnin);
      for (int t = 0; t <= n; t++) {
        for (int r = 0; r < m; r++) {
          double& a = out->alphahat[r + (size_t)m * t];
          diff[r] = ah_i[r + (size_t)m * t] - a;
          a = (a * (cnt - 1) + ah_i[r + (size_t)m * t]) / cnt;
          after[r] = ah_i[r + (size_t)m * t] - a;
        }
        for (int c = 0; c < m; c++)
          for (int r = 0; r < m; r++) {
            const size_t q = r + (size_t)m * c + (size_t)m * m * t;
            out->Vt[q] = (out->Vt[q] * (cnt - 1) + vt_i[q]) / cnt;
            valpha[q] += diff[r] * after[c];
          }
      }
    }

    if (i > job->burnin && n_values % job->thin == 0) {
      if (new_value) {
        if (n_stored >= out->capacity) return set_error(BSSM_ERR_INVALID, "stored-value capacity exhausted");
        std::memcpy(out->theta + (size_t)d * n_stored, theta.data(), d * sizeof(double));
        out->counts[n_stored] = 1;
        out->posterior[n_stored] = logprior + ll[0];
        // (filtered chains store them too: the coarse-level ssm_sde chain is the "approximate" chain of its IS types,
        // src/approx_mcmc.cpp:236-330, and its log-likelihood estimates are what the correction divides by)
        if (out->approx_loglik) out->approx_loglik[n_stored] = ll[0];
        if (out->prior) out->prior[n_stored] = logprior;
        if (!filtered && out->modes) std::memcpy(out->modes + (size_t)prow * n * n_stored, mode.data(), (size_t)prow * n * sizeof(double));
        if (otype == 1) std::memcpy(out->alpha + sz_a * n_stored, sampled.data(), sz_a * sizeof(double));
        n_stored++;
        new_value = false;
      } else {
        out->counts[n_stored - 1]++;
      }
    }
    if (!job->end_ram || i <= job->burnin) adapt_S(S, u.data(), d, acceptance_prob, job->target_acceptance, i, job->gamma);
  }
  if (n_stored == 0) return set_error(BSSM_ERR_INVALID, "No proposals were accepted in MCMC. Check your model.");
  if (otype == 2) {  // Var[E(alpha)] + E[Var(alpha)]
    const double cnt = (double)(job->iter - job->burnin);
    for (size_t q = 0; q < sz_v; q++) out->Vt[q] += valpha[q] / cnt;
  }
  out->n_stored = n_stored;
  out->acceptance_rate = (double)accepted / (double)(job->iter - job->burnin);
  return BSSM_OK;
}

// argument checks shared by the two entry points; returns the effective output type through *otype
int check_job(const bssm_mcmc_job* job, const bssm_mcmc_out* out, bool fn, int* otype) {
  using bssm::set_error;
  if (job->n_par < 1) return set_error(BSSM_ERR_INVALID, "No unknown parameters ('model$theta' has length of zero).");
  if (job->burnin > job->iter) return set_error(BSSM_ERR_INVALID, "Argument 'burnin' should be smaller than 'iter'.");
  if (job->thin < 1) return set_error(BSSM_ERR_INVALID, "thin must be positive");
  if (!out->S || !out->theta || !out->counts || !out->posterior || out->capacity < 1)
    return set_error(BSSM_ERR_INVALID, "output storage missing");
  const int type = job->mcmc_type;
  if (type != BSSM_MCMC_APPROX && type != BSSM_MCMC_PM && type != BSSM_MCMC_DA) return set_error(BSSM_ERR_INVALID, "unknown mcmc_type");
  const bool filtered = type != BSSM_MCMC_APPROX;
  const bool method_ok = job->sampling_method == BSSM_PSI || job->sampling_method == BSSM_BSF || (fn && job->sampling_method == BSSM_EKPF);
  if (filtered && !method_ok)
    return set_error(BSSM_ERR_UNSUPPORTED, "sampling_method must be 1 (psi) or 2 (bsf) (4, ekf, for ssm_nlg); SPDK is not a particle filter path");
  if (filtered && job->nsim < 2)
    return set_error(BSSM_ERR_INVALID, "Number of state samples less than 2, use 'mcmc_type' 'approx' instead.");
  *otype = filtered ? job->output_type : 3;
  if (*otype == 1 && !out->alpha) return set_error(BSSM_ERR_INVALID, "output_type 1 needs alpha storage");
  if (*otype == 2 && (!out->alphahat || !out->Vt)) return set_error(BSSM_ERR_INVALID, "output_type 2 needs alphahat and Vt");
  return BSSM_OK;
}

}  // namespace

extern "C" {

void bssm_mcmc_draws(uint32_t seed, uint64_t iter, int32_t n_par, double* normals, double* uniforms2) {
  const ChainRng rng{seed};
  if (normals) rng.normals(iter, n_par, normals);
  if (uniforms2) { uniforms2[0] = rng.uniform(iter, 0); uniforms2[1] = rng.uniform(iter, 1); }
}

double bssm_log_prior_builtin(int model_type, const int32_t* flags, int32_t n_xreg, const double* theta, int32_t n_par,
                              const int32_t* prior_distributions, const double* prior_parameters) {
  return log_prior_builtin(model_type, flags, n_xreg, theta, n_par, prior_distributions, prior_parameters);
}

int bssm_run_mcmc(bssm_ung_model* model, const double* theta0, const bssm_mcmc_job* job, bssm_mcmc_out* out) {
  using bssm::set_error;
  if (!model || !theta0 || !job || !out) return set_error(BSSM_ERR_INVALID, "NULL argument");
  const bool builtin = job->model_type == BSSM_BSM_NG || job->model_type == BSSM_SVM_MODEL || job->model_type == BSSM_AR1_NG;
  if (builtin && !job->update_fn && (!job->flags || !job->prior_distributions || !job->prior_parameters))
    return set_error(BSSM_ERR_INVALID, "built-in models need flags and the prior tables");
  if (!builtin && (!job->update_fn || !job->prior_fn))
    return set_error(BSSM_ERR_INVALID, "ssm_ung / ssm_mng chains need update_fn and prior_fn");
  UngTarget tg;
  int rc = check_job(job, out, false, &tg.otype);
  if (rc != BSSM_OK) return rc;
  tg.job = job;
  tg.wm.init(model, job);
  return run_chain(tg, theta0, job, out);
}

int bssm_run_mcmc_fn(const bssm_fn_model* model, const double* theta0, const bssm_mcmc_job* job, bssm_mcmc_out* out) {
  using bssm::set_error;
  if (!model || !theta0 || !job || !out) return set_error(BSSM_ERR_INVALID, "NULL argument");
  if (!job->prior_fn) return set_error(BSSM_ERR_INVALID, "ssm_nlg / ssm_sde chains need prior_fn (the model's log_prior_pdf)");
  if (job->n_par != model->n_theta) return set_error(BSSM_ERR_INVALID, "n_par must equal the model's n_theta");
  FnTarget tg;
  int rc = check_job(job, out, true, &tg.otype);
  if (rc != BSSM_OK) return rc;
  const bool sde = model->model_id == BSSM_FN_SDE_GBM || model->model_id == BSSM_FN_SDE_POISSON_OU || model->model_id == BSSM_FN_USER_SDE;
  const bool two_level = job->L_c > 0 && job->L_f > job->L_c;
  if (sde && (job->sampling_method != BSSM_BSF || !(job->mcmc_type == BSSM_MCMC_PM || (job->mcmc_type == BSSM_MCMC_DA && two_level))))
    return set_error(BSSM_ERR_UNSUPPORTED, "ssm_sde: bootstrap filter with mcmc_type pm, or da with discretisation levels 1 <= L_c < L_f");
  tg.job = job;
  tg.mod = *model;
  tg.theta.assign(theta0, theta0 + job->n_par);
  tg.mod.theta = tg.theta.data();
  if (sde && job->mcmc_type == BSSM_MCMC_DA) { tg.L_c = job->L_c; tg.L_f = job->L_f; tg.mod.L = job->L_f; }
  return run_chain(tg, theta0, job, out);
}

}  // extern "C"
