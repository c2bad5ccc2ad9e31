// api_filters.cpp -- the C ABI (include/bssm_amd.h), part 1: library / device entry points, the filters and smoothers of the
// ssm_ung / ssm_mng / gaussian families, the functor models (ssm_nlg, ssm_sde) and run-time compiled user models, suggest_N.
// There is no CPU compute path in this library: every filter runs on the HIP device or fails.
#include "api_internal.h"
#include "api_shared.h"
using namespace bssm_api;

namespace bssm { int set_error(int code, const std::string& msg) { return fail(code, msg); } }

namespace {
void transpose_to(const std::vector<double>& src, int rows, int cols, double* dst) {  // src rows x cols col-major -> cols x rows
  for (int j = 0; j < cols; j++) for (int i = 0; i < rows; i++) dst[j + (size_t)cols * i] = src[i + (size_t)rows * j];
}

int run_tables(const std::vector<bssm::FilterTables>& tabs, const std::vector<double>* fpar, int n, int m, int k, int p, int N,
               int psi /* 0 bootstrap, 1 psi-APF, 2 EKF proposal */, bool smoother, uint32_t seed, double base_ll, bssm_filter_out* out);

// one filter, full outputs: shared by bsf / bsf_smoother / psi_smoother
int run_single_filter(const bssm_ung_model* model, uint32_t nsim, uint32_t seed, bool psi, bool smoother,
                      bssm_filter_out* out) {
  if (!out) return fail(BSSM_ERR_INVALID, "out is NULL");
  if (nsim < 1) return fail(BSSM_ERR_INVALID, "nsim must be positive");
  bssm::HostModel hm;
  const std::string err = bssm::ingest_model(model, psi, &hm);
  if (!err.empty()) return fail(BSSM_ERR_INVALID, err);
  if (!have_device()) return no_device();
  const int n = hm.n, m = hm.m, N = (int)nsim;
  std::vector<bssm::FilterTables> tabs(1);
  double base_ll = 0.0;
  const int MP = bssm::pad_dim(m);
  if (psi) {
    bssm::PsiSetup su;
    if (!bssm::psi_setup(hm, nullptr, &su)) {  // non-finite Vt/Ct: the reference returns -Inf (model_ssm_nlg.cpp:988-990 analogue)
      out->logLik = -INFINITY;
      return BSSM_WARN_DEGENERATE;
    }
    bssm::build_psi_tables(hm, su, MP, &tabs[0]);
    base_ll = su.approx_loglik;  // src/model_ssm_ung.cpp:561
  } else {
    bssm::build_bsf_tables(hm, MP, bssm::pad_noise_dim(MP, hm.k), &tabs[0]);
    base_ll = bssm::const_term(hm);  // src/model_ssm_ung.cpp:690-710
  }
  return run_tables(tabs, nullptr, n, m, hm.k, hm.p, N, psi ? 1 : 0, smoother, seed, base_ll, out);
}

// one filter from prepared tables, full outputs (+ optional built-in functor parameters)
int run_tables(const std::vector<bssm::FilterTables>& tabs, const std::vector<double>* fpar, int n, int m, int k, int p, int N,
               int psi, bool smoother, uint32_t seed, double base_ll, bssm_filter_out* out) {
  DeviceBatch db;
  int rc = db.upload(tabs, n, m, k, p, N, psi, true, 0);
  if (rc) return rc;
  if (fpar) {  // functor parameters, optionally followed by the approximating (T_t, C_t) rows of the psi-APF
    std::vector<double> fp(fpar->begin(), fpar->begin() + bssm::FP_SIZE), am(fpar->begin() + bssm::FP_SIZE, fpar->end());
    if ((rc = db.set_fpar(fp))) return rc;
    if (!am.empty() && (rc = db.set_amats(am))) return rc;
  }
  // Everything runs on one non-blocking stream of this host thread; finished time slabs of weights / ancestors leave
  // through the output pipe while the filter is still stepping, the particle cube follows block by block after it.
  static thread_local bssm::TlsStream st_own;
  hipStream_t st = st_own.get();
  if (!st) return fail(BSSM_ERR_HIP, "no stream for the filter call");
  const size_t hist_elems = (size_t)(n + 1) * m * N;
  const size_t out_bytes = (out->alpha ? hist_elems * sizeof(double) : 0) + (out->weights ? db.weights.bytes : 0) +
                           (out->indices ? db.indices.bytes : 0);
  const bool piped = out_bytes >= ((size_t)8 << 20) && !bssm::opt().no_outpipe;
  bssm::StepTiming hooks;
  if (piped && db.use_grid && (out->weights || out->indices) && n >= 64) hooks.progress_every = std::max(16, n / 24);
  if ((rc = db.run(seed, g_call_stream0, st, &hooks))) return rc;
  // declared before the pipe: on an early return the pipe drains (its destructor) before these go back to the arena
  DevBuf cube, smooth, d_mean, d_cov, d_att, d_Ptt;
  // (BSSM_AMD_PROBE_NO_WEIGHTS drops the device weights: nothing to copy then)
  double* const out_weights = db.weights.p ? out->weights : nullptr;
  std::unique_ptr<OutPipe> pipe(piped ? new OutPipe(out_bytes) : nullptr);
  int slab_done = 0;  // weights / ancestor columns [0, slab_done) are on their way
  for (int k = 0; k < hooks.progress_count; k++) {
    const int t1 = hooks.progress_steps[k];
    if (out_weights && (rc = pipe->push(db.weights.as<double>() + (size_t)slab_done * N, out_weights + (size_t)slab_done * N,
                                         (size_t)(t1 - slab_done) * N * sizeof(double), hooks.progress_events[k]))) return rc;
    if (out->indices && (rc = pipe->push(db.indices.as<uint32_t>() + (size_t)slab_done * N, out->indices + (size_t)slab_done * N,
                                         (size_t)(t1 - slab_done) * N * sizeof(uint32_t), hooks.progress_events[k]))) return rc;
    slab_done = t1;
  }
  for (int k = 0; k < hooks.progress_count; k++) (void)hipEventDestroy(hooks.progress_events[k]);
  if (pipe) {  // the rest (the launcher's stream order covers it)
    if (out_weights && (rc = pipe->push(db.weights.as<double>() + (size_t)slab_done * N, out_weights + (size_t)slab_done * N,
                                         (size_t)(n + 1 - slab_done) * N * sizeof(double), nullptr, st))) return rc;
    if (out->indices && (rc = pipe->push(db.indices.as<uint32_t>() + (size_t)slab_done * N, out->indices + (size_t)slab_done * N,
                                         (size_t)(n - slab_done) * N * sizeof(uint32_t), nullptr, st))) return rc;
  }
  double ll = 0;
  int dead = 0;
  HIP_TRY(hipMemcpyAsync(&ll, db.loglik.p, sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(&dead, db.dead.p, sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  const bool degenerate = !std::isfinite(ll);
  out->logLik = degenerate ? -INFINITY : ll + base_ll;
  if (pipe && dead != 0 && slab_done > 0) {
    // a degenerate filter: the slots after the dead step are zero-filled at the END of the run (launch_zero_dead_tail),
    // so slabs that left earlier may hold stale values there -- send those columns again
    if (out_weights && (rc = pipe->push(db.weights.p, out_weights, (size_t)slab_done * N * sizeof(double), nullptr, st))) return rc;
    if (out->indices && (rc = pipe->push(db.indices.p, out->indices, (size_t)slab_done * N * sizeof(uint32_t), nullptr, st))) return rc;
  }

  const double* final_hist = db.hist.as<double>();
  if (smoother) {  // filter_smoother + summary (src/R_bootstrap_filter.cpp:368-369, src/R_psi.cpp:94-95)
    HIP_TRY(smooth.alloc(hist_elems * sizeof(double)));
    HIP_TRY(bssm::launch_backtrace(db.hist.as<double>(), db.indices.as<uint32_t>(), smooth.as<double>(), 1, n, m, N, st));
    final_hist = smooth.as<double>();
  }
  if (out->alpha) {  // the cube first: its transfer is the long pole, the summaries below run under it
    HIP_TRY(cube.alloc(hist_elems * sizeof(double)));
    HIP_TRY(bssm::launch_to_cube(final_hist, cube.as<double>(), 1, n + 1, m, N, st));
    if (pipe) {
      if ((rc = pipe->push(cube.p, out->alpha, cube.bytes, nullptr, st))) return rc;
    } else {
      HIP_TRY(hipMemcpyAsync(out->alpha, cube.p, cube.bytes, hipMemcpyDeviceToHost, st));
    }
  }
  if (smoother) {
    if (out->alphahat || out->Vt) {
      HIP_TRY(d_mean.alloc((size_t)(n + 1) * m * sizeof(double)));
      HIP_TRY(d_cov.alloc((size_t)(n + 1) * m * m * sizeof(double)));
      HIP_TRY(bssm::launch_summary(final_hist, d_mean.as<double>(), d_cov.as<double>(), 1, n + 1, m, N, st));
      std::vector<double> hmean((size_t)(n + 1) * m);
      HIP_TRY(hipMemcpyAsync(hmean.data(), d_mean.p, d_mean.bytes, hipMemcpyDeviceToHost, st));
      if (out->Vt) HIP_TRY(hipMemcpyAsync(out->Vt, d_cov.p, d_cov.bytes, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      if (out->alphahat) transpose_to(hmean, m, n + 1, out->alphahat);  // arma::inplace_trans(alphahat)
    }
  } else if (out->at || out->att || out->Pt || out->Ptt) {  // filter_summary (src/R_bootstrap_filter.cpp:143-147)
    HIP_TRY(d_mean.alloc((size_t)(n + 1) * m * sizeof(double)));
    HIP_TRY(d_att.alloc((size_t)n * m * sizeof(double)));
    HIP_TRY(d_cov.alloc((size_t)(n + 1) * m * m * sizeof(double)));
    HIP_TRY(d_Ptt.alloc((size_t)n * m * m * sizeof(double)));
    HIP_TRY(bssm::launch_filter_summary(final_hist, db.weights.as<double>(), d_mean.as<double>(), d_att.as<double>(),
                                        d_cov.as<double>(), d_Ptt.as<double>(), 1, n, m, N, st));
    std::vector<double> h1((size_t)(n + 1) * m), h2((size_t)n * m);
    HIP_TRY(hipMemcpyAsync(h1.data(), d_mean.p, d_mean.bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h2.data(), d_att.p, d_att.bytes, hipMemcpyDeviceToHost, st));
    if (out->Pt) HIP_TRY(hipMemcpyAsync(out->Pt, d_cov.p, d_cov.bytes, hipMemcpyDeviceToHost, st));
    if (out->Ptt) HIP_TRY(hipMemcpyAsync(out->Ptt, d_Ptt.p, d_Ptt.bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (out->at) transpose_to(h1, m, n + 1, out->at);
    if (out->att) transpose_to(h2, m, n, out->att);
  }
  if (!pipe) {
    if (out_weights) HIP_TRY(hipMemcpyAsync(out_weights, db.weights.p, db.weights.bytes, hipMemcpyDeviceToHost, st));
    if (out->indices) HIP_TRY(hipMemcpyAsync(out->indices, db.indices.p, db.indices.bytes, hipMemcpyDeviceToHost, st));
  }
  HIP_TRY(hipStreamSynchronize(st));
  if (pipe && (rc = pipe->finish())) return rc;
  HIP_TRY(hipDeviceSynchronize());
  return degenerate ? BSSM_WARN_DEGENERATE : BSSM_OK;
}

}  // namespace

// =====================================================================================
extern "C" {

int bssm_amd_device_count(void) {
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
  return cnt;
}
const char* bssm_amd_last_error(void) { return g_err.c_str(); }
// (stream format 2 = PACKED normals for draws of more than two, round 4; normalisation 2 = pf_canon.h rows for N <= 8192, round 5:
// results for a given seed are not bit-reproducible across these numbers -- the parity bar against the reference is unaffected)
const char* bssm_amd_version(void) { return "bssm_amd 0.5 (gfx950; random-stream format 2, normalisation 2)"; }
// 0: not calibrated, 1: the current device's tail stream was measured to run beside its null stream, 2: measured, none of the candidates did
int bssm_amd_tail_overlap_state(int32_t* candidates_tried) {
  (void)have_device();
  int tried = 0;
  const int st = bssm::tail_overlap_state(&tried);
  if (candidates_tried) *candidates_tried = tried;
  return st;
}
void bssm_amd_trim_cache(void) { arena().trim(); stage_arena().trim(); }
void bssm_amd_last_is_timing(double out[2]) { out[0] = g_is_timing[0]; out[1] = g_is_timing[1]; }
void bssm_amd_last_is_timing4(double out[4]) { for (int q = 0; q < 4; q++) out[q] = g_is_timing4[q]; }

void bssm_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  const bssm::u32x4 r = bssm::philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
  out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

int bssm_bsf(const bssm_ung_model* model, uint32_t nsim, uint32_t seed, int gaussian, int model_type,
             bssm_filter_out* out) {
  if (gaussian && model_type == 0)
    return fail(BSSM_ERR_UNSUPPORTED, "Bootstrap particle filtering for multivariate gaussian model is not yet supported. ");
  if (!is_gaussian_type_ok(gaussian, model_type)) return fail(BSSM_ERR_UNSUPPORTED, "model_type not handled by this entry point");
  // (a multivariate model's `distribution` is ignored in favour of its per-series codes -- include/bssm_amd.h -- and a Gaussian series there is no
  // Gaussian MODEL: src/model_ssm_mng.cpp:407-442)
  if (model && model->p <= 1 && (gaussian != 0) != (model->distribution == BSSM_GAUSSIAN))
    return fail(BSSM_ERR_INVALID, "gaussian flag and model->distribution disagree");
  return run_single_filter(model, nsim, seed, false, false, out);
}

int bssm_bsf_smoother(const bssm_ung_model* model, uint32_t nsim, uint32_t seed, int gaussian, int model_type,
                      bssm_filter_out* out) {
  if (gaussian && model_type == 0)
    return fail(BSSM_ERR_UNSUPPORTED, "Bootstrap particle smoothing for multivariate gaussian model is not yet supported. ");
  if (!is_gaussian_type_ok(gaussian, model_type)) return fail(BSSM_ERR_UNSUPPORTED, "model_type not handled by this entry point");
  // (a multivariate model's `distribution` is ignored in favour of its per-series codes -- include/bssm_amd.h -- and a Gaussian series there is no
  // Gaussian MODEL: src/model_ssm_mng.cpp:407-442)
  if (model && model->p <= 1 && (gaussian != 0) != (model->distribution == BSSM_GAUSSIAN))
    return fail(BSSM_ERR_INVALID, "gaussian flag and model->distribution disagree");
  return run_single_filter(model, nsim, seed, false, true, out);
}

int bssm_psi_smoother(const bssm_ung_model* model, uint32_t nsim, uint32_t seed, int model_type, bssm_filter_out* out) {
  if (!is_gaussian_type_ok(0, model_type)) return fail(BSSM_ERR_UNSUPPORTED, "model_type not handled by this entry point");
  return run_single_filter(model, nsim, seed, true, true, out);
}

// ---------------------------------------------------------------- built-in non-linear / SDE models
}  // extern "C"
namespace bssm_api {   // (functor_tables, to_fn_model, functor_psi_tables are called from api_postcorrect.cpp too)
// length of the interval one Milstein call covers: 1 in every filter (src/model_ssm_sde.cpp:27,61); bssm_milstein sets its caller's t
static thread_local double g_milstein_tend = 1.0;
// known_tv_params of user models (BSSM_USER_TV) in device memory: uploaded once per (device, content) and kept -- the
// parameter blocks of every filter (and of device-resident plans) carry the device address, and a matrix is a few KB.  Entries
// are found by a hash and CONFIRMED by comparing dimensions and content; the cache is bounded (256 MiB per process), beyond
// which the call fails instead of growing without limit.
int device_known_tv(const double* host, int rows, int cols, double** dev) {
  struct Entry { int rows, cols; std::vector<double> copy; double* p; };
  static std::mutex mu;
  static std::multimap<std::pair<int, uint64_t>, Entry> cache;
  static size_t kept_bytes = 0;
  int d = 0;
  (void)hipGetDevice(&d);
  const size_t cnt = (size_t)rows * cols;
  uint64_t h = 1469598103934665603ull ^ ((uint64_t)rows << 32 | (uint32_t)cols);
  for (size_t q = 0; q < cnt; q++) { uint64_t b; std::memcpy(&b, &host[q], 8); h = (h ^ b) * 1099511628211ull; }
  std::lock_guard<std::mutex> lk(mu);
  auto range = cache.equal_range({d, h});
  for (auto it = range.first; it != range.second; ++it) {
    const Entry& e = it->second;
    if (e.rows == rows && e.cols == cols && std::memcmp(e.copy.data(), host, cnt * sizeof(double)) == 0) { *dev = e.p; return BSSM_OK; }
  }
  if (kept_bytes + cnt * sizeof(double) > ((size_t)256 << 20))
    return fail(BSSM_ERR_UNSUPPORTED, "more than 256 MiB of distinct known_tv_params matrices kept on the device by this process");
  double* p = nullptr;
  HIP_TRY(hipMalloc((void**)&p, cnt * sizeof(double)));
  const hipError_t ce = hipMemcpy(p, host, cnt * sizeof(double), hipMemcpyHostToDevice);
  if (ce != hipSuccess) { (void)hipFree(p); return fail(BSSM_ERR_HIP, std::string("known_tv_params upload: ") + hipGetErrorString(ce)); }
  cache.emplace(std::make_pair(d, h), Entry{rows, cols, std::vector<double>(host, host + cnt), p});
  kept_bytes += cnt * sizeof(double);
  *dev = p;
  return BSSM_OK;
}
// params_only (may be NULL): stop after the parameter block and hand back a1 (2) and P1 (4) -- what the device set-up needs
int functor_tables(const bssm_fn_model* mod, bssm::FilterTables* ft, std::vector<double>* fpar, int* m_out, int* k_out,
                   double* params_only) {
  if (!mod || !mod->y || !mod->theta) return fail(BSSM_ERR_INVALID, "model, y, theta must be non-NULL");
  const int id = mod->model_id, n = mod->n;
  if (id < bssm::FN_NLG_GROWTH || id > bssm::FN_USER_SDE) return fail(BSSM_ERR_UNSUPPORTED, "unknown model id");
  const bool user = id >= bssm::FN_USER_NLG;
  if (n < 1) return fail(BSSM_ERR_INVALID, "n must be positive");
  const int need_theta[] = {0, 3, 4, 2, 1, 3, 3, 0, 0}, need_known[] = {0, 6, 0, 0, 0, 0, 0, 0, 0};
  if (mod->n_theta < need_theta[id] || mod->n_known < need_known[id]) return fail(BSSM_ERR_INVALID, "theta / known_params too short for this model");
  if (mod->n_theta > bssm::FP_LONG_MAX || mod->n_known > bssm::FP_LONG_MAX) return fail(BSSM_ERR_INVALID, "at most 64 theta / known_params");
  if (!user && (mod->n_theta > 8 || mod->n_known > 8)) return fail(BSSM_ERR_INVALID, "the shipped models take at most 8 theta / known_params");
  if (user && mod->user_slot < 1) return fail(BSSM_ERR_INVALID, "user-defined model without a loaded library (bssm_user_model_load)");
  if (id == bssm::FN_USER_NLG && (mod->n_states < 1 || mod->n_states > 16)) return fail(BSSM_ERR_UNSUPPORTED, "user-defined ssm_nlg models: n_states must be 1 ... 16");
  if (id == bssm::FN_USER_NLG && mod->n_states > 2 && !mod->use_ext) return fail(BSSM_ERR_INVALID, "user-defined ssm_nlg model with more than 2 states: fill R_ext / a1_ext / P1_ext and set use_ext");
  if (id == bssm::FN_USER_NLG && !(mod->H > 0)) return fail(BSSM_ERR_INVALID, "user-defined ssm_nlg model: H must be positive");
  const bool ext = id == bssm::FN_USER_NLG && mod->use_ext;
  if (ext && (mod->n_etas < 1 || mod->n_etas > mod->n_states)) return fail(BSSM_ERR_INVALID, "user-defined ssm_nlg model: n_etas must be in 1 ... n_states");
  const int m = id == bssm::FN_NLG_GROWTH ? 2 : (id == bssm::FN_USER_NLG ? mod->n_states : 1), MP = bssm::pad_dim(m);
  *m_out = m; *k_out = m;
  const double* th = mod->theta;
  const double* kp = mod->known_params;
  fpar->assign(bssm::FP_SIZE, 0.0);
  double* fp = fpar->data();
  fp[bssm::FP_ID] = id; fp[bssm::FP_L] = mod->L; fp[bssm::FP_POS] = mod->positive ? 1.0 : 0.0; fp[bssm::FP_X0] = mod->x0;
  fp[bssm::FP_TEND] = g_milstein_tend;
  {  // theta / known_params: in the header up to 8 entries each, else in the tail behind the two matrices (pf_functors.h)
    int tail = bssm::fp_size(MP);
    const int thoff = mod->n_theta > 8 ? tail : bssm::FP_THETA;
    if (mod->n_theta > 8) tail += mod->n_theta;
    const int knoff = mod->n_known > 8 ? tail : bssm::FP_KNOWN;
    fp[bssm::FP_THOFF] = thoff; fp[bssm::FP_KNOFF] = knoff;
    for (int q = 0; q < mod->n_theta; q++) fp[thoff + q] = th[q];
    for (int q = 0; q < mod->n_known; q++) fp[knoff + q] = kp[q];
  }
  double a1[2] = {0, 0}, P1[4] = {0, 0, 0, 0}, H = 1.0;
  switch (id) {  // R/cpp_example_models.R: a1_fn, P1_fn, H_fn, R_fn of each snippet
    case bssm::FN_NLG_GROWTH: a1[0] = kp[2]; a1[1] = kp[3]; P1[0] = kp[4]; P1[3] = kp[5]; H = std::exp(th[0]);
      fp[bssm::FP_R] = std::exp(th[1]); fp[bssm::FP_R + 1] = std::exp(th[2]); break;
    case bssm::FN_NLG_AR_EXP: a1[0] = th[0]; P1[0] = std::pow(std::exp(th[2]), 2) / (1 - std::pow(th[1], 2)); H = std::exp(th[3]);
      fp[bssm::FP_R] = std::exp(th[2]); break;
    case bssm::FN_NLG_SIN_EXP: a1[0] = 0; P1[0] = 1; H = std::exp(th[0]); fp[bssm::FP_R] = std::exp(th[1]); break;
    case bssm::FN_NLG_LINEAR_GAUSSIAN: a1[0] = 0; P1[0] = 1; H = std::exp(th[0]); fp[bssm::FP_R] = 1.0; break;
    case bssm::FN_SDE_GBM: H = th[2]; if (mod->L < 0 || mod->L > 20) return fail(BSSM_ERR_INVALID, "L must be in 0..20"); break;
    case bssm::FN_SDE_POISSON_OU: fp[bssm::FP_R] = std::exp(th[2]); if (mod->L < 0 || mod->L > 20) return fail(BSSM_ERR_INVALID, "L must be in 0..20"); break;
    case bssm::FN_USER_NLG:
      if (!ext) {
        for (int j = 0; j < m; j++) { a1[j] = mod->a1[j]; fp[bssm::FP_R + j] = mod->R_diag[j]; }
        for (int q = 0; q < m * m; q++) P1[q] = mod->P1[q];
      }
      H = mod->H;
      break;
    case bssm::FN_USER_SDE: if (mod->L < 0 || mod->L > 20) return fail(BSSM_ERR_INVALID, "L must be in 0..20"); break;
  }
  fp[bssm::FP_SLOT] = user ? mod->user_slot : 0;
  fp[bssm::FP_M] = m;
  if (id == bssm::FN_USER_NLG) {  // full R, MP x MP column-major, zero-padded (the user-model kernels read this block)
    if (ext) { for (int c = 0; c < mod->n_etas; c++) for (int j = 0; j < m; j++) fp[bssm::FP_RMAT + j + MP * c] = mod->R_ext[j + m * c]; }
    else for (int j = 0; j < m; j++) fp[bssm::FP_RMAT + j + MP * j] = mod->R_diag[j];
  }
  if (id == bssm::FN_USER_NLG) {
    // inverse lower Cholesky factor of R R' on its non-zero diagonal (the psi-APF's transition-density ratio, pf_functors.h);
    // NaN when R R' is not positive definite there -- only the psi-APF reads it, and its host set-up then fails loudly
    constexpr int S = 16;  // stride of the work arrays: the largest state dimension of a user model
    double RR[S * S] = {0}, Lc[S * S] = {0}, Li[S * S] = {0};
    const int kk = ext ? mod->n_etas : m;
    for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) {
      double acc = 0;
      for (int q = 0; q < kk; q++) acc += fp[bssm::FP_RMAT + i + MP * q] * fp[bssm::FP_RMAT + j + MP * q];
      RR[i + S * j] = acc;
    }
    int nz[S], cnt = 0;
    for (int j = 0; j < m; j++) if (RR[j + S * j] > 2.220446049250313e-16) nz[cnt++] = j;
    bool pd = true;
    for (int a = 0; a < cnt && pd; a++)
      for (int b2 = 0; b2 <= a; b2++) {
        double v = RR[nz[a] + S * nz[b2]];
        for (int q = 0; q < b2; q++) v -= Lc[a + S * q] * Lc[b2 + S * q];
        if (a == b2) { if (!(v > 0)) { pd = false; break; } Lc[a + S * a] = std::sqrt(v); }
        else Lc[a + S * b2] = v / Lc[b2 + S * b2];
      }
    for (int c = 0; c < cnt && pd; c++) {  // forward substitution column by column: Lc Li = I
      for (int a = c; a < cnt; a++) {
        double v = a == c ? 1.0 : 0.0;
        for (int q = c; q < a; q++) v -= Lc[a + S * q] * Li[q + S * c];
        Li[a + S * c] = v / Lc[a + S * a];
      }
    }
    for (int a = 0; a < cnt; a++) for (int c = 0; c <= a; c++)
      fp[bssm::fp_rlinv(MP) + nz[a] + MP * nz[c]] = pd ? Li[a + S * c] : std::nan("");
    if (mod->known_tv_params && mod->tv_rows > 0 && mod->tv_cols > 0 && have_device()) {  // BSSM_USER_TV snippets: matrix on the device
      double* dtv = nullptr;
      if (int rc = device_known_tv(mod->known_tv_params, mod->tv_rows, mod->tv_cols, &dtv)) return rc;
      uint64_t bits = (uint64_t)(uintptr_t)dtv;
      std::memcpy(&fp[bssm::FP_TVPTR], &bits, sizeof bits);
      fp[bssm::FP_TVROWS] = mod->tv_rows; fp[bssm::FP_TVCOLS] = mod->tv_cols;
    }
  }
  if (id >= bssm::FN_NLG_GROWTH && id <= bssm::FN_NLG_LINEAR_GAUSSIAN)  // shipped models: diagonal R, the psi weight reads 1 / R_j
    for (int j = 0; j < m; j++) fp[bssm::fp_rlinv(MP) + j + MP * j] = 1.0 / fp[bssm::FP_R + j];
  fp[bssm::FP_H] = H;
  fp[bssm::FP_LOGC] = -0.5 * (std::log(2.0 * 3.14159265358979323846) + 2.0 * std::log(H));
  if (params_only) {
    if (ext) return fail(BSSM_ERR_UNSUPPORTED, "the psi-APF / EKF particle filter of a user model take R_diag / a1 / P1 (at most 2 states, constant diagonal R)");
    for (int j = 0; j < 2; j++) params_only[j] = a1[j];
    for (int q = 0; q < 4; q++) params_only[2 + q] = P1[q];
    return BSSM_OK;
  }
  // tables: no linear transition operands; one observation series carrying y (and lgamma(y + 1) for the Poisson model)
  ft->MP = MP; ft->KP = MP; ft->mats_tstride = 0;
  ft->mats.assign((size_t)MP * MP * 2 + MP, 0.0);
  const int stride = bssm::ROW_HDR + bssm::SER_HDR + MP;
  ft->obs.assign((size_t)n * stride, 0.0);
  for (int t = 0; t < n; t++) {
    double* row = ft->obs.data() + (size_t)t * stride;
    const double y = mod->y[t];
    if (!std::isfinite(y)) continue;
    row[bssm::RH_FLAG] = 1.0;
    double* ob = row + bssm::ROW_HDR;
    ob[bssm::OB_FLAG] = 1.0; ob[bssm::OB_Y] = y;
    ob[bssm::OB_AUX] = id == bssm::FN_SDE_POISSON_OU ? std::lgamma(y + 1.0) : 0.0;
  }
  // initial factor: psd_chol(P1) (src/model_ssm_nlg.cpp:1072); the shipped P1 are diagonal
  ft->init.assign((size_t)MP + (size_t)MP * MP, 0.0);
  if (ext) {  // psd_chol of any P1 (src/psd_chol.cpp:5-18): Cholesky factor of the rows / columns with a non-zero diagonal
    const double eps = 2.220446049250313e-16;
    double dmax = mod->P1_ext[0];
    for (int q = 1; q < m; q++) dmax = std::max(dmax, mod->P1_ext[q + m * q]);
    const double thr = std::max(eps, eps * m * dmax);
    constexpr int S = 16;
    bool nz[S];
    double Lf[S * S] = {0};
    for (int j = 0; j < m; j++) nz[j] = mod->P1_ext[j + m * j] > thr;
    for (int j = 0; j < m; j++) {
      ft->init[j] = mod->a1_ext[j];
      if (!nz[j]) continue;
      double d = mod->P1_ext[j + m * j];
      for (int q = 0; q < j; q++) d -= Lf[j + S * q] * Lf[j + S * q];
      if (!(d > 0)) return fail(BSSM_ERR_INVALID, "P1 is not positive semi-definite");
      Lf[j + S * j] = std::sqrt(d);
      for (int i = j + 1; i < m; i++) {
        if (!nz[i]) continue;
        double v = mod->P1_ext[i + m * j];
        for (int q = 0; q < j; q++) v -= Lf[i + S * q] * Lf[j + S * q];
        Lf[i + S * j] = v / Lf[j + S * j];
      }
    }
    for (int c = 0; c < m; c++) for (int j = 0; j < m; j++) ft->init[MP + j + MP * c] = Lf[j + S * c];
    return BSSM_OK;
  }
  if (id == bssm::FN_USER_NLG && m == 2 && (P1[1] != 0.0 || P1[2] != 0.0)) {  // a full 2 x 2 prior covariance
    const double thr = std::max(2.220446049250313e-16, 2.220446049250313e-16 * 2 * std::max(P1[0], P1[3]));
    ft->init[0] = a1[0]; ft->init[1] = a1[1];
    if (P1[0] > thr && P1[3] > thr) {
      const double l00 = std::sqrt(P1[0]), l10 = P1[1] / l00, d = P1[3] - l10 * l10;
      if (!(d > 0)) return fail(BSSM_ERR_INVALID, "P1 is not positive semi-definite");
      ft->init[MP + 0] = l00; ft->init[MP + 1] = l10; ft->init[MP + 3] = std::sqrt(d);
    } else {
      if (P1[0] > thr) ft->init[MP + 0] = std::sqrt(P1[0]);
      if (P1[3] > thr) ft->init[MP + 3] = std::sqrt(P1[3]);
    }
    return BSSM_OK;
  }
  for (int j = 0; j < m; j++) {
    ft->init[j] = a1[j];
    const double d = P1[j + m * j];
    double dmax = P1[0];
    for (int q = 1; q < m; q++) dmax = std::max(dmax, P1[q + m * q]);
    const double thr = std::max(2.220446049250313e-16, 2.220446049250313e-16 * m * dmax);
    ft->init[MP + j + MP * j] = d > thr ? std::sqrt(d) : 0.0;
  }
  return BSSM_OK;
}

// max_states: 8 for the psi-APF and the EKF particle filter of user models (any constant R with R R' positive definite on its non-zero
// diagonal, any P1; more than four states since late round 4)
int to_fn_model(const bssm_fn_model* mod, bssm::FnModel* f, int max_states) {
  if (!mod || !mod->y || !mod->theta) return fail(BSSM_ERR_INVALID, "model, y, theta must be non-NULL");
  const bool user = mod->model_id == bssm::FN_USER_NLG;
  if (!user && (mod->model_id < bssm::FN_NLG_GROWTH || mod->model_id > bssm::FN_NLG_LINEAR_GAUSSIAN))
    return fail(BSSM_ERR_UNSUPPORTED, "the psi-APF and the EKF particle filter are for ssm_nlg models");
  if (mod->n_theta > bssm::FP_LONG_MAX || mod->n_known > bssm::FP_LONG_MAX) return fail(BSSM_ERR_INVALID, "at most 64 theta / known_params");
  f->id = mod->model_id; f->n = mod->n; f->m = mod->model_id == bssm::FN_NLG_GROWTH ? 2 : (user ? mod->n_states : 1);
  if (user) {
    if (mod->user_slot < 1 || mod->user_slot > 63) return fail(BSSM_ERR_INVALID, "user-defined model without a loaded library");
    const bssm::UserHostFns& h = g_user_host[mod->user_slot];
    const int ms = mod->n_states;
    if (ms < 1 || ms > max_states)
      return fail(BSSM_ERR_UNSUPPORTED, max_states > 8 ? "psi-APF of a user model: at most 16 states" : "EKF particle filter of a user model (and the psi-APF of one with a state-dependent R_fn): at most 8 states");
    // (a model compiled in process -- bssm_user_model_jit -- has no host code: its T / Z / Jacobians reach the host set-up through a one-thread
    // evaluation kernel of the same snippet, round 6)
    if (!h.T || !h.Z || !h.Tg || !h.Zg)
      return fail(BSSM_ERR_UNSUPPORTED, "this user model was compiled without BSSM_USER_JACOBIANS (bssm_user_Tg, bssm_user_Zg): the psi-APF and the EKF particle filter need them");
    // the state-independent numbers: R_diag / a1 / P1 (up to 2 states) or the R_ext / a1_ext / P1_ext blocks
    const double* Rsrc = mod->use_ext ? mod->R_ext : nullptr;
    const double* asrc = mod->use_ext ? mod->a1_ext : mod->a1;
    const double* Psrc = mod->use_ext ? mod->P1_ext : mod->P1;
    const int ks = mod->use_ext ? mod->n_etas : ms;
    if (max_states == 2) {  // EKF particle filter: its device code factors 2 x 2 covariances built from a diagonal R
      if (ks != ms) return fail(BSSM_ERR_UNSUPPORTED, "EKF particle filter of a user model: constant diagonal R (n_etas = n_states)");
      for (int c = 0; c < ms; c++) for (int j = 0; j < ms; j++) {
        if (j == c) continue;
        if (Rsrc && Rsrc[j + ms * c] != 0.0) return fail(BSSM_ERR_UNSUPPORTED, "EKF particle filter of a user model: constant diagonal R");
        if (Psrc[j + ms * c] != 0.0) return fail(BSSM_ERR_UNSUPPORTED, "EKF particle filter of a user model: P1 must be diagonal");
      }
    }
    f->user = &h;
    f->k = ks;
    f->uH = mod->H;
    for (int q = 0; q < 256; q++) f->uRm[q] = 0.0;
    if (Rsrc) { for (int c = 0; c < ks; c++) for (int j = 0; j < ms; j++) f->uRm[j + ms * c] = Rsrc[j + ms * c]; }  // m x k, zero columns behind
    else for (int j = 0; j < ms; j++) f->uRm[j + ms * j] = mod->R_diag[j];
    for (int j = 0; j < ms; j++) { f->uR[j] = f->uRm[j + ms * j]; f->ua1[j] = asrc[j]; }
    for (int q = 0; q < ms * ms; q++) f->uP1[q] = Psrc[q];
    f->known_tv = mod->known_tv_params; f->tv_rows = mod->tv_rows; f->tv_cols = mod->tv_cols;
    if (h.has_tv && (!mod->known_tv_params || mod->tv_rows < 1 || mod->tv_cols < 1))
      return fail(BSSM_ERR_INVALID, "this user model was compiled with BSSM_USER_TV: known_tv_params (tv_rows x tv_cols) must be given");
  }
  f->y.assign(mod->y, mod->y + mod->n);
  for (int q = 0; q < mod->n_theta; q++) f->theta[q] = mod->theta[q];
  for (int q = 0; q < mod->n_known; q++) f->known[q] = mod->known_params[q];
  return BSSM_OK;
}

// psi-APF of a built-in ssm_nlg model: tables + functor parameters + approximating (T_t, C_t)
int functor_psi_tables(const bssm_fn_model* mod, const double* is_mode, int max_iter, double conv_tol, bssm::FilterTables* ft,
                       std::vector<double>* fpar, std::vector<double>* amats, double* approx_loglik, bool* ok, int* m_out) {
  bssm::FnModel f;
  int rc = to_fn_model(mod, &f, 16);   // (psi-APF: up to 16 states since round 6; a snippet with a state-dependent R_fn has no psi kernels beyond 8)
  if (rc) return rc;
  bssm::FilterTables dummy;
  int m = 1, k = 1;
  if ((rc = functor_tables(mod, &dummy, fpar, &m, &k))) return rc;  // fpar block (id, theta, H, R ...)
  for (int q = 0, MPq = bssm::pad_dim(m); q < MPq * MPq; q++)
    if (std::isnan((*fpar)[bssm::fp_rlinv(MPq) + q])) return fail(BSSM_ERR_INVALID, "psi-APF of a user model: R R' must be positive definite on its non-zero diagonal");
  bssm::FnSetup su;
  *ok = bssm::fn_psi_setup(f, is_mode, max_iter, conv_tol, &su);
  *approx_loglik = su.approx_loglik;
  *m_out = m;
  if (*ok) bssm::build_fn_psi_tables(f, su, ft, amats);
  return BSSM_OK;
}

int run_functor(const bssm_fn_model* mod, uint32_t nsim, uint32_t seed, bool smoother, bssm_filter_out* out) {
  if (!out) return fail(BSSM_ERR_INVALID, "out is NULL");
  if (nsim < 1) return fail(BSSM_ERR_INVALID, "nsim must be positive");
  std::vector<bssm::FilterTables> tabs(1);
  std::vector<double> fpar;
  int m = 1, k = 1;
  const int rc = functor_tables(mod, &tabs[0], &fpar, &m, &k);
  if (rc) return rc;
  if (!have_device()) return no_device();
  return run_tables(tabs, &fpar, mod->n, m, k, 1, (int)nsim, 0, smoother, seed, 0.0, out);
}
}  // namespace bssm_api
extern "C" {

namespace {
// EKF-proposal particle filter of a built-in ssm_nlg model: bootstrap-style tables, the initial proposal (att1, L1) as
// the initial table, and [a1 | P1 | att1 | L1] behind the functor parameters for the time-0 weights
int functor_ekpf_tables(const bssm_fn_model* mod, bssm::FilterTables* ft, std::vector<double>* fpar, int* m_out) {
  bssm::FnModel f;
  int rc = to_fn_model(mod, &f);  // (up to 8 states since late round 4, any constant R, any P1 -- as the psi-APF)
  if (rc) return rc;
  int m = 1, k = 1;
  if ((rc = functor_tables(mod, ft, fpar, &m, &k))) return rc;
  const int MP = bssm::pad_dim(m);
  if (mod->model_id == bssm::FN_USER_NLG && m <= 2) {  // the 1- / 2-state device code builds R R' from the diagonal of R
    for (int c = 0; c < m; c++) for (int j = 0; j < m; j++)
      if (j != c && f.uRm[j + m * c] != 0.0) return fail(BSSM_ERR_UNSUPPORTED, "EKF particle filter of a user model with 1 or 2 states: constant diagonal R");
  }
  double a1[8], P1[64], att1[8], L1[64];
  bssm::fn_ekpf_init(f, a1, P1, att1, L1);
  // the initial table and the [a1 | P1 | att1 | L1] block in the kernels' padded layout (MP, MP x MP column-major)
  std::fill(ft->init.begin(), ft->init.end(), 0.0);
  std::vector<double> blk((size_t)2 * MP + 2 * MP * MP, 0.0);
  for (int j = 0; j < m; j++) { ft->init[j] = att1[j]; blk[j] = a1[j]; blk[MP + MP * MP + j] = att1[j]; }
  for (int c = 0; c < m; c++) for (int j = 0; j < m; j++) {
    ft->init[MP + j + MP * c] = L1[j + m * c];
    blk[MP + j + MP * c] = P1[j + m * c];
    blk[2 * MP + MP * MP + j + MP * c] = L1[j + m * c];
  }
  fpar->insert(fpar->end(), blk.begin(), blk.end());
  *m_out = m;
  return BSSM_OK;
}
int run_ekpf(const bssm_fn_model* mod, uint32_t nsim, uint32_t seed, bool smoother, bssm_filter_out* out) {
  if (!out || nsim < 1) return fail(BSSM_ERR_INVALID, "bad arguments");
  std::vector<bssm::FilterTables> tabs(1);
  std::vector<double> fpar;
  int m = 1;
  const int rc = functor_ekpf_tables(mod, &tabs[0], &fpar, &m);
  if (rc) return rc;
  if (!have_device()) return no_device();
  return run_tables(tabs, &fpar, mod->n, m, m, 1, (int)nsim, 2, smoother, seed, 0.0, out);
}
}  // namespace

int bssm_ekpf_nlg(const bssm_fn_model* model, uint32_t nsim, uint32_t seed, bssm_filter_out* out) {
  return run_ekpf(model, nsim, seed, false, out);
}
int bssm_ekpf_smoother_nlg(const bssm_fn_model* model, uint32_t nsim, uint32_t seed, bssm_filter_out* out) {
  return run_ekpf(model, nsim, seed, true, out);
}

int bssm_bsf_nlg(const bssm_fn_model* model, uint32_t nsim, uint32_t seed, bssm_filter_out* out) {
  if (model && bssm::fn_is_sde(model->model_id)) return fail(BSSM_ERR_INVALID, "not an ssm_nlg model id");
  return run_functor(model, nsim, seed, false, out);
}
int bssm_bsf_smoother_nlg(const bssm_fn_model* model, uint32_t nsim, uint32_t seed, bssm_filter_out* out) {
  if (model && bssm::fn_is_sde(model->model_id)) return fail(BSSM_ERR_INVALID, "not an ssm_nlg model id");
  return run_functor(model, nsim, seed, true, out);
}
int bssm_bsf_sde(const bssm_fn_model* model, uint32_t nsim, uint32_t seed, bssm_filter_out* out) {
  if (model && !bssm::fn_is_sde(model->model_id)) return fail(BSSM_ERR_INVALID, "not an ssm_sde model id");
  return run_functor(model, nsim, seed, false, out);
}
// R_milstein (src/R_milstein.cpp:4-22 -> milstein(), src/milstein.cpp:18-34): the terminal value of ONE Milstein path from model->x0
// over [0, t_end] in 2^L steps.  It is the prior draw of particle 0 of a bootstrap filter with that seed (src/model_ssm_sde.cpp:25-28
// calls the same function with t = 1), so it runs through the filter kernels themselves -- built-in and user-defined models alike.
int bssm_milstein(const bssm_fn_model* model, double t_end, uint32_t seed, double* x_out) {
  if (!model || !x_out) return fail(BSSM_ERR_INVALID, "NULL argument");
  if (!bssm::fn_is_sde(model->model_id)) return fail(BSSM_ERR_INVALID, "not an ssm_sde model id");
  if (!(t_end > 0) || !std::isfinite(t_end)) return fail(BSSM_ERR_INVALID, "t must be positive");
  bssm_fn_model one = *model;
  const double missing = std::nan("");
  one.n = 1; one.y = &missing;
  constexpr int N = 64;
  std::vector<double> alpha((size_t)2 * N), w((size_t)2 * N);
  bssm_filter_out out;
  std::memset(&out, 0, sizeof out);
  out.alpha = alpha.data(); out.weights = w.data();
  struct Restore { ~Restore() { g_milstein_tend = 1.0; } } restore;
  g_milstein_tend = t_end;
  const int rc = run_functor(&one, N, seed, false, &out);
  if (rc < 0) return rc;
  *x_out = alpha[0];
  return BSSM_OK;
}
int bssm_bsf_smoother_sde(const bssm_fn_model* model, uint32_t nsim, uint32_t seed, bssm_filter_out* out) {
  if (model && !bssm::fn_is_sde(model->model_id)) return fail(BSSM_ERR_INVALID, "not an ssm_sde model id");
  return run_functor(model, nsim, seed, true, out);
}
int bssm_psi_smoother_nlg(const bssm_fn_model* model, uint32_t nsim, uint32_t seed, int32_t max_iter, double conv_tol,
                          bssm_filter_out* out) {
  if (!out || nsim < 1) return fail(BSSM_ERR_INVALID, "bad arguments");
  std::vector<bssm::FilterTables> tabs(1);
  std::vector<double> fpar, amats;
  double al = 0; bool ok = true; int m = 1;
  const int rc = functor_psi_tables(model, nullptr, max_iter, conv_tol > 0 ? conv_tol : 1e-8, &tabs[0], &fpar, &amats, &al, &ok, &m);
  if (rc) return rc;
  if (!ok) { out->logLik = -INFINITY; return BSSM_WARN_DEGENERATE; }  // src/model_ssm_nlg.cpp:988-990
  if (!have_device()) return no_device();
  fpar.insert(fpar.end(), amats.begin(), amats.end());  // run_tables splits them again
  return run_tables(tabs, &fpar, model->n, m, m, 1, (int)nsim, 1, true, seed, al, out);
}

int bssm_fn_psi_setup(const bssm_fn_model* model, const double* is_mode, int32_t max_iter, double conv_tol, double* mode,
                      double* scales, double* approx_loglik, double* alphahat, double* Vt, double* Ct, double* Zt, double* Tt,
                      double* Dt, double* Cc) {
  bssm::FnModel f;
  const int rc = to_fn_model(model, &f, 16);
  if (rc) return rc;
  bssm::FnSetup su;
  const bool loglik_only = !alphahat && !Vt && !Ct;  // (what the approximate MCMC chain asks for every iteration)
  const bool ok = bssm::fn_psi_setup(f, is_mode, max_iter, conv_tol > 0 ? conv_tol : 1e-8, &su, loglik_only);
  auto cp = [](double* dst, const std::vector<double>& v) { if (dst) std::memcpy(dst, v.data(), v.size() * sizeof(double)); };
  cp(mode, su.mode); cp(scales, su.scales); cp(alphahat, su.alphahat); cp(Vt, su.Vt); cp(Ct, su.Ct);
  cp(Zt, su.Z); cp(Tt, su.T); cp(Dt, su.D); cp(Cc, su.C);
  if (approx_loglik) *approx_loglik = su.approx_loglik;
  return ok ? BSSM_OK : BSSM_ERR_NUMERIC;
}

int bssm_fn_loglik_batch(const bssm_fn_model* model, uint32_t nsim, uint32_t sampling_method, uint32_t seed, uint32_t stream0,
                         uint32_t n_rep, double* loglik) {
  return bssm_fn_loglik_batch_ex(model, nsim, sampling_method, seed, stream0, n_rep, 100, 1e-8, loglik, nullptr);
}

// One single-filter smoother call of a functor model on Philox stream `stream0` (the chains key their filters by
// (seed, 2 * iteration [+ 1]) and must be able to re-create the accepted particle system)
int bssm_fn_smoother_stream(const bssm_fn_model* model, uint32_t sampling_method, uint32_t nsim, uint32_t seed,
                            uint32_t stream0, int32_t max_iter, double conv_tol, bssm_filter_out* out) {
  if (!model) return fail(BSSM_ERR_INVALID, "model is NULL");
  g_call_stream0 = stream0;
  int rc;
  if (sampling_method == BSSM_PSI) rc = bssm_psi_smoother_nlg(model, nsim, seed, max_iter, conv_tol, out);
  else if (sampling_method == BSSM_EKPF) rc = bssm_ekpf_smoother_nlg(model, nsim, seed, out);
  else if (sampling_method != BSSM_BSF) rc = fail(BSSM_ERR_UNSUPPORTED, "sampling_method must be 1 (psi), 2 (bsf) or 4 (ekf particle filter)");
  else if (bssm::fn_is_sde(model->model_id)) rc = bssm_bsf_smoother_sde(model, nsim, seed, out);
  else rc = bssm_bsf_smoother_nlg(model, nsim, seed, out);
  g_call_stream0 = 0;
  return rc;
}

int bssm_fn_loglik_batch_ex(const bssm_fn_model* model, uint32_t nsim, uint32_t sampling_method, uint32_t seed, uint32_t stream0,
                            uint32_t n_rep, int32_t max_iter, double conv_tol, double* loglik, double* approx_loglik) {
  if (!loglik || n_rep < 1 || nsim < 1) return fail(BSSM_ERR_INVALID, "bad arguments");
  if (approx_loglik) *approx_loglik = 0.0;
  if (sampling_method != BSSM_PSI && sampling_method != BSSM_BSF && sampling_method != BSSM_EKPF)
    return fail(BSSM_ERR_UNSUPPORTED, "sampling_method must be 1 (psi), 2 (bsf) or 4 (ekf particle filter)");
  std::vector<bssm::FilterTables> tabs(1);
  std::vector<double> fpar, amats;
  int m = 1, k = 1, rc;
  double base = 0.0;
  const bool psi = sampling_method == BSSM_PSI, ekpf = sampling_method == BSSM_EKPF;
  if (psi) {
    bool ok = true;
    if ((rc = functor_psi_tables(model, nullptr, max_iter, conv_tol > 0 ? conv_tol : 1e-8, &tabs[0], &fpar, &amats, &base, &ok, &m))) return rc;
    k = m;
    if (approx_loglik) *approx_loglik = ok ? base : -INFINITY;  // of the very set-up the particles are drawn from
    if (!ok) { for (uint32_t r = 0; r < n_rep; r++) loglik[r] = -INFINITY; return BSSM_WARN_DEGENERATE; }
  } else if (ekpf) {
    if ((rc = functor_ekpf_tables(model, &tabs[0], &fpar, &m))) return rc;
    k = m;
    amats.assign(fpar.begin() + bssm::FP_SIZE, fpar.end());
    fpar.resize(bssm::FP_SIZE);
  } else {
    if ((rc = functor_tables(model, &tabs[0], &fpar, &m, &k))) return rc;
  }
  if (!have_device()) return no_device();
  std::vector<bssm::FilterTables> all(n_rep, tabs[0]);
  DeviceBatch db;
  if ((rc = db.upload(all, model->n, m, k, 1, (int)nsim, psi ? 1 : (ekpf ? 2 : 0), false, 0))) return rc;
  if ((rc = db.set_fpar(fpar))) return rc;
  if ((psi || ekpf) && (rc = db.set_amats(amats))) return rc;
  db.const_terms.assign(n_rep, base);
  if ((rc = db.run(seed, stream0, nullptr))) return rc;
  if ((rc = db.fetch_loglik(nullptr, loglik))) return rc;
  bool deg = false;
  for (uint32_t r = 0; r < n_rep; r++) if (!std::isfinite(loglik[r])) { loglik[r] = -INFINITY; deg = true; }
  return deg ? BSSM_WARN_DEGENERATE : BSSM_OK;
}

// replaces _bssm_gaussian_psi_smoother (src/R_psi.cpp:14-50): ssm_mlg (model_type 0) and the ssm_ulg family (1..3)
int bssm_gaussian_psi_smoother(const bssm_ung_model* model, uint32_t nsim, uint32_t seed, int model_type, double* alpha) {
  if (!alpha || nsim < 1) return fail(BSSM_ERR_INVALID, "bad arguments");
  if (model_type < 0 || model_type > 3) return fail(BSSM_ERR_UNSUPPORTED, "model_type not handled by this entry point");
  bssm::PsiSetup su;
  int n = 0, m = 0;
  if (model_type == 0) {  // ssm_mlg (src/R_psi.cpp:20-25): the multivariate smoother on the host, the draws on the device as below
    const std::string err = bssm::mlg_smoothing_setup(model, &su);
    if (!err.empty()) return fail(err.find("non-finite") != std::string::npos ? BSSM_ERR_NUMERIC : BSSM_ERR_INVALID, err);
    n = model->n; m = model->m;
  } else {
    bssm::HostModel hm;
    const std::string err = bssm::ingest_model(model, false, &hm);
    if (!err.empty()) return fail(BSSM_ERR_INVALID, err);
    if (!hm.gaussian()) return fail(BSSM_ERR_INVALID, "gaussian_psi_smoother needs a linear-Gaussian model");
    if (!bssm::gaussian_smoothing_setup(hm, &su)) return fail(BSSM_ERR_NUMERIC, "smoother produced non-finite values");
    n = hm.n; m = hm.m;
  }
  if (!have_device()) return no_device();
  const int N = (int)nsim, MS = 2 * m * m + m;
  std::vector<double> mats((size_t)n * MS), init((size_t)m + m * m);
  for (int t = 0; t < n; t++) {
    double* mt = mats.data() + (size_t)t * MS;
    const double* Ct1 = su.Ct.data() + (size_t)m * m * (t + 1); const double* Vt1 = su.Vt.data() + (size_t)m * m * (t + 1);
    for (int q = 0; q < m * m; q++) { mt[q] = Ct1[q]; mt[m * m + q] = Vt1[q]; }
    for (int j = 0; j < m; j++) { double s = 0; for (int c = 0; c < m; c++) s += Ct1[j + m * c] * su.alphahat[c + (size_t)m * t]; mt[2 * m * m + j] = su.alphahat[j + (size_t)m * (t + 1)] - s; }
  }
  for (int j = 0; j < m; j++) init[j] = su.alphahat[j];
  for (int q = 0; q < m * m; q++) init[m + q] = su.Vt[q];
  DevBuf dm, di, hist, cube;
  HIP_TRY(dm.alloc(mats.size() * sizeof(double))); HIP_TRY(di.alloc(init.size() * sizeof(double)));
  HIP_TRY(hist.alloc((size_t)(n + 1) * m * N * sizeof(double))); HIP_TRY(cube.alloc(hist.bytes));
  HIP_TRY(hipMemcpy(dm.p, mats.data(), dm.bytes, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(di.p, init.data(), di.bytes, hipMemcpyHostToDevice));
  HIP_TRY(bssm::launch_sim_smoother(dm.as<double>(), di.as<double>(), hist.as<double>(), n, m, N, seed, nullptr));
  HIP_TRY(bssm::launch_to_cube(hist.as<double>(), cube.as<double>(), 1, n + 1, m, N, nullptr));
  HIP_TRY(hipMemcpy(alpha, cube.p, cube.bytes, hipMemcpyDeviceToHost));
  return BSSM_OK;
}

int bssm_psi_setup(const bssm_ung_model* model, const double* is_mode, double* approx_y, double* approx_H, double* mode,
                   double* scales, double* approx_loglik, double* alphahat, double* Vt, double* Ct) {
  bssm::HostModel hm;
  const std::string err = bssm::ingest_model(model, is_mode == nullptr, &hm);
  if (!err.empty()) return fail(BSSM_ERR_INVALID, err);
  bssm::PsiSetup su;
  const bool ok = bssm::psi_setup(hm, is_mode, &su);
  auto cp = [](double* dst, const std::vector<double>& v) { if (dst) std::memcpy(dst, v.data(), v.size() * sizeof(double)); };
  cp(approx_y, su.approx_y); cp(approx_H, su.approx_H); cp(mode, su.mode); cp(scales, su.scales);
  cp(alphahat, su.alphahat); cp(Vt, su.Vt); cp(Ct, su.Ct);
  if (approx_loglik) *approx_loglik = su.approx_loglik;
  return ok ? BSSM_OK : BSSM_ERR_NUMERIC;
}

int bssm_gaussian_smoother(const bssm_ung_model* model, double* alphahat, double* Vt) {
  if (!model || !alphahat || !Vt) return fail(BSSM_ERR_INVALID, "NULL argument");
  bssm::HostModel hm;
  const std::string err = bssm::ingest_model(model, false, &hm);
  if (!err.empty()) return fail(BSSM_ERR_INVALID, err);
  if (!hm.gaussian()) return fail(BSSM_ERR_UNSUPPORTED, "bssm_gaussian_smoother: linear-Gaussian (ssm_ulg family) models only");
  std::vector<double> a, v;
  if (!bssm::gaussian_smoother(hm, &a, &v)) return fail(BSSM_ERR_NUMERIC, "the Kalman smoother produced non-finite values");
  std::memcpy(alphahat, a.data(), a.size() * sizeof(double));
  std::memcpy(Vt, v.data(), v.size() * sizeof(double));
  return BSSM_OK;
}

int bssm_nongaussian_loglik_batch(const bssm_ung_model* model, uint32_t nsim, uint32_t sampling_method, uint32_t seed,
                                  uint32_t stream0, uint32_t n_rep, double* loglik) {
  if (!loglik || n_rep < 1) return fail(BSSM_ERR_INVALID, "loglik NULL or n_rep < 1");
  if (sampling_method != BSSM_PSI && sampling_method != BSSM_BSF)
    return fail(BSSM_ERR_UNSUPPORTED, "sampling_method must be 1 (psi) or 2 (bsf); SPDK is not a particle filter path");
  const bool psi = sampling_method == BSSM_PSI;
  bssm::HostModel hm;
  const std::string err = bssm::ingest_model(model, psi || nsim == 0, &hm);
  if (!err.empty()) return fail(BSSM_ERR_INVALID, err);
  if (nsim == 0) {  // src/model_ssm_ung.cpp:214-234
    const double a = bssm::approx_loglik_only(hm);
    for (uint32_t r = 0; r < n_rep; r++) loglik[r] = a;
    return BSSM_OK;
  }
  if (!have_device()) return no_device();
  std::vector<bssm::FilterTables> tabs(1);
  double base = 0;
  const int MP = bssm::pad_dim(hm.m);
  if (psi) {
    bssm::PsiSetup su;
    if (!bssm::psi_setup(hm, nullptr, &su)) { for (uint32_t r = 0; r < n_rep; r++) loglik[r] = -INFINITY; return BSSM_WARN_DEGENERATE; }
    bssm::build_psi_tables(hm, su, MP, &tabs[0]);
    base = su.approx_loglik;
  } else {
    bssm::build_bsf_tables(hm, MP, bssm::pad_noise_dim(MP, hm.k), &tabs[0]);
    base = bssm::const_term(hm);
  }
  // replicas share one table: stride 0 between filters
  DeviceBatch db;
  int rc = db.upload(tabs, hm.n, hm.m, hm.k, hm.p, (int)nsim, psi, false, 0);
  if (rc) return rc;
  db.fb.F = (int)n_rep;
  db.fb.mats_fstride = 0; db.fb.obs_fstride = 0; db.fb.init_fstride = 0;
  HIP_TRY(db.loglik.alloc(n_rep * sizeof(double)));
  db.fb.loglik = db.loglik.as<double>();
  HIP_TRY(db.dead.alloc(n_rep * sizeof(int)));
  db.fb.dead_at = db.dead.as<int>();
  const int cap = bssm::wg_filter_capacity(MP);
  db.use_grid = !((int)nsim <= cap && (n_rep >= 32 || nsim <= 2048));
  db.use_grid = bssm::prefer_step_shape(db.fb, db.use_grid);
  if (bssm::opt().force_path == 2) db.use_grid = true;
  if (bssm::opt().force_path == 1 && (int)nsim <= cap) db.use_grid = false;
  if (bssm::opt().force_path == 3 && bssm::row_filter_supported(db.fb)) db.use_grid = true;
  if (db.use_grid && (rc = db.alloc_step_workspace())) return rc;
  db.const_terms.assign(n_rep, base);
  if ((rc = db.run(seed, stream0, nullptr))) return rc;
  if ((rc = db.fetch_loglik(nullptr, loglik))) return rc;
  bool deg = false;
  for (uint32_t r = 0; r < n_rep; r++) if (!std::isfinite(loglik[r])) { loglik[r] = -INFINITY; deg = true; }
  return deg ? BSSM_WARN_DEGENERATE : BSSM_OK;
}

int bssm_nongaussian_loglik(const bssm_ung_model* model, uint32_t nsim, uint32_t sampling_method, uint32_t seed,
                            int model_type, double* loglik) {
  if (!is_gaussian_type_ok(0, model_type)) return fail(BSSM_ERR_UNSUPPORTED, "model_type not handled by this entry point");
  return bssm_nongaussian_loglik_batch(model, nsim, sampling_method, seed, 0, 1, loglik);
}

int bssm_update_model_builtin(int model_type, const int32_t* flags, const double* theta, int32_t n_par,
                              bssm_ung_model* model, double* T, double* R, double* P1, double* a1, double* C,
                              double* beta) {
  if (!flags || !theta || !model) return fail(BSSM_ERR_INVALID, "NULL argument");
  const std::string err = bssm::update_model_builtin(model_type, flags, theta, n_par, model, T, R, P1, a1, C, beta);
  if (!err.empty()) return fail(BSSM_ERR_UNSUPPORTED, err);
  return BSSM_OK;
}

int bssm_update_model_builtin_batch(int model_type, const int32_t* flags, const double* theta, int32_t n_par, int32_t count,
                                    const bssm_ung_model* model, double* T, double* R, double* P1, double* a1, double* C,
                                    double* beta, double* phi) {
  if (!flags || !theta || !model || !T || !R || !P1 || !a1 || !C || !phi || count < 0 || n_par < 1) return fail(BSSM_ERR_INVALID, "NULL argument");
  if (!model->T || !model->R || !model->P1 || !model->a1 || !model->C) return fail(BSSM_ERR_INVALID, "the model's T, R, P1, a1, C must be set");
  const size_t n = model->n, m = model->m, k = model->k;
  const size_t sT = m * m * (model->Ttv ? n : 1), sR = m * k * (model->Rtv ? n : 1), sP = m * m, sa = m, sC = m * (model->Ctv ? n : 1),
               sb = model->n_xreg > 0 ? (size_t)model->n_xreg : 0;
  if (sb && (!beta || !model->beta)) return fail(BSSM_ERR_INVALID, "beta required with regressors");
  for (int32_t i = 0; i < count; i++) {
    double *Ti = T + sT * i, *Ri = R + sR * i, *Pi = P1 + sP * i, *ai = a1 + sa * i, *Ci = C + sC * i, *bi = sb ? beta + sb * i : nullptr;
    std::memcpy(Ti, model->T, sT * sizeof(double)); std::memcpy(Ri, model->R, sR * sizeof(double));
    std::memcpy(Pi, model->P1, sP * sizeof(double)); std::memcpy(ai, model->a1, sa * sizeof(double));
    std::memcpy(Ci, model->C, sC * sizeof(double));
    if (sb) std::memcpy(bi, model->beta, sb * sizeof(double));
    bssm_ung_model mi = *model;
    const std::string err = bssm::update_model_builtin(model_type, flags, theta + (size_t)n_par * i, n_par, &mi, Ti, Ri, Pi, ai, Ci, bi);
    if (!err.empty()) return fail(BSSM_ERR_UNSUPPORTED, "draw " + std::to_string(i) + ": " + err);
    phi[i] = mi.phi;
  }
  return BSSM_OK;
}

// ------------------------------------------------------------------ user-defined models (run-time compiled)
}  // extern "C"
namespace {
std::string library_dir() {
  Dl_info info;
  if (dladdr((const void*)&bssm_amd_version, &info) && info.dli_fname) {
    std::string p(info.dli_fname);
    const size_t k = p.rfind('/');
    return k == std::string::npos ? std::string(".") : p.substr(0, k);
  }
  return ".";
}
uint64_t fnv1a(const std::string& s) {
  uint64_t h = 1469598103934665603ull;
  for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
  return h;
}
std::mutex g_user_mu;
std::map<std::string, int> g_user_loaded;  // library path -> registry slot
}  // namespace
extern "C" {

int bssm_user_model_build(const char* source, int32_t n_states, char* lib_path, int32_t lib_path_capacity) {
  if (!source || !lib_path || lib_path_capacity < 16) return fail(BSSM_ERR_INVALID, "NULL argument");
  if (n_states < 1 || n_states > 16) return fail(BSSM_ERR_UNSUPPORTED, "user-defined models: n_states must be 1 ... 16");
  const std::string dir = library_dir();
  const std::string& env_cache = bssm::opt().user_cache;
  const bool env = !env_cache.empty();
  std::string cache = env ? env_cache : dir + "/user_models";
  if (!env && access(dir.c_str(), W_OK) != 0) {  // the package directory is read-only (a system-wide install): a per-user cache
    // never a shared directory such as /tmp: the library name is predictable and a cached library is dlopen'ed
    const char* home = std::getenv("HOME");
    if (!home || !*home) return fail(BSSM_ERR_INVALID, "user-defined models: the package directory is read-only and HOME is unset; set BSSM_AMD_USER_CACHE to a private directory");
    cache = std::string(home) + "/.cache/bssm_amd/user_models";
  }
  // the cache key covers everything that is compiled into the library: the kernel headers, the state dimension, the snippet
  std::string key = std::string(bssm_amd_version()) + "|" + std::to_string(n_states) + "|" + source;
  for (const char* hdr : {"pf_filter_impl.h", "pf_grid_v2.h", "pf_row_step.h", "pf_canon.h", "pf_functors.h", "pf_device.h", "pf_kernels.h", "pf_math.h", "pf_math_tables.h", "options.h", "tls_stream.h", "user_model_tu.hip"}) {
    FILE* hf = std::fopen((dir + "/csrc/" + hdr).c_str(), "rb");
    if (!hf) return fail(BSSM_ERR_INVALID, "kernel sources not found next to the library: " + dir + "/csrc/" + hdr);
    char buf[1 << 16];
    size_t got;
    while ((got = std::fread(buf, 1, sizeof buf, hf)) > 0) key.append(buf, got);
    std::fclose(hf);
  }
  char hbuf[32];
  std::snprintf(hbuf, sizeof hbuf, "%016llx", (unsigned long long)fnv1a(key));
  const std::string stem = cache + "/um_" + hbuf, snippet = stem + ".h", so = cache + "/libbssm_user_" + hbuf + ".so", log = stem + ".log";
  if ((int)so.size() + 1 > lib_path_capacity) return fail(BSSM_ERR_INVALID, "lib_path buffer too small");
  // the paths go through a shell command line inside single quotes: refuse anything that could end the quoting
  // (and the snippet path sits inside a double-quoted -D value: no double quote or backslash either)
  for (const std::string* pth : {&dir, (const std::string*)&cache})
    if (pth->find_first_of("'\"\\\n") != std::string::npos)
      return fail(BSSM_ERR_INVALID, "unsupported character in the library / cache path: " + *pth);
  std::lock_guard<std::mutex> lk(g_user_mu);
  // a cached library is loaded into the caller's process: trust it only when the cache directory and the file belong to
  // this user (or root) and nobody else can write to them
  auto private_to_us = [](const std::string& path, bool is_dir) -> bool {
    struct stat sb;
    if (stat(path.c_str(), &sb) != 0) return false;
    if (is_dir ? !S_ISDIR(sb.st_mode) : !S_ISREG(sb.st_mode)) return false;
    if (sb.st_uid != geteuid() && sb.st_uid != 0) return false;
    return (sb.st_mode & (S_IWGRP | S_IWOTH)) == 0;
  };
  // a directory of OURS with loose permissions is tightened -- only the library's own default cache (created by the library, or
  // left behind by an older version of it): a caller-supplied BSSM_AMD_USER_CACHE that others can write to is refused, never
  // chmod-ed (it may be a shared directory such as /tmp or /scratch, and this process may be root)
  auto make_private_dir = [&](const std::string& path) -> bool {
    if (env) return false;
    struct stat sb;
    if (stat(path.c_str(), &sb) != 0 || !S_ISDIR(sb.st_mode) || sb.st_uid != geteuid()) return false;
    return chmod(path.c_str(), sb.st_mode & 07777 & ~(S_IWGRP | S_IWOTH)) == 0;
  };
  struct stat sb_so;
  if (stat(so.c_str(), &sb_so) == 0 && (private_to_us(cache, true) || make_private_dir(cache)) && private_to_us(so, false)) {
    // already built: the hash covers the source, n_states and the version.  (A cached library that fails the ownership /
    // permission test is not loaded: it is rebuilt below and replaced by rename.)
    std::strcpy(lib_path, so.c_str());
    return BSSM_OK;
  }
  if (std::system(("mkdir -p -m 0700 '" + cache + "'").c_str()) != 0) return fail(BSSM_ERR_INVALID, "cannot create " + cache);
  if (!private_to_us(cache, true) && !make_private_dir(cache))
    return fail(BSSM_ERR_INVALID, "the user-model cache " + cache + " belongs to another user or is writable by others (set BSSM_AMD_USER_CACHE to a private directory)");
  // several processes (ranks) may build the same model at once: every file of a build carries the pid until it is complete,
  // so no process ever reads a file another one is writing; the finished library appears by one rename
  const std::string pid = std::to_string((long)getpid());
  const std::string snippet_final = snippet, log_final = log;
  const std::string snippet_tmp = stem + "." + pid + ".h", log_tmp = stem + "." + pid + ".log";
  FILE* f = std::fopen(snippet_tmp.c_str(), "w");
  if (!f) return fail(BSSM_ERR_INVALID, "cannot write " + snippet_tmp);
  std::fputs(source, f);
  std::fclose(f);
  const std::string tmp = so + ".tmp." + pid;
  const char* hipcc_env = std::getenv("HIPCC");
  const std::string hipcc = hipcc_env ? hipcc_env : "/opt/rocm/bin/hipcc";
  // six parts side by side (sampler x kernel family -- user_model_tu.hip), then one link: ~8 s of wall clock on six cores
  // instead of ~27 s for the single unit
  const std::string common = hipcc + " -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I'" + dir + "/csrc' -DBSSM_USER_M=" + std::to_string(bssm::pad_dim(n_states)) +
                             " -DBSSM_USER_NSTATES=" + std::to_string(n_states) + " -DBSSM_USER_SNIPPET='\"" + snippet_tmp + "\"' ";
  // (at most `jobs` compilers at a time: the host's hardware threads divided by the ranks of the node -- LOCAL_WORLD_SIZE of the
  // launcher -- so that eight ranks building a model at once do not start 48 compilers; BSSM_AMD_USER_BUILD_JOBS overrides)
  int jobs = 6;
  {
    const unsigned hw = std::thread::hardware_concurrency();
    const char* lws = std::getenv("LOCAL_WORLD_SIZE");
    const int ranks = lws && std::atoi(lws) > 0 ? std::atoi(lws) : 1;
    if (hw > 0) jobs = std::max(1, std::min(6, (int)hw / ranks));
    if (const char* ej = std::getenv("BSSM_AMD_USER_BUILD_JOBS")) jobs = std::max(1, std::min(6, std::atoi(ej)));
  }
  // More than 8 states: no EKF particle filter (its per-particle covariance is a register array) -- parts 2 and 5 are left out; the psi-APF
  // (parts 1 and 4, round 6) for snippets with Jacobians and a CONSTANT R (a state-dependent R_fn needs two general m x m densities per particle:
  // 8 states at most); everything else at 9 ... 16 states is the bootstrap sampler alone (parts 0 and 3)
  const std::string src_s(source);
  const bool wide = n_states > 8;
  const bool wide_psi = wide && src_s.find("BSSM_USER_JACOBIANS") != std::string::npos && src_s.find("BSSM_USER_RFN") == std::string::npos && src_s.find("BSSM_USER_NLG") != std::string::npos;
  const bool bsf_only = wide && !wide_psi;
  std::string cmd = "( ", objs, logs;
  for (int part = 0; part < 6; part++) {
    if (bsf_only && part % 3 != 0) continue;
    if (wide_psi && part % 3 == 2) continue;
    const std::string obj = tmp + ".p" + std::to_string(part) + ".o", plog = log_tmp + ".p" + std::to_string(part);
    cmd += common + (bsf_only ? "-DBSSM_USER_BSF_ONLY=1 " : (wide_psi ? "-DBSSM_USER_NO_EKF=1 " : "")) + "-DBSSM_USER_PART=" + std::to_string(part) + " -c '" + dir + "/csrc/user_model_tu.hip' -o '" + obj + "' > '" + plog + "' 2>&1 & ";
    if ((part + 1) % jobs == 0 && part + 1 < 6) cmd += "wait ; ";
    objs += " '" + obj + "'";
    logs += " '" + plog + "'";
  }
  cmd += "wait ) ; cat" + logs + " > '" + log_tmp + "' 2>/dev/null ; rm -f" + logs + " ; ( for o in" + objs + " ; do test -s \"$o\" || exit 1 ; done ) && " + hipcc +
         " --offload-arch=gfx950 -shared -fPIC" + objs + " -o '" + tmp + "' >> '" + log_tmp + "' 2>&1 && mv '" + tmp + "' '" + so + "' ; rc=$? ; rm -f" + objs + " ; exit $rc";
  const int build_rc = std::system(cmd.c_str());
  (void)std::rename(snippet_tmp.c_str(), snippet_final.c_str());  // kept next to the library for reference
  (void)std::rename(log_tmp.c_str(), log_final.c_str());
  if (build_rc != 0) {
    std::string tail;
    if (FILE* lf = std::fopen(log.c_str(), "r")) {
      char buf[4096];
      const size_t got = std::fread(buf, 1, sizeof buf - 1, lf);
      buf[got] = 0;
      tail = buf;
      std::fclose(lf);
    }
    return fail(BSSM_ERR_INVALID, "compiling the user model failed (" + log + "):\n" + tail);
  }
  std::strcpy(lib_path, so.c_str());
  return BSSM_OK;
}

int bssm_user_model_load(const char* lib_path) {
  if (!lib_path) return fail(BSSM_ERR_INVALID, "NULL argument");
  std::lock_guard<std::mutex> lk(g_user_mu);
  auto it = g_user_loaded.find(lib_path);
  if (it != g_user_loaded.end()) return it->second;
  void* h = dlopen(lib_path, RTLD_NOW | RTLD_LOCAL);
  if (!h) return fail(BSSM_ERR_INVALID, std::string("dlopen failed: ") + dlerror());
  auto wg = (bssm::user_wg_launcher_t)dlsym(h, "bssm_user_launch_wg");
  auto grid = (bssm::user_grid_launcher_t)dlsym(h, "bssm_user_launch_grid");
  if (!wg || !grid) return fail(BSSM_ERR_INVALID, "the library does not export the user-model launchers");
  if (auto set_opt = (void (*)(const bssm::Options& (*)()))dlsym(h, "bssm_user_set_options")) set_opt(&bssm::opt);  // the library reads this library's switches
  const int slot = bssm::register_user_launchers(wg, grid);
  if (slot < 1) return fail(BSSM_ERR_UNSUPPORTED, "too many user models loaded");
  g_user_host[slot].T = (decltype(bssm::UserHostFns::T))dlsym(h, "bssm_user_host_T");
  g_user_host[slot].Z = (decltype(bssm::UserHostFns::Z))dlsym(h, "bssm_user_host_Z");
  g_user_host[slot].Tg = (decltype(bssm::UserHostFns::Tg))dlsym(h, "bssm_user_host_Tg");
  g_user_host[slot].Zg = (decltype(bssm::UserHostFns::Zg))dlsym(h, "bssm_user_host_Zg");
  if (auto has_tv = (int (*)())dlsym(h, "bssm_user_host_has_tv")) g_user_host[slot].has_tv = has_tv();
  g_user_host[slot].H = (decltype(bssm::UserHostFns::H))dlsym(h, "bssm_user_host_H");
  g_user_host[slot].R = (decltype(bssm::UserHostFns::R))dlsym(h, "bssm_user_host_R");
  g_user_loaded[lib_path] = slot;
  return slot;
}

// ------------------------------------------------------------------ user-defined models compiled IN PROCESS (hiprtc)
// The reference compiles a user's model inside the R session (Rcpp::sourceCpp, R/cpp_example_models.R:762) and calls it through
// pointers (src/model_ssm_nlg.h:10-21).  bssm_user_model_build above needs the ROCm compiler driver on the machine and builds every
// kernel variant of every sampler (10-18 s).  A model that is only ever run through the BOOTSTRAP filter needs neither: its device
// functions are compiled by hiprtc (libhiprtc: the compiler as a library, no host compiler, no process) into exactly the kernel
// variant a call asks for, the first time it asks -- ~1-2 s per variant, code objects cached on disk by content hash next to the
// hipcc-built libraries.  What such a model cannot do: the psi-APF and the EKF particle filter, whose set-up runs the snippet's functions
// ON THE HOST (mode search, linearisation) -- those entry points refuse a JIT slot with a message that names bssm_user_model_build.
// Kernel shapes: workgroup-resident (512-thread form) and the row family (up to 16 384 particles); both evaluate pf_canon.h, so a model
// gives the same bits whether it was built by hipcc or compiled here.
}  // extern "C"
#include <hip/hiprtc.h>
#include <utility>
#include "pf_canon.h"
namespace {
constexpr int kJitWgThreads = 512;  // = WG_NT_MAX of pf_filter_impl.h: the 512-thread form of the workgroup-resident kernel
struct JitModel {
  std::string source;
  int n_states = 0, MP = 0, fn_id = 0;
  std::mutex mu;
  std::map<std::string, hipFunction_t> kernels;  // "<device>|<name expression>" -> loaded function
};
JitModel* g_jit[64] = {};

std::string jit_cache_dir() {
  const std::string& env_cache = bssm::opt().user_cache;
  if (!env_cache.empty()) return env_cache;
  const std::string dir = library_dir();
  if (access(dir.c_str(), W_OK) == 0) return dir + "/user_models";
  const char* home = std::getenv("HOME");
  return (home && *home) ? std::string(home) + "/.cache/bssm_amd/user_models" : std::string();
}
bool jit_private_file(const std::string& path) {  // ours (or root's), not writable by anybody else: only such a code object is loaded
  struct stat sb;
  if (stat(path.c_str(), &sb) != 0 || !S_ISREG(sb.st_mode)) return false;
  if (sb.st_uid != geteuid() && sb.st_uid != 0) return false;
  return (sb.st_mode & (S_IWGRP | S_IWOTH)) == 0;
}
bool jit_private_dir(const std::string& path) {
  struct stat sb;
  if (stat(path.c_str(), &sb) != 0 || !S_ISDIR(sb.st_mode)) return false;
  if (sb.st_uid != geteuid() && sb.st_uid != 0) return false;
  return (sb.st_mode & (S_IWGRP | S_IWOTH)) == 0;
}

// mkdir -p with mode 0700, by mkdir(2): nothing of the path reaches a shell
void make_private_dirs(const std::string& path) {
  for (size_t at = 1; at <= path.size(); at++)
    if (at == path.size() || path[at] == '/') (void)mkdir(path.substr(0, at).c_str(), 0700);
}

// Round 6: the host side of the psi-APF / EKF particle filter (mode search, linearisation: host_model.cpp fn_psi_setup, fn_ekpf_init) calls the
// model's T / Z / T' / Z' (/ H / R) at one state at a time.  A model compiled in process has no host code, so those calls run THIS kernel --
// one thread, the snippet's own __device__ functions, operands and results in a pinned block both sides see -- ~15 us per call
// (a psi set-up of 100 time points: a few thousand calls; models on a hot path build with hipcc, whose library carries host entry points).
const char* const kJitEvalKernel = R"JIT(
struct bssm_jit_eval_block { double alpha[16], theta[64], known[64], out[256]; };
extern "C" __global__ void bssm_user_eval(int what, int t, const double* ktv, int rows, int cols, bssm_jit_eval_block* blk) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  (void)what; (void)t; (void)ktv; (void)rows; (void)cols;
#ifdef BSSM_USER_NLG
#ifdef BSSM_USER_TV
#define BSSM_JTV bssm_user_tv{t, ktv, rows, cols},
#else
#define BSSM_JTV
#endif
  switch (what) {
    case 0: bssm_user_T(BSSM_JTV blk->alpha, blk->theta, blk->known, blk->out); break;
    case 1: blk->out[0] = bssm_user_Z(BSSM_JTV blk->alpha, blk->theta, blk->known); break;
#ifdef BSSM_USER_JACOBIANS
    case 2: bssm_user_Tg(BSSM_JTV blk->alpha, blk->theta, blk->known, blk->out); break;
    case 3: bssm_user_Zg(BSSM_JTV blk->alpha, blk->theta, blk->known, blk->out); break;
#endif
#ifdef BSSM_USER_HFN
    case 4: blk->out[0] = bssm_user_H(BSSM_JTV blk->alpha, blk->theta, blk->known); break;
#endif
#ifdef BSSM_USER_RFN
    case 5: bssm_user_R(BSSM_JTV blk->alpha, blk->theta, blk->known, blk->out); break;
#endif
    default: break;
  }
#endif
}
)JIT";

// the translation unit hiprtc sees: the snippet, then the kernel templates (device view of the headers: __HIPCC_RTC__)
std::string jit_translation_unit(const JitModel& jm) {
  return "#define BSSM_USER_MODEL 1\n#define BSSM_FN __device__ inline\n"
         "struct bssm_user_tv { int t; const double* known_tv; int rows, cols; };\n" + jm.source +
         "\n#define BSSM_USER_M " + std::to_string(jm.MP) + "\n#define BSSM_USER_NSTATES " + std::to_string(jm.n_states) +
         "\n#define BSSM_FN_PSI_MASK 7\n#include \"pf_filter_impl.h\"\n" + kJitEvalKernel;
}

// compile (or load from the disk cache) ONE kernel variant for the current device; nullptr + *err on failure
hipFunction_t jit_kernel(JitModel& jm, const std::string& name, std::string* err) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const std::string key = std::to_string(dev) + "|" + name;
  std::lock_guard<std::mutex> lk(jm.mu);
  auto it = jm.kernels.find(key);
  if (it != jm.kernels.end()) return it->second;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { *err = "no device properties"; return nullptr; }
  std::string arch = prop.gcnArchName;
  if (arch.find(':') != std::string::npos) arch = arch.substr(0, arch.find(':'));
  const std::string dir = library_dir(), tu = jit_translation_unit(jm);
  // cache key: everything that is compiled into the code object
  std::string hk = std::string(bssm_amd_version()) + "|" + arch + "|" + name + "|" + tu;
  for (const char* hdr : {"pf_filter_impl.h", "pf_grid_v2.h", "pf_row_step.h", "pf_canon.h", "pf_functors.h", "pf_device.h", "pf_kernels.h", "pf_math.h", "pf_math_tables.h"}) {
    FILE* hf = std::fopen((dir + "/csrc/" + hdr).c_str(), "rb");
    if (!hf) { *err = "kernel sources not found next to the library: " + dir + "/csrc/" + hdr; return nullptr; }
    char buf[1 << 16];
    size_t got;
    while ((got = std::fread(buf, 1, sizeof buf, hf)) > 0) hk.append(buf, got);
    std::fclose(hf);
  }
  char hbuf[32];
  std::snprintf(hbuf, sizeof hbuf, "%016llx", (unsigned long long)fnv1a(hk));
  const std::string cache = jit_cache_dir();
  const std::string co_path = cache.empty() ? std::string() : cache + "/jit_" + hbuf + ".hsaco", ln_path = co_path + ".name";
  std::vector<char> code;
  std::string lowered;
  bool from_cache = false;
  for (int attempt = 0; attempt < 2; attempt++) {   // (a second pass only when a CACHED code object failed to load: it is deleted and compiled again)
  code.clear(); lowered.clear(); from_cache = false;
  if (attempt == 0 && !co_path.empty() && jit_private_dir(cache) && jit_private_file(co_path) && jit_private_file(ln_path)) {
    if (FILE* cf = std::fopen(co_path.c_str(), "rb")) {
      std::fseek(cf, 0, SEEK_END); const long sz = std::ftell(cf); std::fseek(cf, 0, SEEK_SET);
      code.resize(sz > 0 ? (size_t)sz : 0);
      if (sz <= 0 || std::fread(code.data(), 1, (size_t)sz, cf) != (size_t)sz) code.clear();
      std::fclose(cf);
    }
    if (FILE* nf = std::fopen(ln_path.c_str(), "r")) { char nb[512] = {0}; if (std::fgets(nb, sizeof nb, nf)) lowered = nb; std::fclose(nf); }
    while (!lowered.empty() && (lowered.back() == '\n' || lowered.back() == ' ')) lowered.pop_back();
    if (lowered.empty()) code.clear();
    from_cache = !code.empty();
  }
  if (code.empty()) {
    const double t0 = now_ms();
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, tu.c_str(), "bssm_user_model.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) { *err = "hiprtcCreateProgram failed"; return nullptr; }
    const bool plain_name = name.find('<') == std::string::npos && name.find("::") == std::string::npos;   // an extern "C" kernel: its own name
    if (!plain_name) (void)hiprtcAddNameExpression(prog, name.c_str());
    const std::string o_arch = "--offload-arch=" + arch, o_inc = "-I" + dir + "/csrc";
    const char* opts[] = {o_arch.c_str(), "-O3", "-std=c++17", o_inc.c_str()};
    const hiprtcResult rc = hiprtcCompileProgram(prog, 4, opts);
    if (rc != HIPRTC_SUCCESS) {
      size_t ls = 0;
      (void)hiprtcGetProgramLogSize(prog, &ls);
      std::string log(ls, 0);
      if (ls) (void)hiprtcGetProgramLog(prog, &log[0]);
      (void)hiprtcDestroyProgram(&prog);
      *err = "compiling the user model in process failed (" + std::string(hiprtcGetErrorString(rc)) + "):\n" + log.substr(0, 4000);
      return nullptr;
    }
    size_t cs = 0;
    (void)hiprtcGetCodeSize(prog, &cs);
    code.resize(cs);
    (void)hiprtcGetCode(prog, code.data());
    const char* ln = nullptr;
    if (plain_name) ln = name.c_str();
    else if (hiprtcGetLoweredName(prog, name.c_str(), &ln) != HIPRTC_SUCCESS || !ln) { (void)hiprtcDestroyProgram(&prog); *err = "no lowered name for " + name; return nullptr; }
    lowered = ln;
    (void)hiprtcDestroyProgram(&prog);
    if (bssm::opt().trace) std::fprintf(stderr, "[bssm_amd] hiprtc: %s in %.0f ms (%zu bytes)\n", name.c_str(), now_ms() - t0, cs);
    if (!co_path.empty()) {  // best effort: a cache that cannot be written only costs the next process a recompile
      make_private_dirs(cache);   // mkdir(2), no shell: the path comes from the environment (BSSM_AMD_USER_CACHE / HOME)
      if (jit_private_dir(cache)) {
        const std::string pid = "." + std::to_string((long)getpid());
        const std::string tmp = co_path + ".tmp" + pid, ln_tmp = ln_path + ".tmp" + pid;   // (ranks compiling the same variant at once: a file each)
        if (FILE* cf = std::fopen(tmp.c_str(), "wb")) {
          const bool okw = std::fwrite(code.data(), 1, code.size(), cf) == code.size();
          std::fclose(cf);
          (void)chmod(tmp.c_str(), 0600);
          if (FILE* nf = std::fopen(ln_tmp.c_str(), "w")) { std::fputs(lowered.c_str(), nf); std::fclose(nf); (void)chmod(ln_tmp.c_str(), 0600); (void)std::rename(ln_tmp.c_str(), ln_path.c_str()); }
          if (okw) (void)std::rename(tmp.c_str(), co_path.c_str()); else (void)std::remove(tmp.c_str());
        }
      }
    }
  }
  hipModule_t mod = nullptr;
  hipFunction_t fn = nullptr;
  if (hipModuleLoadData(&mod, code.data()) == hipSuccess && hipModuleGetFunction(&fn, mod, lowered.c_str()) == hipSuccess) {
    jm.kernels[key] = fn;
    return fn;
  }
  (void)hipGetLastError();
  if (mod) (void)hipModuleUnload(mod);
  if (!from_cache) break;
  // a stale or corrupt cache entry must not fail every later process: drop it and take the compile path
  (void)std::remove(co_path.c_str());
  (void)std::remove(ln_path.c_str());
  }
  *err = "loading the compiled user-model kernel failed: " + name;
  return nullptr;
}

JitModel* jit_of(int slot) { return slot >= 1 && slot <= 63 ? g_jit[slot] : nullptr; }


// ---- host entry points of a model compiled in process: one evaluation-kernel launch per call (see kJitEvalKernel)
struct JitEvalBlock { double alpha[16], theta[64], known[64], out[256]; };
int jit_eval(int slot, int what, int t, const double* ktv, int rows, int cols, const double* a, const double* th, const double* kn, double* out, int out_cnt) {
  JitModel* jm = jit_of(slot);
  if (!jm) return -1;
  std::string err;
  hipFunction_t fn = jit_kernel(*jm, "bssm_user_eval", &err);
  if (!fn) { fail(BSSM_ERR_INVALID, err); return -1; }
  static thread_local JitEvalBlock* blk = nullptr;   // pinned, visible to the device: no copies around the launch
  static thread_local int blk_dev = -1;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!blk || blk_dev != dev) {
    if (blk) (void)hipHostFree(blk);
    if (hipHostMalloc((void**)&blk, sizeof(JitEvalBlock), hipHostMallocMapped) != hipSuccess) { blk = nullptr; (void)hipGetLastError(); return -1; }
    blk_dev = dev;
  }
  const int ms = jm->n_states;
  for (int q = 0; q < ms && q < 16; q++) blk->alpha[q] = a[q];
  std::memcpy(blk->theta, th, sizeof blk->theta);   // (the caller's arrays are FnModel::theta / known: 64 entries each)
  std::memcpy(blk->known, kn, sizeof blk->known);
  double* dtv = nullptr;
  if (ktv && rows > 0 && cols > 0 && device_known_tv(ktv, rows, cols, &dtv) != BSSM_OK) return -1;
  void* dblk = nullptr;
  if (hipHostGetDevicePointer(&dblk, blk, 0) != hipSuccess) { (void)hipGetLastError(); return -1; }
  static thread_local bssm::TlsStream st_tls;
  hipStream_t st = st_tls.get();
  int what_ = what, t_ = t, rows_ = rows, cols_ = cols;
  const double* dtv_c = dtv;
  void* args[] = {&what_, &t_, &dtv_c, &rows_, &cols_, &dblk};
  if (hipModuleLaunchKernel(fn, 1, 1, 1, 64, 1, 1, 0, st, args, nullptr) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { (void)hipGetLastError(); return -1; }
  for (int q = 0; q < out_cnt; q++) out[q] = blk->out[q];
  return 0;
}
template <int SLOT> struct JitTramp {
  static int ms() { JitModel* jm = jit_of(SLOT); return jm ? jm->n_states : 1; }
  static void T(int t, const double* ktv, int r, int c, const double* a, const double* th, const double* kn, double* out) { if (jit_eval(SLOT, 0, t, ktv, r, c, a, th, kn, out, ms())) for (int q = 0; q < ms(); q++) out[q] = std::nan(""); }
  static double Z(int t, const double* ktv, int r, int c, const double* a, const double* th, const double* kn) { double v = std::nan(""); (void)jit_eval(SLOT, 1, t, ktv, r, c, a, th, kn, &v, 1); return v; }
  static void Tg(int t, const double* ktv, int r, int c, const double* a, const double* th, const double* kn, double* out) { if (jit_eval(SLOT, 2, t, ktv, r, c, a, th, kn, out, ms() * ms())) for (int q = 0; q < ms() * ms(); q++) out[q] = std::nan(""); }
  static void Zg(int t, const double* ktv, int r, int c, const double* a, const double* th, const double* kn, double* out) { if (jit_eval(SLOT, 3, t, ktv, r, c, a, th, kn, out, ms())) for (int q = 0; q < ms(); q++) out[q] = std::nan(""); }
  static double H(int t, const double* ktv, int r, int c, const double* a, const double* th, const double* kn) { double v = std::nan(""); (void)jit_eval(SLOT, 4, t, ktv, r, c, a, th, kn, &v, 1); return v; }
  static void R(int t, const double* ktv, int r, int c, const double* a, const double* th, const double* kn, double* out) { if (jit_eval(SLOT, 5, t, ktv, r, c, a, th, kn, out, ms() * ms())) for (int q = 0; q < ms() * ms(); q++) out[q] = std::nan(""); }
};
template <int SLOT> void jit_bind_one(int slot, bssm::UserHostFns* h, bool jac, bool hfn, bool rfn) {
  if (slot != SLOT) return;
  h->T = JitTramp<SLOT>::T; h->Z = JitTramp<SLOT>::Z;
  if (jac) { h->Tg = JitTramp<SLOT>::Tg; h->Zg = JitTramp<SLOT>::Zg; }
  if (hfn) h->H = JitTramp<SLOT>::H;
  if (rfn) h->R = JitTramp<SLOT>::R;
}
template <int... S> void jit_bind_all(int slot, bssm::UserHostFns* h, bool jac, bool hfn, bool rfn, std::integer_sequence<int, S...>) {
  (jit_bind_one<S + 1>(slot, h, jac, hfn, rfn), ...);
}
void jit_bind(int slot, bssm::UserHostFns* h, bool jac, bool hfn, bool rfn) { jit_bind_all(slot, h, jac, hfn, rfn, std::make_integer_sequence<int, 63>{}); }

// workgroup-resident shape: the 512-thread form of launch_wg_mk (pf_filter_impl.h), bootstrap sampler
hipError_t jit_launch_wg(const bssm::FilterBatch* bp, hipStream_t stream) {
  JitModel* jm = jit_of(bp->user_slot);
  if (!jm || bp->psi < 0 || bp->psi > 2 || (bp->psi == 2 && jm->MP > 8) || (bp->psi == 1 && jm->MP > 8 && jm->source.find("BSSM_USER_RFN") != std::string::npos) || bp->MP != jm->MP || bp->KP != jm->MP) { fail(BSSM_ERR_UNSUPPORTED, "a user model compiled in process: EKF particle filter (and the psi-APF with a state-dependent R_fn) up to 8 states"); return hipErrorInvalidValue; }
  bssm::FilterBatch b = *bp;
  b.user_slot = 0;
  const int MP = jm->MP, N = b.N;
  int ppt = 1;
  while (ppt < 16 && N > kJitWgThreads * ppt) ppt *= 2;
  if (ppt == 16 && MP > 2) return hipErrorInvalidValue;
  const int nt = ppt == 1 ? (N + 63) / 64 * 64 : kJitWgThreads;
  const size_t lds = ((size_t)(MP + 1) * nt * ppt + 2 * bssm::CANON_MAX_ROWS + (size_t)(2 * MP * MP + MP)) * sizeof(double);
  if (N > nt * ppt || lds > 160 * 1024) return hipErrorInvalidValue;
  char name[160];
  std::snprintf(name, sizeof name, "bssm::pf_wg_kernel<%d,%d,%d,%d,%d,%d,false>", MP, MP, ppt, jm->fn_id, kJitWgThreads, b.psi);
  std::string err;
  hipFunction_t fn = jit_kernel(*jm, name, &err);
  if (!fn) { fail(BSSM_ERR_INVALID, err); return hipErrorInvalidValue; }
  bssm::note_filter_kernel((std::string(name) + " [hiprtc]").c_str());
  hipError_t e;
  if ((e = hipMemsetAsync(b.dead_at, 0, sizeof(int) * b.F, stream)) != hipSuccess) return e;
  void* args[] = {&b};
  return hipModuleLaunchKernel(fn, (unsigned)b.F, 1, 1, (unsigned)nt, 1, 1, (unsigned)lds, stream, args, nullptr);
}

// one launch per time step: the row family (pf_row_step.h, the time loop of launch_row_loop) up to 16 384 particles, the step kernel
// (pf_grid_v2.h, the time loop of launch_grid_mk) beyond -- with module launches
hipError_t jit_launch_grid(const bssm::FilterBatch* bp, const bssm::GridWorkspace* wsp_in, hipStream_t stream, bssm::StepTiming* timing) {
  JitModel* jm = jit_of(bp->user_slot);
  if (!jm || bp->psi < 0 || bp->psi > 2 || (bp->psi == 2 && jm->MP > 8) || (bp->psi == 1 && jm->MP > 8 && jm->source.find("BSSM_USER_RFN") != std::string::npos) || bp->MP != jm->MP || bp->KP != jm->MP) { fail(BSSM_ERR_UNSUPPORTED, "a user model compiled in process: EKF particle filter (and the psi-APF with a state-dependent R_fn) up to 8 states"); return hipErrorInvalidValue; }
  const bssm::GridWorkspace& ws = *wsp_in;
  constexpr int kGridNT = 256, kMaxBlocks = 256;  // GRID_NT (pf_filter_impl.h), V2_MAX_BLOCKS (pf_grid_v2.h)
  const bool rows = ws.chunk == bssm::CANON_ROW;
  if (rows && bp->N > bssm::CANON_ROW * bssm::ROW_MAX_ROWS) return hipErrorInvalidValue;
  const int ppt = rows ? 0 : ws.chunk / kGridNT;
  if (!rows && (ws.B > kMaxBlocks || (ppt != 1 && ppt != 2 && ppt != 4))) return hipErrorInvalidValue;
  bssm::FilterBatch b = *bp;
  b.user_slot = 0;
  const int MP = jm->MP;
  const bool wx = !rows && b.weights != nullptr && bssm::opt().grid_v2_wx != 0;  // weights exchange, as launch_grid_mk
  char n_step[160], n_init[160];
  if (rows) {
    std::snprintf(n_step, sizeof n_step, "bssm::pf_row_step<%d,%d,%d,%d>", MP, MP, jm->fn_id, bp->psi);
    std::snprintf(n_init, sizeof n_init, "bssm::pf_row_init<%d,%d,%d,%d>", MP, MP, jm->fn_id, bp->psi);
  } else {
    std::snprintf(n_step, sizeof n_step, "bssm::pf_grid_step_v2_fn<%d,%d,%s,%d,%d>", MP, ppt, wx ? "true" : "false", jm->fn_id, bp->psi);
    std::snprintf(n_init, sizeof n_init, "bssm::pf_grid_init_v2<%d,%d,false,%d,%d>", MP, ppt, jm->fn_id, bp->psi);
  }
  std::string err;
  hipFunction_t f_step = jit_kernel(*jm, n_step, &err), f_init = f_step ? jit_kernel(*jm, n_init, &err) : nullptr;
  if (!f_step || !f_init) { fail(BSSM_ERR_INVALID, err); return hipErrorInvalidValue; }
  bssm::note_filter_kernel((std::string(n_step) + " [hiprtc]").c_str());
  hipError_t e;
  if ((e = hipMemsetAsync(b.dead_at, 0, sizeof(int) * b.F, stream)) != hipSuccess) return e;
  if ((e = hipMemsetAsync(ws.llinc, 0, sizeof(double) * b.F * (b.n + 1), stream)) != hipSuccess) return e;
  const int wgs = rows ? (ws.B + 3) / 4 : ws.B;
  int parts = ((rows ? b.F >= 8 : ((long long)b.F * ws.B >= (ws.chunk >= 2 * kGridNT ? 384 : 512) && b.F >= 2)) && ws.side) ? 2 : 1;
  hipStream_t lanes[2] = {stream, parts == 2 ? ws.side : stream};
  bssm::GridWorkspace w2[2] = {ws, ws};
  for (int q = 0; q < 2; q++) { w2[q].wx = wx ? 1 : 0; w2[q].fw = 0; }
  int fcount[2] = {b.F, 0};
  hipEvent_t fork = nullptr, join = nullptr;
  if (parts == 2) {
    fcount[0] = b.F / 2; fcount[1] = b.F - fcount[0];
    w2[1].f0 = ws.f0 + fcount[0];
    if ((e = hipEventCreateWithFlags(&fork, hipEventDisableTiming)) != hipSuccess) return e;
    (void)hipEventRecord(fork, stream);
    (void)hipStreamWaitEvent(lanes[1], fork, 0);
  }
  if (timing) timing->concurrent = parts;
  for (int t = -1; t < b.n; t++)
    for (int q = 0; q < parts; q++) {
      int tt = t;
      void* a_init[] = {&b, &w2[q]};
      void* a_step[] = {&b, &w2[q], &tt};
      e = hipModuleLaunchKernel(t < 0 ? f_init : f_step, (unsigned)wgs, (unsigned)fcount[q], 1, 256, 1, 1, 0, lanes[q], t < 0 ? a_init : a_step, nullptr);
      if (e != hipSuccess) return e;
    }
  if (parts == 2) {
    if ((e = hipEventCreateWithFlags(&join, hipEventDisableTiming)) != hipSuccess) return e;
    (void)hipEventRecord(join, lanes[1]);
    (void)hipStreamWaitEvent(stream, join, 0);
    (void)hipEventDestroy(join);
    (void)hipEventDestroy(fork);
  }
  return bssm::launch_loglik_finalize(b, ws, stream);
}
}  // namespace
extern "C" {

int bssm_user_model_jit(const char* source, int32_t n_states) {
  if (!source) return fail(BSSM_ERR_INVALID, "NULL argument");
  if (n_states < 1 || n_states > 16) return fail(BSSM_ERR_UNSUPPORTED, "user-defined models: n_states must be 1 ... 16");
  if (!have_device()) return no_device();
  const std::string src(source);
  const bool sde = src.find("BSSM_USER_SDE") != std::string::npos, nlg = src.find("BSSM_USER_NLG") != std::string::npos;
  if (sde == nlg) return fail(BSSM_ERR_INVALID, "the snippet must define exactly one of BSSM_USER_NLG and BSSM_USER_SDE");
  {  // the same snippet again (the R shim asks on every filter call): the slot it already has, as bssm_user_model_load does by path
    std::lock_guard<std::mutex> lk(g_user_mu);
    for (int q = 1; q < 64; q++) if (g_jit[q] && g_jit[q]->n_states == n_states && g_jit[q]->source == src) return q;
  }
  auto* jm = new JitModel();
  jm->source = src; jm->n_states = n_states; jm->MP = bssm::pad_dim(n_states); jm->fn_id = sde ? bssm::FN_USER_SDE : bssm::FN_USER_NLG;
  // the smallest workgroup-resident variant now: a snippet that does not compile fails HERE, with the compiler's log, and a first
  // filter call of up to 512 particles finds its kernel ready
  char name[160];
  std::snprintf(name, sizeof name, "bssm::pf_wg_kernel<%d,%d,1,%d,%d,0,false>", jm->MP, jm->MP, jm->fn_id, kJitWgThreads);
  std::string err;
  if (!jit_kernel(*jm, name, &err)) { delete jm; return fail(BSSM_ERR_INVALID, err); }
  std::lock_guard<std::mutex> lk(g_user_mu);
  for (int q = 1; q < 64; q++) if (g_jit[q] && g_jit[q]->n_states == n_states && g_jit[q]->source == src) { delete jm; return q; }  // (a concurrent first call won)
  const int slot = bssm::register_user_launchers(jit_launch_wg, jit_launch_grid);
  if (slot < 1) { delete jm; return fail(BSSM_ERR_UNSUPPORTED, "too many user models loaded"); }
  g_jit[slot] = jm;
  g_user_host[slot] = bssm::UserHostFns();
  g_user_host[slot].has_tv = src.find("BSSM_USER_TV") != std::string::npos ? 1 : 0;
  g_user_host[slot].jit = 1;
  if (nlg)   // host entry points for the psi-APF / EKF set-up: the evaluation kernel behind function pointers (round 6)
    jit_bind(slot, &g_user_host[slot], src.find("BSSM_USER_JACOBIANS") != std::string::npos, src.find("BSSM_USER_HFN") != std::string::npos,
             src.find("BSSM_USER_RFN") != std::string::npos);
  bssm::mark_user_slot_rows_only(slot);
  return slot;
}

// ------------------------------------------------------------------ exported helpers
int bssm_amd_reload_env(void) {
  bssm::reload_options();
  return BSSM_OK;
}

int bssm_stratified_sample(double* p, const double* r, uint32_t n_elem, uint32_t N, uint32_t* out) {
  if (!p || !r || !out || n_elem < 1 || N < 1) return fail(BSSM_ERR_INVALID, "NULL argument or empty input");
  for (uint32_t j = 0; j < N; j++)
    if (!(r[j] >= 0.0 && r[j] < 1.0)) return fail(BSSM_ERR_INVALID, "r must hold U(0,1) draws (src/stratified_sample.cpp:6)");
  if (!have_device()) return no_device();
  DevBuf dp, dr, dout;
  HIP_TRY(dp.alloc(n_elem * sizeof(double)));
  HIP_TRY(dr.alloc(N * sizeof(double)));
  HIP_TRY(dout.alloc(N * sizeof(uint32_t)));
  HIP_TRY(hipMemcpy(dp.p, p, dp.bytes, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dr.p, r, dr.bytes, hipMemcpyHostToDevice));
  HIP_TRY(bssm::launch_stratified_sample(dp.as<double>(), dr.as<double>(), dout.as<uint32_t>(), (int)n_elem, (int)N, nullptr));
  HIP_TRY(hipMemcpy(out, dout.p, dout.bytes, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(p, dp.p, dp.bytes, hipMemcpyDeviceToHost));  // the reference takes p by reference and leaves the cumsum in it (:9,:12-13)
  return BSSM_OK;
}

}  // extern "C"
namespace bssm_api {   // (sample_sd is used by the multi-device suggest_N of api_postcorrect.cpp)
// arma::stddev (N - 1 normalisation), src/R_postcorrection.cpp:35
double sample_sd(const std::vector<double>& x) {
  const size_t n = x.size();
  if (n < 2) return 0.0;
  double mean = 0.0;
  for (double v : x) mean += v;
  mean /= (double)n;
  double ss = 0.0;
  for (double v : x) ss += (v - mean) * (v - mean);
  return std::sqrt(ss / (double)(n - 1));
}
int suggest_n_core(const uint32_t* candidates, int32_t n_candidates, uint32_t replications, double* sds,
                   const std::function<int(uint32_t, uint32_t, double*)>& batch) {
  if (!candidates || !sds || n_candidates < 1 || replications < 1) return fail(BSSM_ERR_INVALID, "bad arguments");
  bool deg = false;
  std::vector<double> ll(replications);
  for (int i = 0; i < n_candidates; i++) {
    if (candidates[i] < 1) return fail(BSSM_ERR_INVALID, "candidates must be positive");
    // the `replications` runs of one candidate are ONE device batch; streams i * replications + j keep candidates independent
    const int rc = batch(candidates[i], (uint32_t)i * replications, ll.data());
    if (rc < 0) return rc;
    deg = deg || rc == BSSM_WARN_DEGENERATE;
    sds[i] = sample_sd(ll);
  }
  return deg ? BSSM_WARN_DEGENERATE : BSSM_OK;
}
}  // namespace bssm_api
extern "C" {

int bssm_suggest_n_nongaussian(const bssm_ung_model* model, const uint32_t* candidates, int32_t n_candidates,
                               uint32_t replications, uint32_t seed, int model_type, double* sds) {
  if (!is_gaussian_type_ok(0, model_type)) return fail(BSSM_ERR_UNSUPPORTED, "model_type not handled by this entry point");
  return suggest_n_core(candidates, n_candidates, replications, sds, [&](uint32_t nsim, uint32_t stream0, double* ll) {
    return bssm_nongaussian_loglik_batch(model, nsim, BSSM_PSI, seed, stream0, replications, ll);
  });
}

int bssm_suggest_n_nonlinear(const bssm_fn_model* model, const uint32_t* candidates, int32_t n_candidates,
                             uint32_t replications, uint32_t seed, double* sds) {
  return suggest_n_core(candidates, n_candidates, replications, sds, [&](uint32_t nsim, uint32_t stream0, double* ll) {
    return bssm_fn_loglik_batch(model, nsim, BSSM_PSI, seed, stream0, replications, ll);
  });
}
}  // extern "C"
