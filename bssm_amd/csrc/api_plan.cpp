// api_plan.cpp -- the C ABI (include/bssm_amd.h), part 3: device-resident plans (fixed models, only the seed or theta changes)
// and the debug / measurement hooks.
#include "api_internal.h"
#include "api_shared.h"
using namespace bssm_api;

extern "C" {
// ------------------------------------------------------------------ device-resident plan
struct bssm_plan {
  DeviceBatch db;
  int n_models = 0, n_rep = 0;
  bool psi = false;
  double bytes_per_pstep = 0;
  // local_approx = FALSE: pseudo-observations of each model's first approximation, reused by bssm_plan_update
  std::vector<bssm::GlobalApprox> global;
  int launch_sequences = 1;  // concurrent step-launch sequences of the last bssm_plan_run / bssm_plan_time_kernel
};

namespace {
// tables (one per model x replica) and additive log-likelihood terms of a plan's models
int plan_tables(const bssm_ung_model* models, int n_models, int n_rep, bool psi, std::vector<bssm::FilterTables>* tabs,
                std::vector<double>* consts, int* n, int* m, int* k, int* pp, std::vector<bssm::GlobalApprox>* global) {
  if (global) global->resize(n_models);
  tabs->clear(); consts->clear();
  for (int i = 0; i < n_models; i++) {
    bssm::HostModel hm;
    const std::string err = bssm::ingest_model(&models[i], psi, &hm);
    if (!err.empty()) return fail(BSSM_ERR_INVALID, err);
    if (i == 0) { *n = hm.n; *m = hm.m; *k = hm.k; *pp = hm.p; }
    else if (hm.n != *n || hm.m != *m || hm.k != *k || hm.p != *pp) return fail(BSSM_ERR_INVALID, "models of a plan must share n, m, k, p");
    bssm::FilterTables ft;
    double base;
    const int MP = bssm::pad_dim(*m);
    if (psi) {
      bssm::PsiSetup su;
      bssm::GlobalApprox* g = (global && !hm.local_approx) ? &(*global)[i] : nullptr;
      if (!bssm::psi_setup(hm, nullptr, &su, g)) return fail(BSSM_ERR_NUMERIC, "psi set-up produced non-finite values");
      if (g && !g->set) { g->y = su.approx_y; g->H = su.approx_H; g->set = true; }
      bssm::build_psi_tables(hm, su, MP, &ft);
      base = su.approx_loglik;
    } else {
      bssm::build_bsf_tables(hm, MP, bssm::pad_noise_dim(MP, *k), &ft);
      base = bssm::const_term(hm);
    }
    for (int r = 0; r < n_rep; r++) { tabs->push_back(ft); consts->push_back(base); }
  }
  return BSSM_OK;
}
}  // namespace

bssm_plan* bssm_plan_create(const bssm_ung_model* models, int32_t n_models, int32_t n_rep, uint32_t nsim,
                            uint32_t sampling_method, uint32_t flags) {
  if (!models || n_models < 1 || n_rep < 1 || nsim < 1) { fail(BSSM_ERR_INVALID, "bad plan arguments"); return nullptr; }
  if (sampling_method != BSSM_PSI && sampling_method != BSSM_BSF) { fail(BSSM_ERR_UNSUPPORTED, "sampling_method must be psi or bsf"); return nullptr; }
  if (!have_device()) { no_device(); return nullptr; }
  const bool psi = sampling_method == BSSM_PSI;
  std::vector<bssm::FilterTables> tabs;
  std::vector<double> consts;
  int n = 0, m = 0, k = 0, pp = 1;
  std::unique_ptr<bssm_plan> p(new bssm_plan);
  if (plan_tables(models, n_models, n_rep, psi, &tabs, &consts, &n, &m, &k, &pp, &p->global) != BSSM_OK) return nullptr;
  p->n_models = n_models; p->n_rep = n_rep; p->psi = psi;
  const bool keep = flags & BSSM_PLAN_KEEP_HISTORY;
  p->db.want_factored = keep && (flags & BSSM_PLAN_FACTORED_WEIGHTS);
  if (p->db.upload(tabs, n, m, k, pp, (int)nsim, psi, keep, flags) != BSSM_OK) return nullptr;
  p->db.const_terms = consts;
  // full-output filter 16m + 24 bytes per particle-step; log-likelihood only 16m + 16
  p->bytes_per_pstep = keep ? 16.0 * m + 24.0 : 16.0 * m + 16.0;
  return p.release();
}

// New parameter values for the same model structure (pseudo-marginal / delayed-acceptance MCMC proposes a theta per
// iteration, src/mcmc.cpp:337-491): the tables are rebuilt on the host and copied into the plan's EXISTING device buffers
// -- no allocation, no change of kernel shape.
int bssm_plan_update(bssm_plan* plan, const bssm_ung_model* models, void* hip_stream) {
  if (!plan || !models) return fail(BSSM_ERR_INVALID, "NULL argument");
  std::vector<bssm::FilterTables> tabs;
  std::vector<double> consts;
  int n = 0, m = 0, k = 0, pp = 1;
  const int rc = plan_tables(models, plan->n_models, plan->n_rep, plan->psi, &tabs, &consts, &n, &m, &k, &pp, &plan->global);
  if (rc) return rc;
  const bssm::FilterBatch& fb = plan->db.fb;
  if (n != fb.n || m != fb.m || k != fb.k || pp != fb.p || tabs[0].MP != fb.MP || tabs[0].KP != fb.KP ||
      tabs[0].mats.size() != fb.mats_fstride || tabs[0].obs.size() != fb.obs_fstride || tabs[0].mats_tstride != fb.mats_tstride)
    return fail(BSSM_ERR_INVALID, "bssm_plan_update: the models must keep the structure the plan was created with");
  const size_t ms = fb.mats_fstride, os = fb.obs_fstride, is = fb.init_fstride, F = tabs.size();
  StageBuf stage;
  HIP_TRY(stage.alloc((ms + os + is) * F * sizeof(double)));
  double* hm = (double*)stage.p; double* ho = hm + ms * F; double* hi = ho + os * F;
  for (size_t f = 0; f < F; f++) {
    std::memcpy(hm + ms * f, tabs[f].mats.data(), ms * sizeof(double));
    std::memcpy(ho + os * f, tabs[f].obs.data(), os * sizeof(double));
    std::memcpy(hi + is * f, tabs[f].init.data(), is * sizeof(double));
  }
  hipStream_t st = (hipStream_t)hip_stream;
  HIP_TRY(hipMemcpyAsync(plan->db.mats.p, hm, ms * F * sizeof(double), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(plan->db.obs.p, ho, os * F * sizeof(double), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(plan->db.init.p, hi, is * F * sizeof(double), hipMemcpyHostToDevice, st));
  HIP_TRY(hipStreamSynchronize(st));
  plan->db.const_terms = consts;
  return BSSM_OK;
}

int bssm_plan_approx_loglik(const bssm_plan* plan, double* approx_loglik) {
  if (!plan || !approx_loglik) return fail(BSSM_ERR_INVALID, "NULL argument");
  if (!plan->psi) return fail(BSSM_ERR_INVALID, "a bootstrap plan has no approximate log-likelihood");
  for (size_t f = 0; f < plan->db.const_terms.size(); f++) approx_loglik[f] = plan->db.const_terms[f];
  return BSSM_OK;
}

// sample_or_summarise (src/summary.cpp:82-103) on the plan's kept particle system of filter `filter`:
//   alpha    != NULL: one back-traced trajectory, (n+1) x m, drawn uniformly (the weights of the extra time point are
//                     ones) from the stream (seed, purpose 3) of that filter
//   alphahat != NULL: smoothed means m x (n+1) and covariances Vt m x m x (n+1) of the back-traced system (weights uniform)
int bssm_plan_sample_or_summarise(bssm_plan* plan, int32_t filter, uint32_t seed, void* hip_stream, double* alpha,
                                  double* alphahat, double* Vt) {
  if (!plan) return fail(BSSM_ERR_INVALID, "plan NULL");
  const bssm::FilterBatch& fb = plan->db.fb;
  if (!fb.alpha_hist || !fb.indices) return fail(BSSM_ERR_INVALID, "the plan was created without BSSM_PLAN_KEEP_HISTORY");
  if (filter < 0 || filter >= fb.F) return fail(BSSM_ERR_INVALID, "filter index out of range");
  const int n = fb.n, m = fb.m, N = fb.N;
  hipStream_t st = (hipStream_t)hip_stream;
  const double* hist = fb.alpha_hist + (size_t)filter * (n + 1) * m * N;
  const uint32_t* idx = fb.indices + (size_t)filter * n * N;
  if (alpha) {
    const bssm::u32x4 w = bssm::philox4x32_10(0, 0, 0, bssm::RNG_PICK, seed, fb.stream0 + (uint32_t)filter);
    int pick = (int)((double)bssm::bits53(w.x, w.y) * 0x1.0p-53 * N);
    const uint32_t pk = (uint32_t)(pick >= N ? N - 1 : pick);
    DevBuf d_pick, d_traj;
    HIP_TRY(d_pick.alloc(sizeof(uint32_t)));
    HIP_TRY(d_traj.alloc((size_t)(n + 1) * m * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(d_pick.p, &pk, sizeof pk, hipMemcpyHostToDevice, st));
    HIP_TRY(bssm::launch_pick_trajectory(hist, idx, d_pick.as<uint32_t>(), d_traj.as<double>(), 1, n, m, N, st));
    std::vector<double> tr((size_t)(n + 1) * m);
    HIP_TRY(hipMemcpyAsync(tr.data(), d_traj.p, d_traj.bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int t = 0; t <= n; t++) for (int c = 0; c < m; c++) alpha[t + (size_t)(n + 1) * c] = tr[(size_t)t * m + c];
  }
  if (alphahat || Vt) {
    DevBuf d_b, d_mean, d_cov, d_smooth;
    HIP_TRY(d_mean.alloc((size_t)(n + 1) * m * sizeof(double)));
    HIP_TRY(d_cov.alloc((size_t)(n + 1) * m * m * sizeof(double)));
    if (m <= 8) {   // the fused back-trace + summary kernel
      HIP_TRY(d_b.alloc((size_t)N * sizeof(uint32_t)));
      HIP_TRY(bssm::launch_smooth_summary(hist, idx, d_b.as<uint32_t>(), d_mean.as<double>(), d_cov.as<double>(), 1, n, m, N, st));
    } else {        // up to 64 states (round 6): filter_smoother into a work copy, then summary (src/filter_smoother.cpp:5-18, src/summary.cpp:4-17), as the smoother entry points do
      HIP_TRY(d_smooth.alloc((size_t)(n + 1) * m * N * sizeof(double)));
      HIP_TRY(bssm::launch_backtrace(hist, idx, d_smooth.as<double>(), 1, n, m, N, st));
      HIP_TRY(bssm::launch_summary(d_smooth.as<double>(), d_mean.as<double>(), d_cov.as<double>(), 1, n + 1, m, N, st));
    }
    if (alphahat) HIP_TRY(hipMemcpyAsync(alphahat, d_mean.p, d_mean.bytes, hipMemcpyDeviceToHost, st));
    if (Vt) HIP_TRY(hipMemcpyAsync(Vt, d_cov.p, d_cov.bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  return BSSM_OK;
}

void bssm_plan_destroy(bssm_plan* plan) {
  if (!plan) return;
  (void)hipDeviceSynchronize();  // the plan's buffers return to the arena: nothing may still be running on them
  delete plan;
}

int bssm_plan_run(bssm_plan* plan, uint32_t seed, uint32_t stream0, void* hip_stream) {
  if (!plan) return fail(BSSM_ERR_INVALID, "plan NULL");
  bssm::StepTiming tm;  // no timing requested: only records how many launch sequences the step kernels were issued as
  const int rc = plan->db.run(seed, stream0, (hipStream_t)hip_stream, &tm);
  plan->launch_sequences = tm.concurrent;
  return rc;
}

int bssm_plan_loglik(bssm_plan* plan, void* hip_stream, double* loglik) {
  if (!plan || !loglik) return fail(BSSM_ERR_INVALID, "NULL argument");
  return plan->db.fetch_loglik((hipStream_t)hip_stream, loglik);
}

int bssm_plan_info(const bssm_plan* plan, int32_t* path, int32_t* n_filters, int32_t* blocks, int32_t* threads,
                   double* bytes_per_particle_step) {
  if (!plan) return fail(BSSM_ERR_INVALID, "plan NULL");
  const auto& db = plan->db;
  if (path) *path = db.use_grid ? 1 : 0;
  if (n_filters) *n_filters = db.fb.F;
  if (db.use_grid) {
    if (blocks) *blocks = db.gws.B * db.fb.F;
    if (threads) *threads = 256;
  } else {
    if (blocks) *blocks = db.fb.F;
    const int N = db.fb.N;
    if (threads) *threads = N <= 512 ? (N + 63) / 64 * 64 : 512;
  }
  if (bytes_per_particle_step) *bytes_per_particle_step = plan->bytes_per_pstep;
  return BSSM_OK;
}

int bssm_amd_last_kernel(char* buf, int32_t len) {
  if (!buf || len < 1) return fail(BSSM_ERR_INVALID, "bad arguments");
  std::snprintf(buf, (size_t)len, "%s", bssm::last_filter_kernel());
  return BSSM_OK;
}

int bssm_amd_family_launches(uint64_t* wg, uint64_t* step, uint64_t* row) {
  unsigned long long c[3];
  bssm::family_launches(c);
  if (wg) *wg = c[0];
  if (step) *step = c[1];
  if (row) *row = c[2];
  return BSSM_OK;
}

int bssm_plan_device_outputs(const bssm_plan* plan, void** alpha_hist, void** indices, void** weights) {
  if (!plan) return fail(BSSM_ERR_INVALID, "plan NULL");
  if (alpha_hist) *alpha_hist = plan->db.hist.p;
  if (indices) *indices = plan->db.indices.p;
  if (weights) *weights = plan->db.weights.p;
  return BSSM_OK;
}

int bssm_plan_fetch(const bssm_plan* plan, int32_t filter, void* hip_stream, double* alpha_hist, uint32_t* indices, double* weights) {
  if (!plan) return fail(BSSM_ERR_INVALID, "plan NULL");
  const bssm::FilterBatch& fb = plan->db.fb;
  if (filter < 0 || filter >= fb.F) return fail(BSSM_ERR_INVALID, "filter index outside the plan");
  if (!plan->db.hist.p) return fail(BSSM_ERR_INVALID, "the plan was created without BSSM_PLAN_KEEP_HISTORY");
  hipStream_t st = (hipStream_t)hip_stream;
  const size_t n = fb.n, m = fb.m, N = fb.N, f = (size_t)filter;
  if (alpha_hist) HIP_TRY(hipMemcpyAsync(alpha_hist, plan->db.hist.as<double>() + f * (n + 1) * m * N, (n + 1) * m * N * sizeof(double), hipMemcpyDeviceToHost, st));
  if (indices) HIP_TRY(hipMemcpyAsync(indices, plan->db.indices.as<uint32_t>() + f * n * N, n * N * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  std::vector<double> scales;
  const size_t B = plan->db.gws.B, chunk = plan->db.gws.chunk;
  if (weights) {
    if (!plan->db.weights.p) return fail(BSSM_ERR_INVALID, "the plan keeps no weights");
    HIP_TRY(hipMemcpyAsync(weights, plan->db.weights.as<double>() + f * (n + 1) * N, (n + 1) * N * sizeof(double), hipMemcpyDeviceToHost, st));
    if (plan->db.weights_factored) {
      scales.resize((n + 1) * B);
      HIP_TRY(hipMemcpyAsync(scales.data(), plan->db.gws.wscale + f * (n + 1) * B, scales.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    }
  }
  HIP_TRY(hipStreamSynchronize(st));
  if (weights && plan->db.weights_factored) {
    // factored form -> the reference's exp(w - max_t): one factor per (time point, particle block); the last column (ones,
    // or zeros behind a degenerate step) is final as stored
    for (size_t t = 0; t < n; t++)
      for (size_t blk = 0; blk < B; blk++) {
        const double sc = scales[t * B + blk];
        double* w = weights + t * N + blk * chunk;
        const size_t len = std::min(chunk, N - blk * chunk);
        for (size_t i = 0; i < len; i++) w[i] *= sc;
      }
  }
  return BSSM_OK;
}

int bssm_plan_weights_layout(const bssm_plan* plan, int32_t* factored, int32_t* block_particles, int32_t* n_blocks, void** scales) {
  if (!plan) return fail(BSSM_ERR_INVALID, "plan NULL");
  const bool fw = plan->db.weights_factored;
  if (factored) *factored = fw ? 1 : 0;
  if (block_particles) *block_particles = fw ? plan->db.gws.chunk : 0;
  if (n_blocks) *n_blocks = fw ? plan->db.gws.B : 0;
  if (scales) *scales = fw ? (void*)plan->db.gws.wscale : nullptr;
  return BSSM_OK;
}

int bssm_plan_time(bssm_plan* plan, uint32_t seed, int32_t reps, void* hip_stream, double* ms_per_run) {
  if (!plan || !ms_per_run || reps < 1) return fail(BSSM_ERR_INVALID, "bad arguments");
  hipStream_t st = (hipStream_t)hip_stream;
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  HIP_TRY(hipEventRecord(e0, st));
  for (int r = 0; r < reps; r++) {
    const int rc = plan->db.run(seed + (uint32_t)r, 0, st);
    if (rc) return rc;
  }
  HIP_TRY(hipEventRecord(e1, st));
  HIP_TRY(hipEventSynchronize(e1));
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ms_per_run = (double)ms / reps;
  return BSSM_OK;
}

// test hook: the device elementary functions (0 exp, 1 log, 2 sincos(2 pi x) -> 2n outputs, 3 Box-Muller pairs)
// diagnostic: workgroups of the conditional-factor kernels (time points >= 1) that took the full Jacobi path / all of them, since the last call
int bssm_debug_thin_counts(uint64_t out[2]) {
  unsigned long long c[2] = {0, 0};
  HIP_TRY(bssm::debug_thin_counts(c));
  out[0] = c[0]; out[1] = c[1];
  return BSSM_OK;
}
int bssm_debug_stream_copy(uint64_t n_doubles) {
  if (!have_device()) return no_device();
  DevBuf a, b;
  HIP_TRY(a.alloc(n_doubles * sizeof(double)));
  HIP_TRY(b.alloc(n_doubles * sizeof(double)));
  HIP_TRY(hipMemsetAsync(a.p, 0, a.bytes, nullptr));
  HIP_TRY(bssm::launch_calib_copy(a.as<double>(), b.as<double>(), (size_t)n_doubles, nullptr));
  HIP_TRY(hipDeviceSynchronize());
  return BSSM_OK;
}

int bssm_debug_math(int fn, const double* in, double* out, int32_t n) {
  if (!have_device()) return no_device();
  if (!in || !out || n < 1) return fail(BSSM_ERR_INVALID, "bad arguments");
  const int mult = fn == 4 ? 4 : ((fn == 2 || fn == 3 || fn == 12) ? 2 : 1);  // doubles written per input (debug_math_kernel)
  DevBuf di, dout;
  HIP_TRY(di.alloc((size_t)n * sizeof(double)));
  HIP_TRY(dout.alloc((size_t)n * mult * sizeof(double)));
  HIP_TRY(hipMemcpy(di.p, in, di.bytes, hipMemcpyHostToDevice));
  HIP_TRY(bssm::launch_debug_math(fn, di.as<double>(), dout.as<double>(), n, nullptr));
  HIP_TRY(hipMemcpy(out, dout.p, dout.bytes, hipMemcpyDeviceToHost));
  return BSSM_OK;
}

// debug: cycle stamps of the grid step kernel (block 0, thread 0), averaged over time steps
int bssm_plan_debug_timeline(bssm_plan* plan, uint32_t seed, double* avg_delta_cycles /*[12]*/) {
  if (!plan || !plan->db.use_grid) return fail(BSSM_ERR_INVALID, "grid plan required");
  const int n = plan->db.fb.n;
  DevBuf dbg;
  HIP_TRY(dbg.alloc((size_t)n * 16 * sizeof(long long)));
  HIP_TRY(hipMemset(dbg.p, 0, dbg.bytes));
  plan->db.gws.dbg = dbg.as<long long>();
  const int rc = plan->db.run(seed, 0, nullptr);
  plan->db.gws.dbg = nullptr;
  if (rc) return rc;
  HIP_TRY(hipDeviceSynchronize());
  std::vector<long long> h((size_t)n * 16);
  HIP_TRY(hipMemcpy(h.data(), dbg.p, dbg.bytes, hipMemcpyDeviceToHost));
  for (int k = 0; k < 12; k++) avg_delta_cycles[k] = 0;
  int cnt = 0;
  for (int t = 1; t < n; t++) {
    if (!h[(size_t)t * 16 + 11]) continue;
    for (int k = 1; k < 12; k++) avg_delta_cycles[k] += (double)(h[(size_t)t * 16 + k] - h[(size_t)t * 16 + k - 1]);
    cnt++;
  }
  for (int k = 1; k < 12; k++) avg_delta_cycles[k] /= cnt ? cnt : 1;
  avg_delta_cycles[0] = cnt;
  return BSSM_OK;
}

int bssm_plan_time_kernel(bssm_plan* plan, uint32_t seed, int32_t sample_every, void* hip_stream,
                          double* avg_kernel_ms, int32_t* n_samples) {
  if (!plan || !avg_kernel_ms) return fail(BSSM_ERR_INVALID, "bad arguments");
  bssm::StepTiming tm;
  tm.every = sample_every > 0 ? sample_every : 16;
  const int rc = plan->db.run(seed, 0, (hipStream_t)hip_stream, &tm);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
  *avg_kernel_ms = tm.count ? tm.total_ms / tm.count : 0.0;
  if (n_samples) *n_samples = tm.count;
  plan->launch_sequences = tm.concurrent;
  return BSSM_OK;
}

int bssm_plan_launch_sequences(const bssm_plan* plan) { return plan ? plan->launch_sequences : 0; }

}  // extern "C"

