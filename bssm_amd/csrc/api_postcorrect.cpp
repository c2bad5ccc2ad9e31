// api_postcorrect.cpp -- the C ABI (include/bssm_amd.h), part 2: IS post-correction (one filter per retained theta-draw) as a
// pipeline over chunks of draws, and the same call sharded over several GPUs inside one process.
#include "api_internal.h"
#include "api_shared.h"
using namespace bssm_api;

extern "C" {
// ------------------------------------------------------------------ IS post-correction
namespace {
// psi set-up smoothers of several draws as ONE device batch (lg_setup_kernel, one lane per draw) on a stream of its
// own, so that it overlaps the filter kernels of the previous chunk running on the null stream.  outs[q] = {alphahat,
// Vt, Ct} of draw q; ok[q] = 0 where the result is not finite.  Returns 1 when the draws do not share one layout
// (the caller then runs them on the host), 0 on success, negative on a HIP error.
int device_lg_batch(const std::vector<const bssm::LgPacked*>& lgs, const std::vector<std::array<double*, 3>>& outs,
                    std::vector<char>* ok) {
  const int F = (int)lgs.size();
  if (F == 0) return 0;
  const bssm::LgPacked& L0 = *lgs[0];
  for (int q = 1; q < F; q++) if (!L0.same_layout(*lgs[q])) return 1;
  // highest priority: the lanes of this small kernel compete with filter workgroups that fill every CU
  static thread_local bssm::TlsStream st_tls;
  hipStream_t st = st_tls.get(true);
  if (!st) return fail(BSSM_ERR_HIP, "no stream for the set-up smoother");
  const size_t n = L0.n, m = L0.m, mm = m * m, n1 = n + 1, ins = L0.buf.size();
  const size_t o_at = 0, o_pt = o_at + n1 * m, o_cc = o_pt + n1 * mm, outs_per = o_cc + n1 * mm;
  StageBuf sin, sout;
  HIP_TRY(sin.alloc(ins * F * sizeof(double)));
  HIP_TRY(sout.alloc((outs_per * sizeof(double) + sizeof(int)) * F));
  for (int q = 0; q < F; q++) std::memcpy((double*)sin.p + ins * q, lgs[q]->buf.data(), ins * sizeof(double));
  DevBuf din, din_t, dat, dpt, dcc, dat_t, dpt_t, dcc_t, dK, dv, dF, dok;
  HIP_TRY(din.alloc(ins * F * sizeof(double)));
  HIP_TRY(din_t.alloc(ins * F * sizeof(double)));
  HIP_TRY(dat.alloc(n1 * m * F * sizeof(double)));
  HIP_TRY(dpt.alloc(n1 * mm * F * sizeof(double)));
  HIP_TRY(dcc.alloc(n1 * mm * F * sizeof(double)));
  HIP_TRY(dat_t.alloc(n1 * m * F * sizeof(double)));
  HIP_TRY(dpt_t.alloc(n1 * mm * F * sizeof(double)));
  HIP_TRY(dcc_t.alloc(n1 * mm * F * sizeof(double)));
  HIP_TRY(dK.alloc(n * m * F * sizeof(double)));
  HIP_TRY(dv.alloc(n * F * sizeof(double)));
  HIP_TRY(dF.alloc(n * F * sizeof(double)));
  HIP_TRY(dok.alloc(F * sizeof(int)));
  const double t_in = now_ms();
  HIP_TRY(hipMemcpyAsync(din.p, sin.p, ins * F * sizeof(double), hipMemcpyHostToDevice, st));
  HIP_TRY(bssm::launch_transpose(din.as<double>(), din_t.as<double>(), F, (int)ins, st));  // [F][E] -> [E][F]
  bssm::LgBatchDev b{};
  b.F = F; b.n = (int)n; b.m = (int)m; b.sub = L0.sub;
  b.Zs = L0.Zs; b.Ts = L0.Ts; b.Rs = L0.Rs; b.Ds = L0.Ds; b.Cs = L0.Cs; b.Hs = L0.Hs;
  b.oZ = L0.oZ; b.oT = L0.oT; b.oR = L0.oR; b.oD = L0.oD; b.oC = L0.oC; b.oy = L0.oy; b.oH = L0.oH; b.oa = L0.oa; b.oP = L0.oP;
  b.in_stride = ins;
  b.in = din_t.as<double>(); b.at = dat_t.as<double>(); b.Pt = dpt_t.as<double>(); b.cc = dcc_t.as<double>();
  b.K = dK.as<double>(); b.v = dv.as<double>(); b.Fs = dF.as<double>(); b.ok = dok.as<int>();
  HIP_TRY(bssm::launch_lg_setup(b, st));
  HIP_TRY(bssm::launch_transpose(dat_t.as<double>(), dat.as<double>(), (int)(n1 * m), F, st));   // [E][F] -> [F][E]
  HIP_TRY(bssm::launch_transpose(dpt_t.as<double>(), dpt.as<double>(), (int)(n1 * mm), F, st));
  HIP_TRY(bssm::launch_transpose(dcc_t.as<double>(), dcc.as<double>(), (int)(n1 * mm), F, st));
  double* ho = (double*)sout.p;
  int* hok = (int*)(ho + outs_per * F);
  HIP_TRY(hipMemcpyAsync(ho + o_at * F, dat.p, n1 * m * F * sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(ho + o_pt * F, dpt.p, n1 * mm * F * sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(ho + o_cc * F, dcc.p, n1 * mm * F * sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(hok, dok.p, F * sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (bssm::opt().trace)
    std::fprintf(stderr, "[bssm_amd] psi set-up smoother on device: F=%d n=%zu m=%zu: %.2f ms (copies + kernel)\n", F, n, m, now_ms() - t_in);
  ok->assign(F, 1);
  for (int q = 0; q < F; q++) {
    std::memcpy(outs[q][0], ho + o_at * F + n1 * m * q, n1 * m * sizeof(double));
    std::memcpy(outs[q][1], ho + o_pt * F + n1 * mm * q, n1 * mm * sizeof(double));
    std::memcpy(outs[q][2], ho + o_cc * F + n1 * mm * q, n1 * mm * sizeof(double));
    (*ok)[q] = hok[q] ? 1 : 0;
  }
  return 0;
}
// The device smoother shares the GPU with the filter kernels of the previous chunk (measured 4 ms alone, 10-20 ms
// under contention per 512 draws of n = 300, m = 2); the host smoother costs 0.19 ms per draw and thread.  With fewer
// than 8 host threads the device batch wins (C5 at threads = 1: 0.60 s -> 0.22 s), with more the host pool does.
// ---------------------------------------------------------------------------------------------------------------------------------
// psi set-up of ONE model with 9 ... 64 states (bssm_psi_smoother, the psi log-likelihood calls, every iteration of a pm / da chain with the
// psi-APF, the host set-up of a post-correction): the host finds the mode and the approximating y / H (psi_setup, host_model.cpp); the
// Kalman smoother and the conditional factors run HERE -- the set-up kernels of the post-correction path on one "draw", with the approximating
// y / H written into their work arrays instead of the Laplace step at a stored mode (UngSetupDev::phases = 2 | 4).  A psi log-likelihood call at
// 13 / 25 / 33 states (T = 200): 17 / 87 / 180 ms -> 7 / 23 / 44 ms; 53 states: 0.58 s of host decompositions -> 14 ms where the guards below let it.
static bool device_factors_one(const bssm::HostModel& h, const std::vector<double>& ay, const std::vector<double>& aH, bssm::PsiSetup* out, bool* finite) {
  if (h.p != 1 || h.m < 9 || h.m > 64 || h.n < 1 || h.n >= 65535 || out->mode.size() != (size_t)h.n) return false;
  // OPT-IN (BSSM_AMD_SINGLE_DEVICE_SETUP=1): the kernels' thin factors are a different (more accurate) algorithm than the host routines' decompositions,
  // and where V_{t-1} is not well conditioned the two leave different rounding-level columns in the factors -- log-likelihoods of seasonal models agree
  // to 1e-9 ... 3e-8, not to the 1e-13 of the post-correction path's comparisons (there both sides start from a stored mode); the default keeps every
  // single-model entry point on the host routines, which are the oracle's algorithm step by step
  if (!bssm::opt().single_device_setup || bssm::opt().device_setup == 0 || bssm::opt().generic_setup || !have_device()) return false;
  bssm::LgPacked L0;
  if (!bssm::ung_pack(h, out->mode.data(), &L0)) return false;
  const int n = h.n, m = h.m, M = L0.m;
  const size_t MM = (size_t)M * M, n1 = (size_t)n + 1, E = L0.buf.size();
  const size_t scratch_d = bssm::ung_setup_scratch_doubles(n, M);
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (double)scratch_d * 8.0 > 0.25 * ((double)free_b + (double)arena().cached)) return false;
  static thread_local bssm::TlsStream st_tls;
  hipStream_t st = st_tls.get(true);
  if (!st) return false;
  if (M < 16 || bssm::opt().factors_full) return false;   // (the checks below read what the thin factor kernel leaves)
  DevBuf din, scratch, okb, flags, cond;
  if (din.alloc(E * sizeof(double)) != hipSuccess || scratch.alloc(scratch_d * sizeof(double)) != hipSuccess || okb.alloc(sizeof(int)) != hipSuccess ||
      flags.alloc(n1 * sizeof(int)) != hipSuccess || cond.alloc(n1 * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); return false; }
  // the head of the work arrays (UngWork, F = 1): approximating y minus x'beta (what the smoother sees) | approximating standard deviations
  std::vector<double> head(2 * (size_t)n);
  for (int t = 0; t < n; t++) { head[t] = ay[t] - h.xbeta[t]; head[(size_t)n + t] = aH[t]; }
  bssm::UngSetupDev b{};
  b.F = 1; b.n = n; b.m = M; b.p = 1; b.mr = m;
  b.dm = ((M >= 32 || (M == 16 && !bssm::opt().setup_rows16)) && !bssm::opt().setup_interleaved) ? 1 : 0;
  b.Zs = L0.Zs; b.Ts = L0.Ts; b.Rs = L0.Rs; b.Ds = L0.Ds; b.Cs = L0.Cs;
  b.oZ = L0.oZ; b.oT = L0.oT; b.oR = L0.oR; b.oD = L0.oD; b.oC = L0.oC; b.oy = L0.oy; b.oa = L0.oa; b.oP = L0.oP;
  b.ou = L0.ou; b.ox = L0.ox; b.om = L0.om; b.oh = L0.oh;
  b.in = din.as<double>();   // (one draw: [E][1] is [1][E], no transposition)
  b.scratch = scratch.as<double>();
  b.ok = okb.as<int>();
  b.flags = (M >= 16 && !bssm::opt().factors_full) ? flags.as<int>() : nullptr;
  b.phases = 2 | 4;
  b.cond = cond.as<double>();
  std::vector<double> at(n1 * M), Vf(n1 * MM), Cf(n1 * MM), ratio(n1, 1.0);
  std::vector<int> flagv(n1, 0);
  int okv = 0;
  // offsets of the work arrays (UngWork): ay, aH, ayr, scl (n each) | at | Pt | cc | K | v | Fs | Vf | Cf
  const size_t o_at = 4 * (size_t)n, o_Vf = o_at + n1 * M + 2 * n1 * MM + (size_t)n * M + 2 * (size_t)n, o_Cf = o_Vf + n1 * MM;
  bool good = hipMemcpyAsync(din.p, L0.buf.data(), E * sizeof(double), hipMemcpyHostToDevice, st) == hipSuccess &&
              hipMemcpyAsync(scratch.p, head.data(), head.size() * sizeof(double), hipMemcpyHostToDevice, st) == hipSuccess &&
              bssm::launch_ung_setup(b, st) == hipSuccess &&
              hipMemcpyAsync(at.data(), scratch.as<double>() + o_at, at.size() * sizeof(double), hipMemcpyDeviceToHost, st) == hipSuccess &&
              hipMemcpyAsync(Vf.data(), scratch.as<double>() + o_Vf, Vf.size() * sizeof(double), hipMemcpyDeviceToHost, st) == hipSuccess &&
              hipMemcpyAsync(Cf.data(), scratch.as<double>() + o_Cf, Cf.size() * sizeof(double), hipMemcpyDeviceToHost, st) == hipSuccess &&
              hipMemcpyAsync(ratio.data() + 1, cond.as<double>() + 1, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st) == hipSuccess &&
              hipMemcpyAsync(flagv.data() + 1, flags.as<int>() + 1, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, st) == hipSuccess &&
              hipMemcpyAsync(&okv, okb.p, sizeof(int), hipMemcpyDeviceToHost, st) == hipSuccess;
  good = (hipStreamSynchronize(st) == hipSuccess) && good;   // (always: the buffers above go back to the arena behind it)
  if (!good) { (void)hipGetLastError(); return false; }
  // The device's numbers are taken only where they are the host routines' numbers to well below the parity bar (host_model.cpp's psi_setup is the
  // oracle's algorithm step by step; the kernels' thin path is not):
  //   * every time point took the thin path (a singular V_{t-1} goes through the round-robin Jacobi, whose null-space choices are its own);
  //   * V_{t-1} is well conditioned -- smallest / first pivot of its factor >= 1e-6: a Cholesky solve and the pseudo-inverse by decomposition agree to
  //     ~ condition number x eps, 1e-10 here (found by scripts/fuzz_parity.py: 40 states with one noise term over 33 steps differed by 9e-7 in the states);
  //   * no two kept singular values of a factor closer than 1e-6 of the largest: inside a repeated eigenvalue's space every program picks its own
  //     basis (a seasonal model shorter than its period keeps its prior variances at t = 0: particle states then differ by O(1e-2) between two
  //     correct decompositions while weights and log-likelihood agree).
  // Otherwise the host routines run as before.
  if (okv != 0) {
    if (bssm::opt().trace) {
      double rmin = 1.0; int nfl = 0;
      for (size_t t = 1; t < n1; t++) { rmin = std::min(rmin, ratio[t]); nfl += flagv[t] != 0; }
      std::fprintf(stderr, "[bssm_amd] psi set-up of one model on the device (m = %d, n = %d): smallest pivot ratio %.3g, %d time points left to the full path\n", m, n, rmin, nfl);
    }
    for (size_t t = 1; t < n1; t++) if (flagv[t] != 0 || !(ratio[t] >= 1e-6)) return false;
    std::vector<double> sv(m);
    for (size_t t = 0; t < n1; t++) {
      for (int c = 0; c < m; c++) { double a = 0.0; for (int i = 0; i < m; i++) { const double v = Vf[t * MM + i + (size_t)M * c]; a += v * v; } sv[c] = a; }
      for (int c = 0; c + 1 < m; c++) if (sv[c] > 0.0 && !(sv[c] - sv[c + 1] >= 1e-6 * sv[0])) return false;
    }
  }
  out->alphahat.assign((size_t)m * n1, 0.0);
  out->Vt.assign((size_t)m * m * n1, 0.0);
  out->Ct.assign((size_t)m * m * n1, 0.0);
  bool fin = okv != 0;
  for (size_t t = 0; t < n1 && fin; t++) {
    for (int i = 0; i < m; i++) { const double v = at[t * M + i]; out->alphahat[i + (size_t)m * t] = v; fin = fin && std::isfinite(v); }
    for (int j = 0; j < m; j++)
      for (int i = 0; i < m; i++) {
        const double v = Vf[t * MM + i + (size_t)M * j], c = Cf[t * MM + i + (size_t)M * j];
        out->Vt[i + (size_t)m * j + (size_t)m * m * t] = v;
        out->Ct[i + (size_t)m * j + (size_t)m * m * t] = c;
        fin = fin && std::isfinite(v) && std::isfinite(c);
      }
  }
  *finite = fin;
  return true;
}
namespace { struct InstallDeviceFactors { InstallDeviceFactors() { bssm::device_factors_hook() = &device_factors_one; } } g_install_device_factors; }

bool device_setup_enabled(int n_threads) {
  if (bssm::opt().device_setup >= 0) return bssm::opt().device_setup != 0;
  return n_threads < 8;
}

// Shared core of the IS post-correction drivers: filter `cnt` prepared draws (tables already built), then
// back-trace / summarise as output_type asks.  const_add[i] is added to the device log-likelihood
// (bootstrap: state-independent constants); weight_i = exp(loglik_i - sub[i]) (src/approx_mcmc.cpp:385, 479).
// tables of one pipeline chunk built on the device (fn_setup_kernel): draws [lo, lo + F) in draw order
struct ChunkTables {
  int lo = 0, F = 0;
  size_t ms = 0, os = 0, is = 0, as = 0;
  DevBuf fpar, ext, modes, y, scratch, mats, obs, init, amats, ok, flags;
  // the set-up kernel runs asynchronously: the host thread that prepared the chunk records `ready` behind it and goes on
  // to the next chunk; the device thread waits for the event before the chunk's filters (finish() below)
  hipEvent_t ready = nullptr;
  StageBuf stage, okst;   // pinned input staging (alive until the copies have run) and the kernel's per-draw ok flags
  DevBuf in_a, in_b;      // ung / mng: packed draws and their transpose
  // (an early error return may drop the last reference while the set-up kernels still run: nothing goes back to the arena under them)
  ~ChunkTables() { if (ready) { (void)hipEventSynchronize(ready); (void)hipEventDestroy(ready); } }
  // device thread: wait for the set-up kernel, mark failed draws, give the work arrays back
  // device thread, in front of the chunk's filters: the filter stream waits for the set-up kernel ON THE DEVICE -- the host goes
  // on to enqueue the filter launches under it (round 5: a host-side wait here cost the 293-draw shard ~0.1 ms of idle device
  // between the two, and the 600 launches of its tail round were enqueued 0.75 ms later than they could be)
  int wait_on(hipStream_t filters) {
    if (!ready) return 0;
    return hipStreamWaitEvent(filters, ready, 0) == hipSuccess ? 0 : -1;
  }
  int finish(std::vector<char>* bad) {
    if (!ready) return 0;
    if (hipEventSynchronize(ready) != hipSuccess) return -1;
    const int* okv = (const int*)okst.p;
    for (int q = 0; q < F; q++) if (!okv[q]) (*bad)[lo + q] = 1;  // non-finite Vt / Ct: the filter returns -Inf (src/model_ssm_nlg.cpp:988-990)
    scratch.release(); modes.release(); ext.release(); ok.release(); flags.release(); in_a.release(); in_b.release();
    (void)hipEventDestroy(ready); ready = nullptr;
    return 0;
  }
};
// Output 2 of a post-correction chunk, finished behind the NEXT chunk's filters: the fused back-trace + summary kernel
// (one workgroup per draw, latency-bound gathers: 12 ms of a 90 ms C5 job) runs on a stream of its own while the filter
// kernels of the following chunk occupy the null stream.  The object owns what that kernel reads and writes -- the chunk's
// history / ancestor pools, the per-draw means and covariances, their pinned host copies -- until drain() has copied the
// results into the caller's arrays; the next chunk therefore filters into fresh pools (the arena hands out cached blocks).
struct PendingSummary {
  DevBuf hist_pool, idx_pool, bcur_pool, d_mean, d_cov;
  StageBuf h_mean, h_cov;
  hipEvent_t done = nullptr;
  std::vector<int> draws;  // draw index of every filter of the batch
  size_t n1m = 0, n1mm = 0;
  ~PendingSummary() {
    if (done) { (void)hipEventSynchronize(done); (void)hipEventDestroy(done); }  // nothing is released under a running kernel
  }
  int drain(double* alphahat_i, double* Vt_i) {
    if (!done) return BSSM_OK;
    HIP_TRY(hipEventSynchronize(done));
    (void)hipEventDestroy(done); done = nullptr;
    const double* hm = (const double*)h_mean.p; const double* hc = (const double*)h_cov.p;
    for (size_t f = 0; f < draws.size(); f++) {
      std::memcpy(alphahat_i + (size_t)draws[f] * n1m, hm + f * n1m, n1m * sizeof(double));
      std::memcpy(Vt_i + (size_t)draws[f] * n1mm, hc + f * n1mm, n1mm * sizeof(double));
    }
    return BSSM_OK;
  }
};

struct IsCore {
  std::shared_ptr<PendingSummary> pending;  // the previous chunk's summary, still running (device thread only)
  std::mutex chunk_mu;
  std::map<int, std::shared_ptr<ChunkTables>> chunk_tabs;  // by first draw of the chunk; taken by the device thread
  std::shared_ptr<ChunkTables> take_chunk(int lo) {
    std::lock_guard<std::mutex> lk(chunk_mu);
    auto it = chunk_tabs.find(lo);
    if (it == chunk_tabs.end()) return nullptr;
    auto p = it->second;
    chunk_tabs.erase(it);
    return p;
  }
  int n = 0, m = 0, k = 0, pp = 1;
  bool psi = false;
  int first = 0, is_type = 2, output_type = 3;
  uint32_t nsim = 0, seed = 0;
  const uint32_t* counts = nullptr;
  std::vector<bssm::FilterTables> tabs;
  std::vector<double> fpar;    // fps per draw, or empty
  size_t fps = bssm::FP_SIZE;  // doubles per draw of fpar: fp_size(padded state dimension), pf_functors.h
  std::vector<double> amats;   // per draw n approximating rows (fn_amats_row doubles each), or empty
  int count_draws = 1;         // draws the amats slots were laid out for
  std::vector<double> const_add, sub;
  std::vector<char> bad;
};

// draws [lo, hi) of the prepared job
int run_is_core(IsCore& J, int lo, int hi, double* weights, double* loglik, double* alpha, double* alphahat_i, double* Vt_i) {
  const int cnt = hi - lo, n = J.n, m = J.m, k = J.k, pp = J.pp;
  const bool psi = J.psi;
  std::vector<bssm::FilterTables>& tabs = J.tabs;
  const std::vector<char>& bad = J.bad;
  struct { int first, is_type, output_type; uint32_t nsim, seed; const uint32_t* counts; } jobv{J.first, J.is_type, J.output_type, J.nsim, J.seed, J.counts};
  const auto* job = &jobv;
  const bool need_hist = job->output_type != 3;

  // group draws by particle count (is1 multiplies nsim by count_i, src/approx_mcmc.cpp:376-378)
  std::vector<int> order(cnt);
  for (int i = 0; i < cnt; i++) order[i] = lo + i;
  auto nsim_of = [&](int i) { return (int)(job->is_type == 1 ? job->nsim * job->counts[i] : job->nsim); };
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return nsim_of(a) < nsim_of(b); });
  bool any_degenerate = false;
  size_t pos = 0;
  const double tcore0 = now_ms();
  const std::shared_ptr<ChunkTables> ctabs = J.take_chunk(lo);  // non-null: the tables of [lo, hi) are already in HBM, in draw order
  if (ctabs && ctabs->wait_on(nullptr)) return fail(BSSM_ERR_HIP, "psi set-up kernel failed");
  (void)tcore0;
  while (pos < order.size()) {
    const int N = nsim_of(order[pos]);
    size_t end = pos;
    while (end < order.size() && nsim_of(order[end]) == N) end++;
    // Batches of draws: the kept history (particle system + ancestor table) lives in pools allocated ONCE per particle
    // count and reused by every batch.  A batch is a multiple of 256 filters (one workgroup per CU per round) within
    // ~24 GiB.  Nothing is back-traced into a second copy: output 1 traces the one trajectory it keeps, output 2 fuses
    // back-trace and summary (m <= 8; larger states fall back to backtrace + summary through a smoothed copy).
    const size_t hist_elems_f = (size_t)(n + 1) * m * N, idx_elems_f = (size_t)n * N;
    const bool fused = m <= 8;
    const size_t per_filter = need_hist ? hist_elems_f * (fused ? 8 : 16) + idx_elems_f * 4 + (size_t)N * 4 : 0;
    size_t maxF = (size_t)1 << 20;
    if (per_filter) {
      maxF = std::max<size_t>(1, ((size_t)24 << 30) / per_filter);
      if (maxF >= 256) maxF = maxF / 256 * 256;
    }
    // deferred summary (PendingSummary): the whole chunk is ONE batch of one particle count with the fused kernel
    // Off unless BSSM_AMD_DEFER_SUMMARY is set: measured on C5 with summaries (2 340 draws) the call takes 81.2 ms either way
    // (filter thread 75.5 vs 78.2 ms) -- a filter workgroup of 1024 threads at 128 VGPRs owns the whole register file of its
    // CU, so the summary's workgroups only find room in the tail of a filter launch (profiles/r03_c5_output2_ab.txt) -- and
    // the deferral holds a second set of history pools.
    const bool defer_on = bssm::opt().defer_summary;
    const bool defer = job->output_type == 2 && fused && defer_on && pos == 0 && end == order.size() && (end - pos) <= maxF;
    std::shared_ptr<PendingSummary> pend = defer ? std::make_shared<PendingSummary>() : nullptr;
    DevBuf hist_local, idx_local, smooth_pool, bcur_local;
    DevBuf& hist_pool = pend ? pend->hist_pool : hist_local;
    DevBuf& idx_pool = pend ? pend->idx_pool : idx_local;
    DevBuf& bcur_pool = pend ? pend->bcur_pool : bcur_local;
    if (need_hist) {
      const size_t Fp = std::min(maxF, end - pos);
      HIP_TRY(hist_pool.alloc(Fp * hist_elems_f * sizeof(double)));
      HIP_TRY(idx_pool.alloc(Fp * idx_elems_f * sizeof(uint32_t)));
      if (!fused && job->output_type == 2) HIP_TRY(smooth_pool.alloc(Fp * hist_elems_f * sizeof(double)));
      if (fused && job->output_type == 2) HIP_TRY(bcur_pool.alloc(Fp * (size_t)N * sizeof(uint32_t)));
    }
    while (pos < end) {
      const double tb0 = now_ms();
      const size_t F = std::min(maxF, end - pos);
      std::vector<const bssm::FilterTables*> bt(F);
      for (size_t f = 0; f < F; f++) bt[f] = &tabs[order[pos + f]];
      DeviceBatch db;
      int rc;
      if (ctabs) {
        const size_t off = pos;  // (uniform particle count: `order` is the identity on this chunk)
        rc = db.attach(ctabs->mats.as<double>() + off * ctabs->ms, ctabs->ms, (int)(ctabs->ms / (size_t)n), ctabs->obs.as<double>() + off * ctabs->os,
                       ctabs->os, ctabs->init.as<double>() + off * ctabs->is, ctabs->is, (int)F, n, m, k, pp, N, psi, bssm::pad_dim(m), bssm::pad_dim(m));
      } else {
        rc = db.upload(bt, n, m, k, pp, N, psi, false, 0, false, 4);
      }
      if (rc) return rc;
      if (need_hist) { db.fb.alpha_hist = hist_pool.as<double>(); db.fb.indices = idx_pool.as<uint32_t>(); db.fb.weights = nullptr; }
      DevBuf fparb, amatb;  // per-draw functor parameters / approximating (T_t, C_t), in batch order
      StageBuf st_fpar, st_amat;
      if (ctabs && ctabs->fpar.p) {
        db.fb.fpar = ctabs->fpar.as<double>() + pos * J.fps; db.fb.fpar_fstride = J.fps;
        db.fb.user_slot = 0;
        db.fb.fn_id = (int)J.fpar[bssm::FP_ID];
        db.fb.amats = ctabs->amats.as<double>() + pos * ctabs->as; db.fb.amats_fstride = ctabs->as;
      } else if (!J.fpar.empty()) {
        HIP_TRY(st_fpar.alloc(F * J.fps * sizeof(double)));
        double* hb = (double*)st_fpar.p;
        for (size_t f = 0; f < F; f++) std::memcpy(hb + f * J.fps, J.fpar.data() + (size_t)order[pos + f] * J.fps, J.fps * sizeof(double));
        HIP_TRY(fparb.alloc(F * J.fps * sizeof(double)));
        HIP_TRY(hipMemcpyAsync(fparb.p, hb, fparb.bytes, hipMemcpyHostToDevice, nullptr));
        db.fb.fpar = fparb.as<double>(); db.fb.fpar_fstride = J.fps;
        db.fb.user_slot = (int)J.fpar[bssm::FP_SLOT];
        db.fb.fn_id = (int)J.fpar[bssm::FP_ID];
      }
      if (!ctabs && !J.amats.empty()) {
        const size_t as = J.amats.size() / (size_t)J.count_draws;  // per-draw slot: n approximating rows (host_model.h: fn_amats_row)
        HIP_TRY(st_amat.alloc(F * as * sizeof(double)));
        double* hb = (double*)st_amat.p;
        for (size_t f = 0; f < F; f++) std::memcpy(hb + f * as, J.amats.data() + (size_t)order[pos + f] * as, as * sizeof(double));
        HIP_TRY(amatb.alloc(F * as * sizeof(double)));
        HIP_TRY(hipMemcpyAsync(amatb.p, hb, amatb.bytes, hipMemcpyHostToDevice, nullptr));
        db.fb.amats = amatb.as<double>(); db.fb.amats_fstride = as;
      }
      const double tb1 = now_ms();
      // Philox stream = global draw index: the draws of a batch are sorted by particle count, so they are launched
      // per run of consecutive draw indices (a single run unless is1 mixes particle counts).
      size_t f0 = 0;
      std::vector<double> ll(F);
      std::deque<DevBuf> tail_ws;  // step-kernel workspaces of the tail filters: alive until the read-back below has waited for them
      while (f0 < F) {
        size_t f1 = f0 + 1;
        while (f1 < F && order[pos + f1] == order[pos + f1 - 1] + 1) f1++;
        bssm::FilterBatch sub = db.fb;
        sub.F = (int)(f1 - f0);
        sub.mats += db.fb.mats_fstride * f0; sub.obs += db.fb.obs_fstride * f0; sub.init += db.fb.init_fstride * f0;
        sub.loglik += f0; sub.dead_at += f0;
        if (sub.fpar) sub.fpar += sub.fpar_fstride * f0;
        if (sub.amats) sub.amats += sub.amats_fstride * f0;
        if (need_hist) {
          sub.alpha_hist += (size_t)f0 * (n + 1) * m * N; sub.indices += (size_t)f0 * n * N;
        }
        sub.seed = job->seed;
        sub.stream0 = (uint32_t)(job->first + order[pos + f0]);
        // every draw through the row family: states of more than 16 dimensions beyond the workgroup-resident capacity (their only step
        // shape), or BSSM_AMD_FORCE_PATH=row (test / measurement hook)
        const bool all_rows = bssm::row_filter_supported(sub) && (bssm::opt().force_path == 3 || (db.use_grid && sub.MP >= 32));
        if (db.use_grid && !all_rows) {
          DevBuf wsb;  // workspace sized for the sub-batch
          int B, chunk;
          HIP_TRY(wsb.alloc(bssm::grid_workspace_bytes(sub, &B, &chunk)));
          bssm::GridWorkspace gw;
          bssm::grid_workspace_carve(sub, wsb.p, &gw);
          HIP_TRY(bssm::launch_grid_filter(sub, gw, nullptr));
          HIP_TRY(hipDeviceSynchronize());
          HIP_TRY(bssm::launch_zero_dead_tail(sub, true, nullptr));
        } else {
          // A particle system that takes more than half a CU's LDS runs one workgroup per CU, so the filters go through the chip in
          // rounds of `cus` workgroups that all take the same time; a sparsely filled LAST round (C5: 2 340 = 9 x 256 + 36; one rank's
          // shard of it at 8 ranks: 293 = 256 + 37) costs a full round.  Those filters go through the ROW family instead (pf_row_step.h:
          // one wave per row of 64 particles, one launch per time step, the whole chip), beside the last full round (below).
          // ON by default since round 5: both families evaluate pf_canon.h, so a draw's result is the same bits whichever ran it
          // (tests/test_gpu_postcorrect.py, tests/test_gpu_fullsize.py assert split == whole bit for bit) -- which draws land in the
          // tail depends on the draws per chunk, the shard layout and the CU count, and the output does not.  (Rounds 3-4 sent the
          // tail through the step kernels, whose reduction shape differs: opt-in, results depended on the device list.)
          // Not when a path is forced, and only for the shapes the row family carries (row_filter_supported).  BSSM_AMD_TAIL_SPLIT=0 turns it off.
          int tail = 0, cus_round = 0;
          bool lean_all = false;   // every full round in the 96-register form, the tail beside all of them (below)
          if (all_rows) tail = sub.F;
          else if ((size_t)(sub.MP + 1) * (size_t)N * sizeof(double) > 80 * 1024 && bssm::opt().force_path == 0 && bssm::opt().tail_split &&
              bssm::row_filter_supported(sub)) {
            int dev = 0; hipDeviceProp_t prop;
            const int cus = hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
            const int r = sub.F % cus, rounds = sub.F / cus;
            if (r > 0 && r <= cus * bssm::opt().tail_max_eighths / 8) { tail = r; cus_round = cus; }
            // Late round 5: a LARGER remainder (up to three quarters of a round) of a chunk of three or four full rounds also goes through the
            // row family -- beside ALL the rounds, which then run the 96-register form: its launch sequence has several rounds to finish in, and
            // the remainder's own round is saved.  C5 shape: 914 = 3 x 256 + 146 draws 16.6 -> 15.7 ms, 1 170 = 4 x 256 + 146: 21.7 -> 20.3,
            // 1 216: 22.3 -> 21.3; with five rounds and more nothing is left of it (1 450: 25.3 -> 25.5; the lean form costs every round ~2.6 %).
            // These calls then sit within 1-3 % of draws / 256 x one round: the chip is packed as well as the kernels' efficiency allows.
            else if (r > 0 && r <= cus * 6 / 8 && rounds >= 3 && rounds <= 4 && bssm::opt().tail_concurrent && bssm::tail_overlap_available() && bssm::opt().tail_max_eighths > 0 &&
                     bssm::wg_lean_form_exists(sub)) { tail = r; cus_round = cus; lean_all = true; }
          }
          bssm::FilterBatch head = sub;
          head.F = sub.F - tail;
          // Round 5, late: with BOTH a full round and a tail, the tail steps BESIDE the last full round instead of behind it -- that round runs
          // the 96-register form of the workgroup kernel where one exists (pf_wg_kernel_lean: same bits), which leaves register file and LDS
          // for the tail's row-family workgroups on every CU.  The tail's launch sequence goes on a stream of its own, forked from and joined
          // to the chunk's stream.  BSSM_AMD_TAIL_CONCURRENT=0: the tail behind the round, as before.
          hipStream_t tstream = nullptr;
          struct EventGuard {   // destroyed on every path out of this block, early error returns included
            hipEvent_t e = nullptr;
            ~EventGuard() { if (e) (void)hipEventDestroy(e); }
          } g_fork, g_join;
          hipEvent_t &ev_fork = g_fork.e, &ev_join = g_join.e;
          bool conc = bssm::opt().tail_concurrent && head.F > 0 && tail > 0 && bssm::tail_overlap_available();
          if (conc) {
            static thread_local bssm::TlsStream tail_stream;   // (at the highest stream priority: measured, no difference)
            tstream = tail_stream.get(false, true);   // (the stream reserved for it at the first entry-point call: options.cpp, prewarm_streams)
            conc = tstream != nullptr && hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) == hipSuccess;
          }
          // filters [f_begin, f_begin + count) of the sub-batch
          auto slice = [&](int f_begin, int count) {
            const size_t h0 = (size_t)f_begin;
            bssm::FilterBatch q = sub;
            q.F = count;
            q.mats += db.fb.mats_fstride * h0; q.obs += db.fb.obs_fstride * h0; q.init += db.fb.init_fstride * h0;
            q.loglik += h0; q.dead_at += h0;
            if (q.fpar) q.fpar += q.fpar_fstride * h0;
            if (q.amats) q.amats += q.amats_fstride * h0;
            if (need_hist) { q.alpha_hist += h0 * (size_t)(n + 1) * m * N; q.indices += h0 * (size_t)n * N; }
            q.stream0 = sub.stream0 + (uint32_t)h0;
            return q;
          };
          auto launch_wg_part = [&](int f_begin, int count, bool lean) -> int {
            if (count <= 0) return 0;
            bssm::FilterBatch q = slice(f_begin, count);
            q.lean = lean ? 1 : 0;
            HIP_TRY(bssm::launch_wg_filter(q, nullptr));
            HIP_TRY(bssm::launch_zero_dead_tail(q, true, nullptr));
            return 0;
          };
          auto launch_tail = [&](hipStream_t st) -> int {
            if (tail <= 0) return 0;
            bssm::FilterBatch tl = slice(head.F, tail);
            tail_ws.emplace_back();
            HIP_TRY(tail_ws.back().alloc(bssm::row_workspace_bytes(tl)));
            bssm::GridWorkspace gw;
            bssm::row_workspace_carve(tl, tail_ws.back().p, &gw);
            HIP_TRY(bssm::launch_grid_filter(tl, gw, st));
            HIP_TRY(bssm::launch_zero_dead_tail(tl, true, st));
            return 0;
          };
          if (conc) {
            // the full rounds before the last one keep the 128-register kernel (the 96-register form spills more: 2 340 draws as ONE lean
            // launch took 41.8 ms against 40.7); the tail is released when they are done and runs beside the last round
            const int last = lean_all ? head.F : (cus_round > 0 && head.F > cus_round ? cus_round : head.F);
            if (int rc2 = launch_wg_part(0, head.F - last, false)) return rc2;
            HIP_TRY(hipEventRecord(ev_fork, nullptr));
            HIP_TRY(hipStreamWaitEvent(tstream, ev_fork, 0));
            // (the round first: enqueueing the tail's ~600 launches takes the host ~0.6 ms, which the round must not wait for)
            if (int rc2 = launch_wg_part(head.F - last, last, true)) return rc2;
            if (int rc2 = launch_tail(tstream)) return rc2;
            HIP_TRY(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(ev_join, tstream));
            HIP_TRY(hipStreamWaitEvent(nullptr, ev_join, 0));
          } else {
            if (int rc2 = launch_wg_part(0, head.F, false)) return rc2;
            if (int rc2 = launch_tail(nullptr)) return rc2;
          }
        }
        f0 = f1;
      }
      // the previous chunk's summary has been running beside these filters: collect it while they finish
      if (J.pending) { const int prc = J.pending->drain(alphahat_i, Vt_i); J.pending.reset(); if (prc) return prc; }
      const double tq = now_ms();
      HIP_TRY(hipMemcpy(ll.data(), db.loglik.p, F * sizeof(double), hipMemcpyDeviceToHost));
      if (ctabs && ctabs->finish(&J.bad)) return fail(BSSM_ERR_HIP, "psi set-up kernel failed");  // (complete: the copy waited for the filters behind it) failed draws, work arrays back
      if (bssm::opt().trace) std::fprintf(stderr, "[bssm_amd] core: launches enqueued after %.3f ms, device done after %.3f ms\n", tq - tb1, now_ms() - tb1);
      const double tb2 = now_ms();
      // smoothing outputs
      DevBuf d_mean, d_cov, d_picks, d_traj;
      std::vector<double> hmean, hcov, htraj;
      if (pend) {  // (the filters are complete: the copy above waited for the null stream)
        static thread_local bssm::TlsStream st_sum_tls;
        hipStream_t ss = st_sum_tls.get();
        if (!ss) return fail(BSSM_ERR_HIP, "no stream for the summary kernel");
        pend->n1m = (size_t)(n + 1) * m; pend->n1mm = pend->n1m * m;
        HIP_TRY(pend->d_mean.alloc(F * pend->n1m * sizeof(double)));
        HIP_TRY(pend->d_cov.alloc(F * pend->n1mm * sizeof(double)));
        HIP_TRY(pend->h_mean.alloc(F * pend->n1m * sizeof(double)));
        HIP_TRY(pend->h_cov.alloc(F * pend->n1mm * sizeof(double)));
        HIP_TRY(bssm::launch_smooth_summary(db.fb.alpha_hist, db.fb.indices, bcur_pool.as<uint32_t>(), pend->d_mean.as<double>(),
                                            pend->d_cov.as<double>(), (int)F, n, m, N, ss));
        HIP_TRY(hipMemcpyAsync(pend->h_mean.p, pend->d_mean.p, F * pend->n1m * sizeof(double), hipMemcpyDeviceToHost, ss));
        HIP_TRY(hipMemcpyAsync(pend->h_cov.p, pend->d_cov.p, F * pend->n1mm * sizeof(double), hipMemcpyDeviceToHost, ss));
        HIP_TRY(hipEventCreateWithFlags(&pend->done, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(pend->done, ss));
        pend->draws.resize(F);
        for (size_t f = 0; f < F; f++) pend->draws[f] = order[pos + f];
        J.pending = pend;
      } else if (job->output_type == 2) {
        HIP_TRY(d_mean.alloc((size_t)F * (n + 1) * m * sizeof(double)));
        HIP_TRY(d_cov.alloc((size_t)F * (n + 1) * m * m * sizeof(double)));
        if (fused) {
          HIP_TRY(bssm::launch_smooth_summary(db.fb.alpha_hist, db.fb.indices, bcur_pool.as<uint32_t>(), d_mean.as<double>(),
                                              d_cov.as<double>(), (int)F, n, m, N, nullptr));
        } else {
          HIP_TRY(bssm::launch_backtrace(db.fb.alpha_hist, db.fb.indices, smooth_pool.as<double>(), (int)F, n, m, N, nullptr));
          HIP_TRY(bssm::launch_summary(smooth_pool.as<double>(), d_mean.as<double>(), d_cov.as<double>(), (int)F, n + 1, m, N, nullptr));
        }
        hmean.resize((size_t)F * (n + 1) * m);
        hcov.resize((size_t)F * (n + 1) * m * m);
        HIP_TRY(hipMemcpy(hmean.data(), d_mean.p, d_mean.bytes, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(hcov.data(), d_cov.p, d_cov.bytes, hipMemcpyDeviceToHost));
      } else if (job->output_type == 1) {  // one uniformly chosen back-traced trajectory per draw (:391-392)
        std::vector<uint32_t> picks(F);
        for (size_t f = 0; f < F; f++) {
          const bssm::u32x4 w = bssm::philox4x32_10(0, 0, 0, bssm::RNG_PICK, job->seed, (uint32_t)(job->first + order[pos + f]));
          const double u = (double)bssm::bits53(w.x, w.y) * 0x1.0p-53;
          int pick = (int)(u * N);
          picks[f] = (uint32_t)(pick >= N ? N - 1 : pick);
        }
        HIP_TRY(d_picks.alloc(F * sizeof(uint32_t)));
        HIP_TRY(d_traj.alloc((size_t)F * (n + 1) * m * sizeof(double)));
        HIP_TRY(hipMemcpy(d_picks.p, picks.data(), d_picks.bytes, hipMemcpyHostToDevice));
        HIP_TRY(bssm::launch_pick_trajectory(db.fb.alpha_hist, db.fb.indices, d_picks.as<uint32_t>(), d_traj.as<double>(), (int)F, n, m, N, nullptr));
        htraj.resize((size_t)F * (n + 1) * m);
        HIP_TRY(hipMemcpy(htraj.data(), d_traj.p, d_traj.bytes, hipMemcpyDeviceToHost));
      }
      for (size_t f = 0; f < F; f++) {
        const int i = order[pos + f];
        double l = ll[f];
        if (bad[i] || !std::isfinite(l)) { l = -INFINITY; any_degenerate = true; }
        const double full = l + J.const_add[i];
        weights[i] = std::exp(full - J.sub[i]);  // src/approx_mcmc.cpp:385 (psi: approx_loglik = 0), :479 (bsf)
        if (loglik) loglik[i] = full;
        if (job->output_type == 1) {  // stored (n+1) x m
          const double* tr = htraj.data() + f * (size_t)(n + 1) * m;
          for (int t = 0; t <= n; t++) for (int c = 0; c < m; c++)
            alpha[t + (size_t)(n + 1) * (c + (size_t)m * i)] = tr[(size_t)t * m + c];
        } else if (job->output_type == 2 && !pend) {
          std::memcpy(alphahat_i + (size_t)i * (n + 1) * m, hmean.data() + f * (size_t)(n + 1) * m, (size_t)(n + 1) * m * sizeof(double));
          std::memcpy(Vt_i + (size_t)i * (n + 1) * m * m, hcov.data() + f * (size_t)(n + 1) * m * m, (size_t)(n + 1) * m * m * sizeof(double));
        }
      }
      if (bssm::opt().trace)
        std::fprintf(stderr, "[bssm_amd] is batch F=%zu N=%d: upload %.2f ms, filter %.2f ms, outputs %.2f ms\n", F, N, tb1 - tb0,
                     tb2 - tb1, now_ms() - tb2);
      pos += F;
    }
  }
  return any_degenerate ? BSSM_WARN_DEGENERATE : BSSM_OK;
}

// Host set-up and device work of the post-correction overlap: the draws are cut into chunks, and while the device
// filters chunk c (on its own host thread, which only waits on HIP) the caller's thread pool prepares chunk c + 1.
// prepare(i) fills draw i's slots of J and returns "" or an error; results do not depend on the chunking because every
// draw has its own Philox stream.
// Optional middle phase: device_chunk(lo, hi) runs once per chunk on the caller's thread between prepare and post
// (the batched psi set-up smoother), post(i) then finishes draw i (table building) on the thread pool.
int run_is_pipeline(IsCore& J, int cnt, int n_threads, const std::function<std::string(int)>& prepare,
                    const std::function<void(int, int)>& finish_chunk, double* weights, double* loglik, double* alpha,
                    double* alphahat_i, double* Vt_i, const std::function<int(int, int)>& device_chunk = nullptr,
                    const std::function<std::string(int)>& post = nullptr, int chunk_hint = 0) {
  if (!have_device()) return no_device();
  int dev = 0;
  (void)hipGetDevice(&dev);
  // Chunk = two rounds of one workgroup per CU, but never fewer than four chunks per call: a shard of a few hundred
  // draws (eight ranks on a 2 340-draw job) would otherwise be ONE chunk and lose the set-up / filter overlap.
  // With the set-up of a chunk as ONE device kernel and next to no per-draw host work (chunk_hint = -1) that rule is
  // wrong: the kernel's latency does not depend on the chunk size (one lane per draw, ~4.6 ms on the C5 shape) and nothing on
  // the host needs hiding, so every chunk boundary only costs: a filter launch cannot start before the previous one has
  // drained, and a launch of fewer workgroups than CUs leaves the rest of the chip idle.  Calls of up to 4 096 draws
  // therefore go as ONE chunk (the dispatcher keeps every CU busy until the last workgroup), longer calls in chunks of 4 096
  // (bounded tables in memory, the next chunk's set-up kernel under this chunk's filters).  Measured: C5 (2 340 draws) 57.1 ->
  // 49.3 ms against chunks of 512; the 8 x job (18 720 draws) 434-470 ms at 512, 395 at 1 024, 359 at 2 048, 340 at 4 096, 338 at
  // 9 360 (profiles/r03_c5_large_chunks.txt).
  // Shards of the C5 job as N ranks would see them (scripts/c5_shard_probe.py, profiles/r03_c5_shard_probe.txt).
  int chunk = 512;
  if (bssm::opt().is_chunk > 0) chunk = bssm::opt().is_chunk;
  else if (chunk_hint > 0) chunk = chunk_hint;
  else if (chunk_hint == -1) {  // (the ssm_nlg path: packing a draw on the host is a memcpy of theta)
    hipDeviceProp_t prop;
    const int cus = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    (void)cus;
    chunk = std::min(cnt, 4096);
    if (bssm::opt().is_big_chunk > 0 && cnt > 4096) chunk = std::max(64, bssm::opt().is_big_chunk);  // measurement option (-DBSSM_PROBES builds)
  }
  else if (cnt < 4 * chunk) chunk = std::max(64, ((cnt + 3) / 4 + 63) / 64 * 64);
  std::vector<std::string> errs(cnt);
  // ONE device thread for the whole call (it only waits on HIP): chunks are handed over through a queue, so its
  // thread-local streams are made once, and chunk c + 1 is prepared while chunk c filters.
  std::mutex mu;
  std::condition_variable cv;
  std::vector<std::pair<int, int>> queue;
  size_t taken = 0;
  bool closing = false;
  int dev_rc = BSSM_OK, worst = BSSM_OK;
  std::string dev_err;
  double t_host = 0, t_dev = 0, t_pack = 0, t_chunk = 0;
  const double t_call0 = now_ms();
  std::thread worker([&]() {
    (void)hipSetDevice(dev);  // the current device is per host thread
    for (;;) {
      std::pair<int, int> job;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return taken < queue.size() || closing; });
        if (taken >= queue.size()) return;
        job = queue[taken++];
      }
      cv.notify_all();  // (the preparing thread may be waiting for room in the queue)
      if (dev_rc < 0) continue;  // drain
      const double t1 = now_ms();
      const int rc = run_is_core(J, job.first, job.second, weights, loglik, alpha, alphahat_i, Vt_i);
      if (bssm::opt().trace) std::fprintf(stderr, "[bssm_amd] device thread: chunk [%d, %d) taken at %.3f ms, returned at %.3f ms of the call\n", job.first, job.second, t1 - t_call0, now_ms() - t_call0);
      if (rc < 0) { dev_err = g_err; dev_rc = rc; }
      else if (rc == BSSM_WARN_DEGENERATE) worst = BSSM_WARN_DEGENERATE;
      t_dev += now_ms() - t1;
    }
  });
  auto drain_pending = [&]() -> int {  // the last chunk's summary (after the device thread has gone)
    if (!J.pending) return BSSM_OK;
    const int rc = J.pending->drain(alphahat_i, Vt_i);
    J.pending.reset();
    return rc;
  };
  auto close = [&]() {
    { std::lock_guard<std::mutex> lk(mu); closing = true; }
    cv.notify_all();
    worker.join();
  };
  for (int lo = 0; lo < cnt; lo += chunk) {
    const int hi = std::min(cnt, lo + chunk);
    const double t0 = now_ms();
    // (chunk_hint = -1: packing a draw is ~0.4 us of copying, starting a host thread ~25 us -- one thread per 1 024 draws; with 32
    // threads a 292-draw shard spent 0.8 of its 10 ms starting them)
    const int prep_threads = chunk_hint == -1 ? std::max(1, std::min(n_threads, (hi - lo + 1023) / 1024)) : n_threads;
    parallel_for(hi - lo, prep_threads, [&](int q) { errs[lo + q] = prepare(lo + q); });
    for (int i = lo; i < hi; i++) if (!errs[i].empty()) { close(); return fail(BSSM_ERR_INVALID, "draw " + std::to_string(i) + ": " + errs[i]); }
    const double tc0 = now_ms();
    if (device_chunk) {
      const int rc = device_chunk(lo, hi);
      if (rc < 0) { close(); return rc; }
    }
    t_chunk += now_ms() - tc0;
    if (post) {
      parallel_for(hi - lo, n_threads, [&](int q) { errs[lo + q] = post(lo + q); });
      for (int i = lo; i < hi; i++) if (!errs[i].empty()) { close(); return fail(BSSM_ERR_INVALID, "draw " + std::to_string(i) + ": " + errs[i]); }
    }
    if (finish_chunk) finish_chunk(lo, hi);
    t_host += now_ms() - t0;
    { std::lock_guard<std::mutex> lk(mu); queue.emplace_back(lo, hi); }
    cv.notify_all();
    // at most two prepared chunks wait behind the one being filtered: the set-up kernels run asynchronously now, and their
    // work arrays are only given back by the device thread
    { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return queue.size() - taken <= 2; }); }
  }
  const double tclose0 = now_ms();
  close();
  if (bssm::opt().trace) std::fprintf(stderr, "[bssm_amd] pipeline: chunks handed over after %.3f ms, device thread joined after %.3f ms\n", tclose0 - t_call0, now_ms() - t_call0);
  if (dev_rc < 0) { J.pending.reset(); return fail(dev_rc, dev_err); }
  { const double td0 = now_ms(); const int rc = drain_pending(); if (rc) return rc; t_dev += now_ms() - td0; }
  g_is_timing[0] = t_host; g_is_timing[1] = t_dev;
  t_pack = t_host - t_chunk;
  g_is_timing4[0] = t_pack; g_is_timing4[1] = t_chunk; g_is_timing4[2] = t_dev; g_is_timing4[3] = now_ms() - t_call0;
  return worst;
}
}  // namespace

int bssm_is_correction(const bssm_is_job* job, double* weights, double* loglik, double* alpha, double* alphahat_i,
                       double* Vt_i) {
  if (!job || !weights) return fail(BSSM_ERR_INVALID, "job/weights NULL");
  const int cnt = job->count;
  if (cnt <= 0) return BSSM_OK;
  if (job->sampling_method != BSSM_PSI && job->sampling_method != BSSM_BSF)
    return fail(BSSM_ERR_UNSUPPORTED, "sampling_method must be psi or bsf");
  if (job->output_type < 1 || job->output_type > 3) return fail(BSSM_ERR_INVALID, "output_type must be 1..3");
  if (job->output_type == 1 && !alpha) return fail(BSSM_ERR_INVALID, "alpha buffer required for output_type 1");
  if (job->output_type == 2 && (!alphahat_i || !Vt_i)) return fail(BSSM_ERR_INVALID, "summary buffers required for output_type 2");
  const bool psi = job->sampling_method == BSSM_PSI;
  if (psi && !job->modes) return fail(BSSM_ERR_INVALID, "modes required for psi");
  if (!psi && !job->approx_loglik) return fail(BSSM_ERR_INVALID, "approx_loglik required for bsf");
  if (!job->counts) return fail(BSSM_ERR_INVALID, "counts NULL");
  if (!job->models) return fail(BSSM_ERR_INVALID, "models NULL");
  if (job->nsim < 1) return fail(BSSM_ERR_INVALID, "nsim must be positive");
  if (job->is_type < 1 || job->is_type > 3) return fail(BSSM_ERR_INVALID, "is_type must be 1..3");
  if (job->first < 0) return fail(BSSM_ERR_INVALID, "first must not be negative");

  // per-draw host set-up (the reference does this inside its OpenMP region, src/approx_mcmc.cpp:372-373), overlapped
  // with the device work of the previous chunk of draws
  std::vector<bssm::HostModel> hms(cnt);
  const int n = job->models[0].n, m = job->models[0].m, k = job->models[0].k, p0 = job->models[0].p > 1 ? job->models[0].p : 1;
  const int MP = bssm::pad_dim(m), KP = psi ? MP : bssm::pad_noise_dim(MP, k);
  IsCore J;
  J.n = n; J.m = m; J.k = k; J.pp = p0; J.psi = psi;
  J.first = job->first; J.is_type = job->is_type; J.output_type = job->output_type; J.nsim = job->nsim; J.seed = job->seed;
  J.counts = job->counts;
  J.tabs.resize(cnt);
  J.bad.assign(cnt, 0);
  J.const_add.assign(cnt, 0.0); J.sub.assign(cnt, 0.0);
  // ---- SURVEY.md 8f rank 1, all of it on the device (univariate observations, m <= 8, uniform particle count): the host
  // ingests and packs the draw; Laplace step at the stored mode, scales, smoother, conditional factors and the filter's
  // tables are ONE kernel per pipeline chunk (ung_setup_kernel), whose outputs the filter reads in place.
  // BSSM_AMD_DEVICE_SETUP = 0 / 1 select the host set-up / the device smoother alone (tested against each other and the oracle).
  // ssm_mng draws (p0 > 1 independent series per time point): the same through mng_setup_kernel + mng_factors_kernel +
  // mng_tables_kernel; the default for any number of host threads since late round 3 (512 draws of the C4 shape: 0.069 s against
  // 0.109 s with 32 host threads; until the factors left the sequential kernel it was 0.26 s and only the default below 8 threads).
  // (n < 65535: the set-up's parallel kernels carry the time point in gridDim.y; longer series take the host set-up)
  // (univariate observations: 9 ... 16 states go padded to 16 -- ung_pack, round 5)
  // (17 ... 64 states go padded to 32 / 64: the smoother with its matrices in LDS, lg_smoother_wave.h)
  // (ssm_mng draws of 9 ... 64 states too since round 6: mng_pack pads them the same way)
  const int m_dev = m > 8 ? (m <= 16 ? 16 : (m <= 32 ? 32 : 64)) : m;   // the state dimension the set-up kernels are instantiated for
  bool dev_full = psi && have_device() && m <= 64 && n < 65535 && job->is_type != 1 && (p0 == 1 ? job->models[0].distribution != BSSM_GAUSSIAN : true);
  if (bssm::opt().device_setup >= 0) dev_full = dev_full && bssm::opt().device_setup == 2;
  if (bssm::opt().generic_setup) dev_full = false;
  if (dev_full && m_dev >= 32) {
    // the work arrays of a padded big state are 4 (n + 1) M^2 doubles per draw (13 MB at M = 64, T = 100) and a chunk is at least 64 draws: a series so
    // long that 64 draws do not fit a quarter of what the device has free (up to three chunks are alive at once) takes the host set-up instead of failing with hipErrorOutOfMemory
    size_t free_b = 0, total_b = 0;
    const double per_draw = (double)(p0 > 1 ? bssm::mng_setup_scratch_doubles(n, m_dev, p0) : bssm::ung_setup_scratch_doubles(n, m_dev)) * sizeof(double);
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && per_draw * 64.0 > 0.25 * ((double)free_b + (double)arena().cached)) dev_full = false;
  }
  if (dev_full) {
    std::vector<bssm::LgPacked> packs(cnt);
    auto prepare_full = [&](int i) -> std::string {
      std::string err = bssm::ingest_model(&job->models[i], false, &hms[i]);
      if (!err.empty()) return err;
      if (hms[i].n != n || hms[i].m != m || hms[i].k != k || hms[i].p != p0) return "all draws must share n, m, k, p";
      const double* mode_i = job->modes + (size_t)n * p0 * i;
      if (!(p0 > 1 ? bssm::mng_pack(hms[i], mode_i, &packs[i]) : bssm::ung_pack(hms[i], mode_i, &packs[i])))
        return "draw cannot be packed for the device set-up";
      return "";
    };
    auto device_chunk_full = [&](int lo, int hi) -> int {
      const int F = hi - lo;
      const bssm::LgPacked& L0 = packs[lo];
      bool same = true;
      for (int i = lo + 1; i < hi; i++) same = same && L0.same_layout(packs[i]);
      if (!same) {  // draws that differ in what varies with time: the host set-up for this chunk (tables uploaded by the filter stage)
        parallel_for(F, job->n_threads, [&](int q) {
          const int i = lo + q;
          bssm::PsiSetup su;
          if (!bssm::psi_setup(hms[i], job->modes + (size_t)n * p0 * i, &su)) J.bad[i] = 1;
          bssm::build_psi_tables(hms[i], su, MP, &J.tabs[i]);
          hms[i] = bssm::HostModel(); packs[i] = bssm::LgPacked();
        });
        return 0;
      }
      static thread_local bssm::TlsStream st_tls;
      hipStream_t st = st_tls.get(true);
      if (!st) return fail(BSSM_ERR_HIP, "no stream for the set-up kernel");
      const size_t E = L0.buf.size();
      auto ct = std::make_shared<ChunkTables>();
      ct->lo = lo; ct->F = F;
      ct->ms = (size_t)n * (2 * MP * MP + MP); ct->os = (size_t)n * (bssm::ROW_HDR + (size_t)p0 * (bssm::SER_HDR + MP)); ct->is = (size_t)MP + (size_t)MP * MP; ct->as = 0;
      HIP_TRY(ct->stage.alloc(E * F * sizeof(double)));
      HIP_TRY(ct->okst.alloc((size_t)F * sizeof(int)));
      double* h = (double*)ct->stage.p;
      parallel_for(F, std::min(job->n_threads, 8), [&](int q) {
        std::memcpy(h + E * q, packs[lo + q].buf.data(), E * sizeof(double));
        hms[lo + q] = bssm::HostModel(); packs[lo + q].buf = std::vector<double>();
      });
      DevBuf& din = ct->in_a; DevBuf& din_t = ct->in_b;
      HIP_TRY(din.alloc(E * F * sizeof(double)));
      HIP_TRY(din_t.alloc(E * F * sizeof(double)));
      HIP_TRY(ct->scratch.alloc((p0 > 1 ? bssm::mng_setup_scratch_doubles(n, m_dev, p0) : bssm::ung_setup_scratch_doubles(n, m_dev)) * F * sizeof(double)));
      HIP_TRY(ct->mats.alloc(ct->ms * F * sizeof(double))); HIP_TRY(ct->obs.alloc(ct->os * F * sizeof(double)));
      HIP_TRY(ct->init.alloc(ct->is * F * sizeof(double)));
      HIP_TRY(ct->ok.alloc((size_t)F * sizeof(int)));
      if (m_dev >= 16) HIP_TRY(ct->flags.alloc((size_t)F * (n + 1) * sizeof(int)));   // (16 and more padded states: thin factor kernel -> full factor kernel)
      const double t0 = now_ms();
      HIP_TRY(hipMemcpyAsync(din.p, h, E * F * sizeof(double), hipMemcpyHostToDevice, st));
      HIP_TRY(bssm::launch_transpose(din.as<double>(), din_t.as<double>(), F, (int)E, st));  // [F][E] -> [E][F]
      bssm::UngSetupDev b{};
      b.F = F; b.n = n; b.m = m_dev; b.p = p0; b.mr = m;
      b.dm = ((m_dev >= 32 || (m_dev == 16 && !bssm::opt().setup_rows16)) && !bssm::opt().setup_interleaved) ? 1 : 0;   // draw-major matrix work arrays for the LDS-resident set-up kernels (round 6; the switch: A/B)
      b.Zs = L0.Zs; b.Ts = L0.Ts; b.Rs = L0.Rs; b.Ds = L0.Ds; b.Cs = L0.Cs;
      b.oZ = L0.oZ; b.oT = L0.oT; b.oR = L0.oR; b.oD = L0.oD; b.oC = L0.oC; b.oy = L0.oy; b.oa = L0.oa; b.oP = L0.oP;
      b.ou = L0.ou; b.ox = L0.ox; b.om = L0.om; b.oh = L0.oh;
      b.in = din_t.as<double>(); b.scratch = ct->scratch.as<double>();
      b.mats = ct->mats.as<double>(); b.obs = ct->obs.as<double>(); b.init = ct->init.as<double>(); b.ok = ct->ok.as<int>();
      b.flags = (m_dev >= 16 && !bssm::opt().factors_full) ? ct->flags.as<int>() : nullptr;
      HIP_TRY(p0 > 1 ? bssm::launch_mng_setup(b, st) : bssm::launch_ung_setup(b, st));
      HIP_TRY(hipMemcpyAsync(ct->okst.p, ct->ok.p, (size_t)F * sizeof(int), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipEventCreateWithFlags(&ct->ready, hipEventDisableTiming));
      HIP_TRY(hipEventRecord(ct->ready, st));
      if (cnt > 4096 || bssm::opt().trace) HIP_TRY(hipStreamSynchronize(st));  // (see bssm_is_correction_nlg)
      if (bssm::opt().trace)
        std::fprintf(stderr, "[bssm_amd] psi set-up on device (Laplace step + smoother + tables): F=%d n=%d m=%d: %.2f ms\n", F, n, m, now_ms() - t0);
      std::lock_guard<std::mutex> lk(J.chunk_mu);
      J.chunk_tabs[lo] = ct;
      return 0;
    };
    // (and bounded by the work arrays: 16 GB per chunk -- up to three chunks are alive at once; the C4 shape at T = 2000 takes 17.6 MB per draw)
    // Since the set-up became four kernels whose only sequential one takes the same time for 64 draws as for 2 048, and the host's share
    // is ingesting and packing a draw, EVERY job of this path goes in chunks of up to 2 048 draws (512 weekly-seasonal bsm_ng draws:
    // 21.4 ms in four chunks, 12.5 ms in one; 2 048 draws 46.9 -> 38.3 ms; two-state draws unchanged)
    int hint = 0;
    {
      const double per_draw = (double)(p0 > 1 ? bssm::mng_setup_scratch_doubles(n, m_dev, p0) : bssm::ung_setup_scratch_doubles(n, m_dev)) * sizeof(double);
      // the budget of a chunk's work arrays: 16 GB, and never more than a twelfth of what the device has free right now plus
      // what this library's arena could hand back (up to three chunks alive at once, beside their staging blocks, tables and
      // the filters' history pools of up to 24 GiB) -- a smaller or shared GPU gets smaller chunks instead of hipErrorOutOfMemory
      size_t free_b = 0, total_b = 0;
      double budget = 16e9;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = std::min(budget, ((double)free_b + (double)arena().cached) / 12.0);
      const int by_mem = (int)std::max(64.0, std::floor(budget / per_draw / 64.0) * 64.0);
      hint = std::min(cnt, std::min(2048, by_mem));
    }
    return run_is_pipeline(J, cnt, job->n_threads, prepare_full, nullptr, weights, loglik, alpha, alphahat_i, Vt_i, device_chunk_full, nullptr, hint);
  }
  const bool dev_setup = psi && device_setup_enabled(job->n_threads) && have_device();
  std::vector<bssm::PsiSetup> sus(dev_setup ? cnt : 0);
  std::vector<bssm::LgPacked> lgs(dev_setup ? cnt : 0);
  std::vector<char> pending(cnt, 0);
  auto prepare = [&](int i) -> std::string {
    std::string err = bssm::ingest_model(&job->models[i], false, &hms[i]);
    if (!err.empty()) return err;
    if (hms[i].n != n || hms[i].m != m || hms[i].k != k || hms[i].p != p0) return "all draws must share n, m, k, p";
    if (psi) {
      const double* mode_i = job->modes + (size_t)n * hms[i].p * i;
      if (dev_setup && bssm::psi_setup_begin(hms[i], mode_i, &sus[i], &lgs[i])) { pending[i] = 1; return ""; }
      bssm::PsiSetup su;
      if (!bssm::psi_setup(hms[i], mode_i, &su)) J.bad[i] = 1;
      bssm::build_psi_tables(hms[i], su, MP, &J.tabs[i]);
    } else {
      bssm::build_bsf_tables(hms[i], MP, KP, &J.tabs[i]);
      J.const_add[i] = bssm::const_term(hms[i]);
      J.sub[i] = job->approx_loglik[i];
    }
    hms[i] = bssm::HostModel();  // the tables carry everything the device needs
    return "";
  };
  // the O(n m^3) smoothers of the chunk as one device batch, then the tables on the thread pool
  auto device_chunk = [&](int lo, int hi) -> int {
    std::vector<const bssm::LgPacked*> lp;
    std::vector<std::array<double*, 3>> outs;
    std::vector<int> who;
    for (int i = lo; i < hi; i++) if (pending[i]) {
      lp.push_back(&lgs[i]); who.push_back(i);
      outs.push_back({sus[i].alphahat.data(), sus[i].Vt.data(), sus[i].Ct.data()});
    }
    std::vector<char> okv;
    const int rc = device_lg_batch(lp, outs, &okv);
    if (rc < 0) return rc;
    if (rc == 1) {  // mixed layouts: the same smoother on the host
      parallel_for((int)who.size(), job->n_threads, [&](int q) {
        const int i = who[q];
        if (!bssm::lg_packed_run(lgs[i], sus[i].alphahat.data(), sus[i].Vt.data(), sus[i].Ct.data())) J.bad[i] = 1;
      });
    } else {
      for (size_t q = 0; q < who.size(); q++) if (!okv[q]) J.bad[who[q]] = 1;
    }
    return 0;
  };
  auto post = [&](int i) -> std::string {
    if (!pending[i]) return "";
    bssm::psi_setup_compact(lgs[i], &sus[i]);  // multivariate observations: the smoother ran one series at a time
    bssm::build_psi_tables(hms[i], sus[i], MP, &J.tabs[i]);
    hms[i] = bssm::HostModel(); sus[i] = bssm::PsiSetup(); lgs[i] = bssm::LgPacked();
    pending[i] = 0;
    return "";
  };
  auto finish_chunk = [&](int lo, int hi) {
    if (psi) return;
    // a device batch shares one table layout: time-invariant draws replicate to n rows if any draw of the chunk is time-varying
    int ts = 0;
    for (int i = lo; i < hi; i++) ts = std::max(ts, J.tabs[i].mats_tstride);
    for (int i = lo; i < hi; i++) {
      bssm::FilterTables& t = J.tabs[i];
      if (t.mats_tstride == ts) continue;
      std::vector<double> rep((size_t)n * ts);
      for (int tt = 0; tt < n; tt++) std::memcpy(rep.data() + (size_t)tt * ts, t.mats.data(), ts * sizeof(double));
      t.mats.swap(rep);
      t.mats_tstride = ts;
    }
  };
  if (dev_setup) return run_is_pipeline(J, cnt, job->n_threads, prepare, finish_chunk, weights, loglik, alpha, alphahat_i, Vt_i, device_chunk, post);
  return run_is_pipeline(J, cnt, job->n_threads, prepare, finish_chunk, weights, loglik, alpha, alphahat_i, Vt_i);
}

// postcorrection_nonlinear (src/R_postcorrection.cpp:252-317) / is_correction_psi<ssm_nlg>, is_correction_bsf<ssm_nlg>
int bssm_is_correction_nlg(const bssm_fn_model* models, int32_t first, int32_t count, const double* modes,
                           const double* approx_loglik, const uint32_t* counts, int32_t sampling_method, int32_t is_type,
                           int32_t output_type, uint32_t nsim, uint32_t seed, int32_t n_threads, double* weights,
                           double* loglik, double* alpha, double* alphahat_i, double* Vt_i) {
  if (!models || !weights || !counts) return fail(BSSM_ERR_INVALID, "models/weights/counts NULL");
  if (count <= 0) return BSSM_OK;
  if (sampling_method != BSSM_PSI && sampling_method != BSSM_BSF) return fail(BSSM_ERR_UNSUPPORTED, "sampling_method must be psi or bsf");
  if (output_type < 1 || output_type > 3) return fail(BSSM_ERR_INVALID, "output_type must be 1..3");
  if (output_type == 1 && !alpha) return fail(BSSM_ERR_INVALID, "alpha buffer required for output_type 1");
  if (output_type == 2 && (!alphahat_i || !Vt_i)) return fail(BSSM_ERR_INVALID, "summary buffers required for output_type 2");
  const bool psi = sampling_method == BSSM_PSI;
  if (psi && !modes) return fail(BSSM_ERR_INVALID, "modes required for psi");
  if (!psi && !approx_loglik) return fail(BSSM_ERR_INVALID, "approx_loglik required for bsf");
  if (nsim < 1) return fail(BSSM_ERR_INVALID, "nsim must be positive");
  if (is_type < 1 || is_type > 3) return fail(BSSM_ERR_INVALID, "is_type must be 1..3");
  if (first < 0) return fail(BSSM_ERR_INVALID, "first must not be negative");
  const int n = models[0].n, id = models[0].model_id, m = id == bssm::FN_NLG_GROWTH ? 2 : (id == bssm::FN_USER_NLG ? models[0].n_states : 1);
  IsCore J;
  J.n = n; J.m = m; J.k = m; J.pp = 1; J.psi = psi;
  J.first = first; J.is_type = is_type; J.output_type = output_type; J.nsim = nsim; J.seed = seed; J.counts = counts;
  J.tabs.resize(count);
  J.bad.assign(count, 0);
  J.const_add.assign(count, 0.0); J.sub.assign(count, 0.0);
  J.fps = (size_t)bssm::fp_size(bssm::pad_dim(m)) + bssm::fp_tail(models[0].n_theta, models[0].n_known);
  for (int i = 1; i < count; i++)
    if (models[i].n_theta != models[0].n_theta || models[i].n_known != models[0].n_known) return fail(BSSM_ERR_INVALID, "every draw must carry the same number of theta / known_params");
  J.fpar.assign((size_t)count * J.fps, 0.0);
  // ---- SURVEY.md 8f rank 1, all of it on the device (built-in models, uniform particle count): per draw the host packs
  // theta and (a1, P1); linearisation at the stored mode, smoother, conditional factors and the filter's tables are ONE
  // kernel per pipeline chunk (fn_setup_kernel), whose outputs the filter reads in place.  BSSM_AMD_DEVICE_SETUP = 0 / 1
  // select the host set-up / the device smoother alone instead (they are tested against each other and the oracle).
  bool dev_full = psi && have_device() && id >= bssm::FN_NLG_GROWTH && id <= bssm::FN_NLG_LINEAR_GAUSSIAN && is_type != 1 && n < 65535;  // (time points ride in gridDim.y)
  if (bssm::opt().device_setup >= 0) dev_full = dev_full && bssm::opt().device_setup == 2;
  for (int i = 1; i < count && dev_full; i++) if (models[i].y != models[0].y || models[i].n != n || models[i].model_id != id) dev_full = false;
  if (dev_full) {
    const int mm = m * m, EXT = m + mm;
    std::vector<double> ext((size_t)count * EXT);
    auto prepare = [&](int i) -> std::string {
      static thread_local std::vector<double> fp;  // (its capacity is reused from draw to draw)
      static thread_local bssm::FilterTables none;
      double ap[6];
      int mm_ = 1, kk_ = 1;
      if (functor_tables(&models[i], &none, &fp, &mm_, &kk_, ap)) return g_err;
      std::memcpy(J.fpar.data() + (size_t)i * J.fps, fp.data(), J.fps * sizeof(double));
      double* e = ext.data() + (size_t)i * EXT;
      for (int j = 0; j < m; j++) e[j] = ap[j];
      if (m == 1) e[1] = ap[2]; else for (int q = 0; q < 4; q++) e[2 + q] = ap[2 + q];
      return "";
    };
    auto device_chunk = [&](int lo, int hi) -> int {
      static thread_local bssm::TlsStream st_tls;
      hipStream_t st = st_tls.get(true);
      if (!st) return fail(BSSM_ERR_HIP, "no stream for the set-up kernel");
      const int F = hi - lo;
      auto ct = std::make_shared<ChunkTables>();
      ct->lo = lo; ct->F = F;
      ct->ms = (size_t)n * (2 * mm + m); ct->os = (size_t)n * (bssm::ROW_HDR + bssm::SER_HDR + m); ct->is = (size_t)m + mm; ct->as = (size_t)n * (mm + m);
      const size_t nf = (size_t)F * J.fps, ne = (size_t)F * EXT, nm = (size_t)F * m * n;
      HIP_TRY(ct->stage.alloc((nf + ne + nm + n) * sizeof(double)));
      HIP_TRY(ct->okst.alloc((size_t)F * sizeof(int)));
      double* h = (double*)ct->stage.p;
      std::memcpy(h, J.fpar.data() + (size_t)lo * J.fps, nf * sizeof(double));
      std::memcpy(h + nf, ext.data() + (size_t)lo * EXT, ne * sizeof(double));
      std::memcpy(h + nf + ne, modes + (size_t)m * n * lo, nm * sizeof(double));
      std::memcpy(h + nf + ne + nm, models[0].y, (size_t)n * sizeof(double));
      HIP_TRY(ct->fpar.alloc(nf * sizeof(double))); HIP_TRY(ct->ext.alloc(ne * sizeof(double)));
      HIP_TRY(ct->modes.alloc(nm * sizeof(double))); HIP_TRY(ct->y.alloc((size_t)n * sizeof(double)));
      HIP_TRY(ct->scratch.alloc(bssm::fn_setup_scratch_doubles(n, m) * (((size_t)F + 63) / 64 * 64) * sizeof(double)));  // (blocks of 64 draws: FnWork)
      HIP_TRY(ct->mats.alloc(ct->ms * F * sizeof(double))); HIP_TRY(ct->obs.alloc(ct->os * F * sizeof(double)));
      HIP_TRY(ct->init.alloc(ct->is * F * sizeof(double))); HIP_TRY(ct->amats.alloc(ct->as * F * sizeof(double)));
      HIP_TRY(ct->ok.alloc((size_t)F * sizeof(int)));
      HIP_TRY(hipMemcpyAsync(ct->fpar.p, h, nf * sizeof(double), hipMemcpyHostToDevice, st));
      HIP_TRY(hipMemcpyAsync(ct->ext.p, h + nf, ne * sizeof(double), hipMemcpyHostToDevice, st));
      HIP_TRY(hipMemcpyAsync(ct->modes.p, h + nf + ne, nm * sizeof(double), hipMemcpyHostToDevice, st));
      HIP_TRY(hipMemcpyAsync(ct->y.p, h + nf + ne + nm, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
      bssm::FnSetupDev b{};
      b.F = F; b.n = n; b.m = m;
      b.fpar = ct->fpar.as<double>(); b.ext = ct->ext.as<double>(); b.modes = ct->modes.as<double>(); b.y = ct->y.as<double>();
      b.scratch = ct->scratch.as<double>();
      b.mats = ct->mats.as<double>(); b.obs = ct->obs.as<double>(); b.init = ct->init.as<double>(); b.amats = ct->amats.as<double>();
      b.ok = ct->ok.as<int>();
      const double t0 = now_ms();
      HIP_TRY(bssm::launch_fn_setup(b, st));
      HIP_TRY(hipMemcpyAsync(ct->okst.p, ct->ok.p, (size_t)F * sizeof(int), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipEventCreateWithFlags(&ct->ready, hipEventDisableTiming));
      HIP_TRY(hipEventRecord(ct->ready, st));
      // Hand-over without waiting only for short jobs (up to 8 pipeline chunks), where the set-up latency in front of the first
      // filters is a visible share of the call (C5, 2 340 draws: 0.071 -> 0.068 s, one host thread 0.100 -> 0.086 s); on
      // long jobs the preparing thread waiting here paces the set-up kernels into the gaps between filter rounds, which
      // measured 12 % faster than letting them run ahead (18 720 draws: 0.49 s against 0.55 s).
      if (count > 4096 || bssm::opt().trace >= 2) HIP_TRY(hipStreamSynchronize(st));
      if (bssm::opt().trace >= 2)
        std::fprintf(stderr, "[bssm_amd] psi set-up on device (linearisation + smoother + tables): F=%d n=%d m=%d: %.2f ms\n", F, n, m, now_ms() - t0);
      std::lock_guard<std::mutex> lk(J.chunk_mu);
      J.chunk_tabs[lo] = ct;
      return 0;
    };
    return run_is_pipeline(J, count, n_threads, prepare, nullptr, weights, loglik, alpha, alphahat_i, Vt_i, device_chunk, nullptr, -1);
  }
  // (a user snippet with a state-dependent R_fn carries R_t R_t' in its approximating rows: fn_amats_row)
  const bool user_rfn = models[0].model_id == bssm::FN_USER_NLG && models[0].user_slot >= 1 && models[0].user_slot <= 63 && g_user_host[models[0].user_slot].R;
  const size_t am_stride = (size_t)n * (size_t)bssm::fn_amats_row(bssm::pad_dim(m), user_rfn);  // per-draw slot of J.amats
  if (psi) { J.amats.assign((size_t)count * am_stride, 0.0); J.count_draws = count; }
  const bool dev_setup = psi && device_setup_enabled(n_threads) && have_device() && m <= 8;   // (lg_setup_kernel: up to 8 states; 9 ... 16: the host set-up)
  std::vector<bssm::FnModel> fms(dev_setup ? count : 0);
  std::vector<bssm::FnSetup> fsus(dev_setup ? count : 0);
  std::vector<bssm::LgPacked> lgs(dev_setup ? count : 0);
  std::vector<char> pending(count, 0);
  // non-finite Vt/Ct: the filter returns -Inf (src/model_ssm_nlg.cpp:988-990); run a harmless bootstrap table in its place
  auto bad_draw = [&](int i, std::vector<double>* fp, std::vector<double>* am) -> bool {
    int mm_ = 1, kk_ = 1;
    J.bad[i] = 1;
    if (functor_tables(&models[i], &J.tabs[i], fp, &mm_, &kk_)) return false;
    // same padding and time-variation as the good draws' tables (build_fn_psi_tables): a batch shares both
    const size_t mp = (size_t)bssm::pad_dim(m);
    J.tabs[i].MP = (int)mp; J.tabs[i].KP = (int)mp;
    J.tabs[i].mats.assign((size_t)n * (mp * mp * 2 + mp), 0.0); J.tabs[i].mats_tstride = (int)(mp * mp * 2 + mp);
    am->assign(am_stride, 0.0);
    return true;
  };
  auto store_amats = [&](int i, const std::vector<double>& am) -> std::string {
    if (am.size() != am_stride) return "internal: approximating rows of unexpected size";
    std::memcpy(J.amats.data() + (size_t)i * am_stride, am.data(), am_stride * sizeof(double));
    return "";
  };
  auto prepare = [&](int i) -> std::string {
    if (models[i].n != n || models[i].model_id != id) return "all draws must share n and model id";
    std::vector<double> fp, am;
    int mm_ = 1, kk_ = 1;
    if (psi) {
      // approximate_for_is(mode_i): scales 0, approx_loglik 0 (src/approx_mcmc.cpp:372-373)
      if (dev_setup) {
        bssm::FilterTables dummy;
        if (to_fn_model(&models[i], &fms[i], 16)) return g_err;
        if (functor_tables(&models[i], &dummy, &fp, &mm_, &kk_)) return g_err;
        std::memcpy(J.fpar.data() + (size_t)i * J.fps, fp.data(), J.fps * sizeof(double));
        if (bssm::fn_psi_setup_begin(fms[i], modes + (size_t)m * n * i, &fsus[i], &lgs[i])) { pending[i] = 1; return ""; }
      }
      double al = 0; bool ok = true;
      if (functor_psi_tables(&models[i], modes + (size_t)m * n * i, 0, 1e-8, &J.tabs[i], &fp, &am, &al, &ok, &mm_)) return g_err;
      if (!ok && !bad_draw(i, &fp, &am)) return g_err;
      { const std::string e = store_amats(i, am); if (!e.empty()) return e; }
    } else {
      if (functor_tables(&models[i], &J.tabs[i], &fp, &mm_, &kk_)) return g_err;
      J.sub[i] = approx_loglik[i];
    }
    std::memcpy(J.fpar.data() + (size_t)i * J.fps, fp.data(), J.fps * sizeof(double));
    return "";
  };
  auto device_chunk = [&](int lo, int hi) -> int {
    std::vector<const bssm::LgPacked*> lp;
    std::vector<std::array<double*, 3>> outs;
    std::vector<int> who;
    for (int i = lo; i < hi; i++) if (pending[i]) {
      lp.push_back(&lgs[i]); who.push_back(i);
      outs.push_back({fsus[i].alphahat.data(), fsus[i].Vt.data(), fsus[i].Ct.data()});
    }
    std::vector<char> okv;
    const int rc = device_lg_batch(lp, outs, &okv);
    if (rc < 0) return rc;
    if (rc == 1) {
      parallel_for((int)who.size(), n_threads, [&](int q) {
        const int i = who[q];
        fsus[i].ok = bssm::lg_packed_run(lgs[i], fsus[i].alphahat.data(), fsus[i].Vt.data(), fsus[i].Ct.data());
      });
    } else {
      for (size_t q = 0; q < who.size(); q++) fsus[who[q]].ok = okv[q] != 0;
    }
    return 0;
  };
  auto post = [&](int i) -> std::string {
    if (!pending[i]) return "";
    std::vector<double> fp, am;
    if (fsus[i].ok) {
      bssm::build_fn_psi_tables(fms[i], fsus[i], &J.tabs[i], &am);
    } else {
      if (!bad_draw(i, &fp, &am)) return g_err;
      std::memcpy(J.fpar.data() + (size_t)i * J.fps, fp.data(), J.fps * sizeof(double));
    }
    { const std::string e = store_amats(i, am); if (!e.empty()) return e; }
    fms[i] = bssm::FnModel(); fsus[i] = bssm::FnSetup(); lgs[i] = bssm::LgPacked();
    pending[i] = 0;
    return "";
  };
  if (dev_setup) return run_is_pipeline(J, count, n_threads, prepare, nullptr, weights, loglik, alpha, alphahat_i, Vt_i, device_chunk, post);
  return run_is_pipeline(J, count, n_threads, prepare, nullptr, weights, loglik, alpha, alphahat_i, Vt_i);
}

int bssm_is_merge(int32_t n, int32_t m, int32_t n_stored, const double* weights, const uint32_t* counts,
                  const double* alphahat_i, const double* Vt_i, double* alphahat, double* Vt) {
  if (!weights || !counts || !alphahat_i || !Vt_i || !alphahat || !Vt) return fail(BSSM_ERR_INVALID, "NULL argument");
  const size_t mt = (size_t)m * (n + 1), mmt = mt * m;
  std::vector<double> Valpha(mmt, 0.0), diff(mt);
  std::fill(alphahat, alphahat + mt, 0.0);
  std::fill(Vt, Vt + mmt, 0.0);
  double sum_w = 0.0;
  for (int i = 0; i < n_stored; i++) {  // src/approx_mcmc.cpp:401-411, in draw order
    const double* ah = alphahat_i + mt * i;
    const double* vt = Vt_i + mmt * i;
    const double wnew = weights[i] * counts[i];
    sum_w += wnew;
    for (size_t q = 0; q < mt; q++) { diff[q] = ah[q] - alphahat[q]; alphahat[q] += wnew / sum_w * diff[q]; }
    for (int t = 0; t <= n; t++)
      for (int c = 0; c < m; c++) for (int j = 0; j < m; j++)
        Valpha[j + m * (c + (size_t)m * t)] += wnew * diff[j + (size_t)m * t] * (ah[c + (size_t)m * t] - alphahat[c + (size_t)m * t]);
    for (size_t q = 0; q < mmt; q++) Vt[q] += wnew / sum_w * (vt[q] - Vt[q]);
  }
  for (size_t q = 0; q < mmt; q++) Vt[q] += Valpha[q] / sum_w;  // :422-424
  return BSSM_OK;
}

// ------------------------------------------------------------------ one call, several GPUs
}  // extern "C"
namespace {
// Shard record of the last multi-device post-correction on this thread (bssm_amd_last_is_shards)
struct ShardInfo { int device, first, count; double setup_ms, device_ms, total_ms; };
thread_local std::vector<ShardInfo> g_shards;
thread_local double g_collective_ms = 0.0;
thread_local int g_gather_kind = 0;  // 0 none, 1 RCCL all-gather, 2 device copies (shards sharing a device), 3 host blocks (no device exchange possible)

// contiguous blocks with (nearly) equal total work: bssm_amd/post_correct.py shard_bounds, so that the single-process
// and the one-process-per-GPU drivers cut the draws at the same places
std::vector<int> shard_bounds(const uint32_t* counts, int cnt, int is_type, int world) {
  std::vector<double> cum(cnt + 1, 0.0);
  for (int i = 0; i < cnt; i++) cum[i + 1] = cum[i] + (is_type == 1 ? (double)counts[i] : 1.0);
  std::vector<int> b(world + 1, 0);
  for (int r = 1; r < world; r++) {
    const double target = cum[cnt] * r / world;
    const int k = (int)(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin());
    b[r] = std::min(std::max(k, b[r - 1]), cnt);
  }
  b[world] = cnt;
  return b;
}

// The exchange step of the one-process multi-device drivers: every shard has a block of `blk` doubles in host memory
// (padded to the largest block); each block goes to its shard's device, ONE ncclAllGather (RCCL over xGMI; comms from
// ncclCommInitAll, cached per device list) gives every device the whole vector, and `gathered` is filled from the first
// device's copy.  RCCL is the fast path, not a requirement: the work is done by the time this runs, so a collective that
// cannot start (librccl of another ROCm stack, ncclCommInitAll failing) degrades to device copies and, failing those, to
// the host blocks -- the call never fails here.  kind: 1 RCCL all-gather, 2 device copies, 3 host blocks.
void gather_shard_blocks(const std::vector<int>& devices, const std::vector<std::vector<double>>& host_blocks, size_t blk,
                         std::vector<double>* gathered, int* kind_out) {
  const int world = (int)devices.size();
  gathered->assign(blk * world, 0.0);
  int kind = 0;
  if (blk == 0 || world == 0) { *kind_out = 0; return; }
  bool distinct = true;
  for (int a = 0; a < world; a++) for (int b = a + 1; b < world; b++) if (devices[a] == devices[b]) distinct = false;
  // BSSM_AMD_GATHER: "copy" forces the device-copy exchange, "host" the host blocks
  const std::string mode = bssm::opt().gather == 1 ? "copy" : (bssm::opt().gather == 2 ? "host" : "");
  int dev0 = 0;
  (void)hipGetDevice(&dev0);
  std::vector<double*> d_send(world, nullptr), d_recv(world, nullptr);
  std::vector<size_t> cap_send(world, 0), cap_recv(world, 0);
  std::vector<hipStream_t> streams(world, nullptr);
  std::string note;
  bool staged = mode != "host";
  for (int r = 0; r < world && staged; r++) {  // the shards' blocks into device memory
    void *ps = nullptr, *pr = nullptr;
    staged = hipSetDevice(devices[r]) == hipSuccess && arena().get(blk * sizeof(double), &ps, &cap_send[r]) == hipSuccess;
    d_send[r] = (double*)ps;
    staged = staged && arena().get(blk * world * sizeof(double), &pr, &cap_recv[r]) == hipSuccess;
    d_recv[r] = (double*)pr;
    staged = staged && hipStreamCreateWithFlags(&streams[r], hipStreamNonBlocking) == hipSuccess;
    staged = staged && hipMemcpyAsync(d_send[r], host_blocks[r].data(), blk * sizeof(double), hipMemcpyHostToDevice, streams[r]) == hipSuccess;
    staged = staged && hipStreamSynchronize(streams[r]) == hipSuccess;
    if (!staged) { note += "staging the blocks on device " + std::to_string(devices[r]) + " failed; "; (void)hipGetLastError(); }
  }
  bool done = false;
  if (staged && distinct && mode != "copy") {
    std::string why;
    std::vector<const double*> snd(d_send.begin(), d_send.end());
    bool ok = bssm::rccl_all_gather(devices, snd, d_recv, blk, streams, &why) == 0;
    for (int r = 0; r < world && ok; r++) {
      (void)hipSetDevice(devices[r]);
      if (hipStreamSynchronize(streams[r]) != hipSuccess) { ok = false; why = "all-gather: stream synchronise failed"; }
    }
    if (ok) { kind = 1; done = true; }
    else { note += "RCCL all-gather unavailable (" + why + "); "; (void)hipGetLastError(); }
  }
  if (staged && !done) {  // shards sharing a GPU cannot be ranks of one communicator: the same exchange as device copies
    (void)hipSetDevice(devices[0]);
    bool ok = true;
    for (int r = 0; r < world && ok; r++)
      ok = hipMemcpyAsync(d_recv[0] + blk * r, d_send[r], blk * sizeof(double), hipMemcpyDefault, streams[0]) == hipSuccess;
    ok = ok && hipStreamSynchronize(streams[0]) == hipSuccess;
    if (ok) { kind = 2; done = true; }
    else { note += "device-copy exchange failed; "; (void)hipGetLastError(); }
  }
  if (done) {
    (void)hipSetDevice(devices[0]);
    if (hipMemcpy(gathered->data(), d_recv[0], gathered->size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) {
      done = false; note += "download of the gathered blocks failed; "; (void)hipGetLastError();
    }
  }
  if (!done) {
    kind = 3;
    for (int r = 0; r < world; r++) std::memcpy(gathered->data() + blk * r, host_blocks[r].data(), blk * sizeof(double));
  }
  for (int r = 0; r < world; r++) {
    (void)hipSetDevice(devices[r]);
    if (streams[r]) (void)hipStreamDestroy(streams[r]);
    if (d_send[r]) arena().put(d_send[r], cap_send[r]);
    if (d_recv[r]) arena().put(d_recv[r], cap_recv[r]);
  }
  (void)hipSetDevice(dev0);
  if (!note.empty() && bssm::opt().trace) std::fprintf(stderr, "[bssm_amd] exchange: %sfell back\n", note.c_str());
  *kind_out = kind;
}

int resolve_devices(const int32_t* devices_in, int n_devices_in, std::vector<int>* devices) {
  int visible = 0;
  HIP_TRY(hipGetDeviceCount(&visible));
  devices->clear();
  if (devices_in && n_devices_in > 0) devices->assign(devices_in, devices_in + n_devices_in);
  else for (int d = 0; d < visible; d++) devices->push_back(d);
  for (int d : *devices) if (d < 0 || d >= visible) return fail(BSSM_ERR_INVALID, "device ordinal " + std::to_string(d) + " is not visible");
  return BSSM_OK;
}

// The draws of one post-correction call spread over `devices` (HIP ordinals; the same ordinal may appear more than
// once = logical shards on one GPU): one host thread per shard sets its device and runs the single-device driver on
// its block (run_shard(lo, hi, threads)), which fills the caller's per-draw alpha / summary blocks directly (every
// draw has a slot of its own) and hands weights / log-likelihoods to `w_loc`, `ll_loc`.  Those two vectors are what
// the shards exchange (gather_shard_blocks).  Mirrors approx_mcmc::is_correction_* spreading the draws over OpenMP
// threads inside one call (src/approx_mcmc.cpp:364-369, schedule(static)).
int run_multi_device(int cnt, const uint32_t* counts, int is_type, const int32_t* devices_in, int n_devices_in, int n_threads,
                     const std::function<int(int, int, int, double*, double*)>& run_shard, double* weights, double* loglik) {
  if (!have_device()) return no_device();
  std::vector<int> devices;
  if (int rc = resolve_devices(devices_in, n_devices_in, &devices)) return rc;
  const int world = (int)devices.size();
  const std::vector<int> bounds = shard_bounds(counts, cnt, is_type, world);
  int block = 0;
  for (int r = 0; r < world; r++) block = std::max(block, bounds[r + 1] - bounds[r]);
  int dev0 = 0;
  (void)hipGetDevice(&dev0);
  const int threads_each = std::max(1, n_threads / world);  // the host cores are shared by the shards

  std::vector<ShardInfo> info(world);
  std::vector<int> rcs(world, BSSM_OK);
  std::vector<std::string> errs(world);
  std::vector<std::thread> pool;
  // every shard's [weights | log-likelihoods] block, padded to the largest block
  std::vector<std::vector<double>> host_blocks(world, std::vector<double>((size_t)2 * block, 0.0));
  const double t_all = now_ms();
  for (int r = 0; r < world; r++) {
    pool.emplace_back([&, r]() {
      const int lo = bounds[r], hi = bounds[r + 1];
      info[r] = ShardInfo{devices[r], lo, hi - lo, 0.0, 0.0, 0.0};
      if (hipSetDevice(devices[r]) != hipSuccess) { errs[r] = "hipSetDevice failed"; rcs[r] = BSSM_ERR_HIP; return; }
      bssm::prewarm_streams();  // this shard's device: its streams in the library's order (once per device, options.cpp)
      const double t0 = now_ms();
      std::vector<double>& buf = host_blocks[r];
      if (hi > lo) {
        const int rc = run_shard(lo, hi, threads_each, buf.data(), buf.data() + block);
        rcs[r] = rc;
        if (rc < 0) { errs[r] = g_err; return; }
        info[r].setup_ms = g_is_timing[0]; info[r].device_ms = g_is_timing[1];
      }
      info[r].total_ms = now_ms() - t0;
    });
  }
  for (auto& th : pool) th.join();
  (void)hipSetDevice(dev0);
  int worst = BSSM_OK;
  for (int r = 0; r < world; r++) {
    if (rcs[r] < 0) return fail(rcs[r], "shard " + std::to_string(r) + " (device " + std::to_string(devices[r]) + "): " + errs[r]);
    if (rcs[r] == BSSM_WARN_DEGENERATE) worst = BSSM_WARN_DEGENERATE;
  }
  // ---- the collective
  const double t_c = now_ms();
  const size_t blk = (size_t)2 * block;
  std::vector<double> gathered;
  int kind = 0;
  gather_shard_blocks(devices, host_blocks, blk, &gathered, &kind);
  for (int r = 0; r < world; r++) {
    const int lo = bounds[r], c = bounds[r + 1] - lo;
    std::memcpy(weights + lo, gathered.data() + blk * r, (size_t)c * sizeof(double));
    if (loglik) std::memcpy(loglik + lo, gathered.data() + blk * r + block, (size_t)c * sizeof(double));
  }
  g_collective_ms = now_ms() - t_c;
  g_gather_kind = kind;
  g_shards = info;
  double th = 0, td = 0;
  for (const ShardInfo& si : info) { th = std::max(th, si.setup_ms); td = std::max(td, si.device_ms); }
  g_is_timing[0] = th; g_is_timing[1] = td;
  if (bssm::opt().trace)
    std::fprintf(stderr, "[bssm_amd] post-correction over %d shard(s): %.1f ms, of which the exchange (%s) %.2f ms\n", world,
                 now_ms() - t_all, kind == 1 ? "RCCL all-gather" : kind == 2 ? "device copies" : "host blocks", g_collective_ms);
  return worst;
}

// suggest_N over several devices in one call (src/R_postcorrection.cpp:24-35 runs the candidates x replications loop on
// one thread): the replications of EVERY candidate are cut into contiguous blocks, one block per shard; a shard runs its
// replications of candidate i as one device batch on streams i * replications + [lo, hi) -- the streams the single-device
// entry point uses, so the log-likelihoods (and the standard deviations) are bit-identical for any device list; ONE
// exchange of the shards' (n_candidates x block) log-likelihood blocks at the end.
int suggest_n_multi(const uint32_t* candidates, int32_t n_candidates, uint32_t replications, const int32_t* devices_in,
                    int n_devices_in, double* sds, double* loglik_out,
                    const std::function<int(uint32_t, uint32_t, uint32_t, double*)>& batch) {
  if (!candidates || !sds || n_candidates < 1 || replications < 1) return fail(BSSM_ERR_INVALID, "bad arguments");
  for (int i = 0; i < n_candidates; i++) if (candidates[i] < 1) return fail(BSSM_ERR_INVALID, "candidates must be positive");
  if (!have_device()) return no_device();
  std::vector<int> devices;
  if (int rc = resolve_devices(devices_in, n_devices_in, &devices)) return rc;
  const int world = (int)devices.size(), reps = (int)replications;
  std::vector<int> bounds(world + 1, 0);
  for (int r = 0; r <= world; r++) bounds[r] = (int)((long long)reps * r / world);
  int block = 0;
  for (int r = 0; r < world; r++) block = std::max(block, bounds[r + 1] - bounds[r]);
  const size_t blk = (size_t)n_candidates * block;
  std::vector<std::vector<double>> host_blocks(world, std::vector<double>(blk, 0.0));
  std::vector<int> rcs(world, BSSM_OK);
  std::vector<std::string> errs(world);
  std::vector<std::thread> pool;
  int dev0 = 0;
  (void)hipGetDevice(&dev0);
  for (int r = 0; r < world; r++) {
    pool.emplace_back([&, r]() {
      const int lo = bounds[r], hi = bounds[r + 1];
      if (hi <= lo) return;
      if (hipSetDevice(devices[r]) != hipSuccess) { errs[r] = "hipSetDevice failed"; rcs[r] = BSSM_ERR_HIP; return; }
      bssm::prewarm_streams();
      for (int i = 0; i < n_candidates; i++) {
        const int rc = batch(candidates[i], (uint32_t)i * replications + (uint32_t)lo, (uint32_t)(hi - lo), host_blocks[r].data() + (size_t)i * block);
        if (rc < 0) { errs[r] = g_err; rcs[r] = rc; return; }
        if (rc == BSSM_WARN_DEGENERATE) rcs[r] = rc;
      }
    });
  }
  for (auto& th : pool) th.join();
  (void)hipSetDevice(dev0);
  bool deg = false;
  for (int r = 0; r < world; r++) {
    if (rcs[r] < 0) return fail(rcs[r], "shard " + std::to_string(r) + " (device " + std::to_string(devices[r]) + "): " + errs[r]);
    deg = deg || rcs[r] == BSSM_WARN_DEGENERATE;
  }
  const double t_c = now_ms();
  std::vector<double> gathered;
  int kind = 0;
  gather_shard_blocks(devices, host_blocks, blk, &gathered, &kind);
  g_collective_ms = now_ms() - t_c;
  g_gather_kind = kind;
  std::vector<double> ll(replications);
  for (int i = 0; i < n_candidates; i++) {
    for (int r = 0; r < world; r++)
      for (int j = bounds[r]; j < bounds[r + 1]; j++) ll[j] = gathered[blk * r + (size_t)i * block + (j - bounds[r])];
    sds[i] = sample_sd(ll);
    if (loglik_out) std::memcpy(loglik_out + (size_t)i * replications, ll.data(), replications * sizeof(double));
  }
  return deg ? BSSM_WARN_DEGENERATE : BSSM_OK;
}
}  // namespace
extern "C" {

int bssm_is_correction_multi(const bssm_is_job* job, const int32_t* devices, int32_t n_devices, double* weights,
                             double* loglik, double* alpha, double* alphahat_i, double* Vt_i) {
  if (!job || !weights) return fail(BSSM_ERR_INVALID, "job/weights NULL");
  if (job->count <= 0) return BSSM_OK;
  if (!job->counts || !job->models) return fail(BSSM_ERR_INVALID, "counts/models NULL");
  const size_t n = job->models[0].n, m = job->models[0].m, p = job->models[0].p > 1 ? job->models[0].p : 1;
  auto run_shard = [&](int lo, int hi, int threads, double* w_loc, double* ll_loc) -> int {
    bssm_is_job sub = *job;
    sub.first = job->first + lo; sub.count = hi - lo;
    sub.models = job->models + lo;
    sub.modes = job->modes ? job->modes + n * p * lo : nullptr;
    sub.approx_loglik = job->approx_loglik ? job->approx_loglik + lo : nullptr;
    sub.counts = job->counts + lo;
    sub.n_threads = threads;
    return bssm_is_correction(&sub, w_loc, ll_loc, alpha ? alpha + (n + 1) * m * lo : nullptr,
                              alphahat_i ? alphahat_i + (n + 1) * m * lo : nullptr, Vt_i ? Vt_i + (n + 1) * m * m * lo : nullptr);
  };
  return run_multi_device(job->count, job->counts, job->is_type, devices, n_devices, job->n_threads, run_shard, weights, loglik);
}

int bssm_is_correction_nlg_multi(const bssm_fn_model* models, int32_t first, int32_t count, const double* modes,
                                 const double* approx_loglik, const uint32_t* counts, int32_t sampling_method,
                                 int32_t is_type, int32_t output_type, uint32_t nsim, uint32_t seed, int32_t n_threads,
                                 const int32_t* devices, int32_t n_devices, double* weights, double* loglik, double* alpha,
                                 double* alphahat_i, double* Vt_i) {
  if (!models || !weights || !counts) return fail(BSSM_ERR_INVALID, "models/weights/counts NULL");
  if (count <= 0) return BSSM_OK;
  const int id = models[0].model_id;
  const size_t n = models[0].n, m = id == bssm::FN_NLG_GROWTH ? 2 : (id == bssm::FN_USER_NLG ? models[0].n_states : 1);
  auto run_shard = [&](int lo, int hi, int threads, double* w_loc, double* ll_loc) -> int {
    return bssm_is_correction_nlg(models + lo, first + lo, hi - lo, modes ? modes + m * n * lo : nullptr,
                                  approx_loglik ? approx_loglik + lo : nullptr, counts + lo, sampling_method, is_type,
                                  output_type, nsim, seed, threads, w_loc, ll_loc, alpha ? alpha + (n + 1) * m * lo : nullptr,
                                  alphahat_i ? alphahat_i + (n + 1) * m * lo : nullptr, Vt_i ? Vt_i + (n + 1) * m * m * lo : nullptr);
  };
  return run_multi_device(count, counts, is_type, devices, n_devices, n_threads, run_shard, weights, loglik);
}

int bssm_suggest_n_nongaussian_multi(const bssm_ung_model* model, const uint32_t* candidates, int32_t n_candidates,
                                     uint32_t replications, uint32_t seed, int model_type, const int32_t* devices,
                                     int32_t n_devices, double* sds, double* loglik) {
  if (!is_gaussian_type_ok(0, model_type)) return fail(BSSM_ERR_UNSUPPORTED, "model_type not handled by this entry point");
  return suggest_n_multi(candidates, n_candidates, replications, devices, n_devices, sds, loglik,
                         [&](uint32_t nsim, uint32_t stream0, uint32_t n_rep, double* ll) {
                           return bssm_nongaussian_loglik_batch(model, nsim, BSSM_PSI, seed, stream0, n_rep, ll);
                         });
}

int bssm_suggest_n_nonlinear_multi(const bssm_fn_model* model, const uint32_t* candidates, int32_t n_candidates,
                                   uint32_t replications, uint32_t seed, const int32_t* devices, int32_t n_devices,
                                   double* sds, double* loglik) {
  return suggest_n_multi(candidates, n_candidates, replications, devices, n_devices, sds, loglik,
                         [&](uint32_t nsim, uint32_t stream0, uint32_t n_rep, double* ll) {
                           return bssm_fn_loglik_batch(model, nsim, BSSM_PSI, seed, stream0, n_rep, ll);
                         });
}

int bssm_amd_last_is_shards(double* out, int32_t capacity, double* collective_ms, int32_t* gather_kind) {
  const int ns = (int)g_shards.size();
  for (int r = 0; r < ns && r < capacity; r++) {
    const ShardInfo& s = g_shards[r];
    double* o = out + (size_t)6 * r;
    o[0] = s.device; o[1] = s.first; o[2] = s.count; o[3] = s.setup_ms; o[4] = s.device_ms; o[5] = s.total_ms;
  }
  if (collective_ms) *collective_ms = g_collective_ms;
  if (gather_kind) *gather_kind = g_gather_kind;
  return ns;
}

int bssm_amd_rccl_available(void) {
  std::string why;
  if (bssm::rccl_available(&why)) return 1;
  g_err = why;
  return 0;
}
}  // extern "C"
