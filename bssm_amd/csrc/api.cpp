// api.cpp -- the C ABI (include/bssm_amd.h): host orchestration around the HIP kernels.
// There is no CPU compute path in this library: every filter runs on the HIP device or fails.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bssm_amd.h"
#include "host_model.h"
#include "options.h"
#include "pf_device.h"
#include "pf_functors.h"
#include "pf_kernels.h"
#include "rccl_gather.h"
#include "tls_stream.h"

namespace {

thread_local std::string g_err;
bssm::UserHostFns g_user_host[64];  // host entry points of the loaded user-model libraries, by registry slot
thread_local uint32_t g_call_stream0 = 0;  // Philox stream of the next single-filter call on this thread (bssm_fn_smoother_stream)
thread_local double g_is_timing[2] = {0.0, 0.0};  // last IS post-correction on this thread: host set-up ms, device phase ms
// the same call split further: per-draw host packing (prepare / post / finish), the per-chunk device set-up call (uploads,
// set-up kernel launch and, where the pipeline waits for it, the wait), the filter thread's busy time, wall time of the call
thread_local double g_is_timing4[4] = {0.0, 0.0, 0.0, 0.0};
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int fail(int code, const std::string& msg) { g_err = msg; return code; }
}  // namespace
namespace bssm { int set_error(int code, const std::string& msg) { return fail(code, msg); } }
namespace {

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      g_err = std::string(#expr) + ": " + hipGetErrorString(e_);                               \
      return BSSM_ERR_HIP;                                                                     \
    }                                                                                          \
  } while (0)

bool have_device() {
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess) return false;
  return cnt > 0;
}
int no_device() {
  return fail(BSSM_ERR_NO_DEVICE,
              "no HIP device available: bssm_amd has no CPU fallback (the particle filters run only on the GPU)");
}

// Device memory arena: freed buffers are kept and handed out again.  A filter call needs GBs of scratch (particle
// history, ancestor table, workspace); hipMalloc / hipFree of that size cost milliseconds to seconds and hipFree
// synchronises the device, which dominated repeated calls (pseudo-marginal MCMC, post-correction batches).  Blocks are
// keyed by (device, size rounded up to 2 MiB above 1 MiB, 4 KiB below); a request takes the smallest cached block that
// is at most 25 % larger.  Everything in this library runs on the caller's stream or the null stream and every entry
// point drains its work before its buffers go out of scope, so a cached block is idle when it is reused.
struct DeviceArena {
  bool pinned_host = false;  // true: hipHostMalloc'd staging blocks instead of device memory
  std::mutex mu;
  std::multimap<std::pair<int, size_t>, void*> free_blocks;
  size_t cached = 0;
  size_t cap;
  DeviceArena() {
    const double gb = bssm::opt().cache_gb;
    cap = (size_t)(gb >= 0.0 ? gb : 96.0) << 30;
  }
  static size_t round_up(size_t b) {
    const size_t g = b >= ((size_t)1 << 20) ? ((size_t)2 << 20) : 4096;
    return (b + g - 1) / g * g;
  }
  hipError_t get(size_t bytes, void** out, size_t* got) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const size_t want = round_up(bytes);
    {
      std::lock_guard<std::mutex> lk(mu);
      auto it = free_blocks.lower_bound({dev, want});
      if (it != free_blocks.end() && it->first.first == dev && it->first.second <= want + want / 4) {
        *out = it->second; *got = it->first.second;
        cached -= *got;
        free_blocks.erase(it);
        return hipSuccess;
      }
    }
    hipError_t e = raw_alloc(out, want);
    while (e != hipSuccess) {  // out of memory: give cached blocks back, largest first, until the request fits (freeing
      (void)hipGetLastError();  // the whole cache -- tens of GB after a large job -- stalls the caller for seconds)
      void* victim = nullptr;
      {
        std::lock_guard<std::mutex> lk(mu);
        auto best = free_blocks.end();
        for (auto it = free_blocks.begin(); it != free_blocks.end(); ++it)
          if (it->first.first == dev && (best == free_blocks.end() || it->first.second > best->first.second)) best = it;
        if (best == free_blocks.end()) break;
        victim = best->second;
        cached -= best->first.second;
        free_blocks.erase(best);
      }
      raw_free(victim);
      e = raw_alloc(out, want);
    }
    *got = want;
    return e;
  }
  hipError_t raw_alloc(void** out, size_t b) { return pinned_host ? hipHostMalloc(out, b, hipHostMallocDefault) : hipMalloc(out, b); }
  void raw_free(void* p) { if (pinned_host) (void)hipHostFree(p); else (void)hipFree(p); }
  void put(void* p, size_t size) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
      std::lock_guard<std::mutex> lk(mu);
      if (cached + size <= cap) { free_blocks.insert({{dev, size}, p}); cached += size; return; }
    }
    raw_free(p);
  }
  void trim() {
    std::lock_guard<std::mutex> lk(mu);
    for (auto& kv : free_blocks) raw_free(kv.second);
    free_blocks.clear();
    cached = 0;
  }
};
DeviceArena& arena() { static DeviceArena* a = new DeviceArena; return *a; }  // leaked on purpose: no hipFree at exit
// pinned staging for host -> device table uploads (pageable copies run at a few GB/s and block the launch thread)
DeviceArena& stage_arena() {
  static DeviceArena* a = [] { DeviceArena* x = new DeviceArena; x->pinned_host = true; x->cap = (size_t)4 << 30; return x; }();
  return *a;
}
struct StageBuf {
  void* p = nullptr;
  size_t capacity = 0;
  StageBuf() {}
  StageBuf(const StageBuf&) = delete;
  StageBuf& operator=(const StageBuf&) = delete;
  ~StageBuf() { if (p) stage_arena().put(p, capacity); }
  hipError_t alloc(size_t b) { return stage_arena().get(b, &p, &capacity); }
};

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;     // requested size
  size_t capacity = 0;  // size of the arena block behind it
  DevBuf() {}
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() { if (p) { arena().put(p, capacity); p = nullptr; } }
  hipError_t alloc(size_t b) {
    release();
    bytes = b;
    if (b == 0) return hipSuccess;
    return arena().get(b, &p, &capacity);
  }
  template <class T> T* as() const { return (T*)p; }
};


// ------------------------------------------------------------------ device -> pageable host output pipe
// The drop-in entry points hand back host buffers the CALLER allocated (R vectors, numpy arrays): pageable, and usually
// never touched, so a plain hipMemcpy into them runs at the page-fault rate (measured 10 GB/s, scripts/pcie_probe.py)
// and only starts when everything before it has finished.  OutPipe streams instead: chunks go device -> pinned ring
// buffer by DMA on a copy stream (57 GB/s, each chunk gated by the event that says its device data is final), and a few
// host threads move every landed chunk pinned -> caller memory in parallel slices (first-touch included: 68 GB/s with 8
// threads, 88 with 16), while the next chunk is in flight and -- for finished time slabs -- the filter is still stepping.
// Reference contract: src/R_bootstrap_filter.cpp:136-155 (outputs are ordinary R arrays).
class OutPipe {
 public:
  explicit OutPipe(size_t total_bytes) {
    int hw = (int)std::thread::hardware_concurrency();
    threads_ = std::max(1, std::min(16, hw / 2));
    if (bssm::opt().copy_threads > 0) threads_ = std::max(1, std::min(64, bssm::opt().copy_threads));
    chunk_ = (size_t)32 << 20;
    if (total_bytes < chunk_ * 2) threads_ = std::min(threads_, 4);
    (void)hipGetDevice(&dev_);
  }
  ~OutPipe() { (void)finish(); }
  OutPipe(const OutPipe&) = delete;
  OutPipe& operator=(const OutPipe&) = delete;

  // copy `bytes` from device `src` to host `dst` once `ready` has fired (nullptr: once everything enqueued on `after` so
  // far has finished)
  int push(const void* src, void* dst, size_t bytes, hipEvent_t ready, hipStream_t after = nullptr) {
    if (bytes == 0) return BSSM_OK;
    if (int rc = start()) return rc;
    if (ready) {
      HIP_TRY(hipStreamWaitEvent(cs_, ready, 0));
    } else {
      hipEvent_t ev;
      HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      HIP_TRY(hipEventRecord(ev, after));
      HIP_TRY(hipStreamWaitEvent(cs_, ev, 0));
      (void)hipEventDestroy(ev);
    }
    for (size_t off = 0; off < bytes; off += chunk_) {
      const size_t len = std::min(chunk_, bytes - off);
      const long k = posted_;
      const int slot = (int)(k % kSlots);
      if (k >= kSlots) {  // the slot's previous chunk must have left the ring
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [&] { return finished_[(size_t)(k - kSlots)] == threads_; });
      }
      HIP_TRY(hipMemcpyAsync(ring_[slot].p, (const char*)src + off, len, hipMemcpyDeviceToHost, cs_));
      HIP_TRY(hipEventRecord(done_[slot], cs_));
      {
        std::lock_guard<std::mutex> lk(mu_);
        chunks_.push_back(Chunk{(char*)dst + off, len, slot});
        finished_.push_back(0);
        posted_++;
      }
      cv_post_.notify_all();
    }
    return BSSM_OK;
  }

  int finish() {
    if (!started_ || joined_) return BSSM_OK;
    {
      std::lock_guard<std::mutex> lk(mu_);
      closing_ = true;
    }
    cv_post_.notify_all();
    for (auto& t : pool_) t.join();
    joined_ = true;
    for (int s = 0; s < kSlots; s++) if (done_[s]) (void)hipEventDestroy(done_[s]);
    return failed_ ? fail(BSSM_ERR_HIP, "output pipe: event wait failed") : BSSM_OK;
  }

 private:
  static constexpr int kSlots = 4;
  struct Chunk { char* dst; size_t len; int slot; };

  int start() {
    if (started_) return BSSM_OK;
    static thread_local bssm::TlsStream cs;
    cs_ = cs.get();
    if (!cs_) return fail(BSSM_ERR_HIP, "output pipe: no copy stream");
    for (int s = 0; s < kSlots; s++) {
      if (ring_[s].alloc(chunk_) != hipSuccess || hipEventCreateWithFlags(&done_[s], hipEventDisableTiming) != hipSuccess) {
        for (int q = 0; q <= s; q++) if (done_[q]) { (void)hipEventDestroy(done_[q]); done_[q] = nullptr; }  // (the ring blocks return with the object)
        return fail(BSSM_ERR_HIP, "output pipe: no pinned ring / events");
      }
    }
    started_ = true;
    for (int q = 0; q < threads_; q++) pool_.emplace_back([this, q] { worker(q); });
    return BSSM_OK;
  }

  void worker(int q) {
    (void)hipSetDevice(dev_);
    for (long k = 0;; k++) {
      Chunk c;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_post_.wait(lk, [&] { return posted_ > k || closing_; });
        if (posted_ <= k) return;
        c = chunks_[(size_t)k];
      }
      if (hipEventSynchronize(done_[c.slot]) != hipSuccess) failed_ = true;
      const size_t lo = c.len * (size_t)q / (size_t)threads_ / 64 * 64, hi = q + 1 == threads_ ? c.len : c.len * (size_t)(q + 1) / (size_t)threads_ / 64 * 64;
      if (hi > lo) std::memcpy(c.dst + lo, (const char*)ring_[c.slot].p + lo, hi - lo);
      {
        std::lock_guard<std::mutex> lk(mu_);
        finished_[(size_t)k]++;
      }
      cv_done_.notify_all();
    }
  }

  int threads_ = 1, dev_ = 0;
  size_t chunk_ = 0;
  hipStream_t cs_ = nullptr;
  StageBuf ring_[kSlots];
  hipEvent_t done_[kSlots] = {};
  std::vector<std::thread> pool_;
  std::mutex mu_;
  std::condition_variable cv_post_, cv_done_;
  std::vector<Chunk> chunks_;
  std::vector<int> finished_;
  long posted_ = 0;
  bool started_ = false, joined_ = false, closing_ = false;
  std::atomic<bool> failed_{false};
};

template <class F>
void parallel_for(int count, int threads, F fn) {
  threads = std::max(1, std::min(threads, count));
  if (threads == 1) { for (int i = 0; i < count; i++) fn(i); return; }
  std::atomic<int> next(0);
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; t++)
    pool.emplace_back([&]() { for (int i = next++; i < count; i = next++) fn(i); });
  for (auto& th : pool) th.join();
}

// ------------------------------------------------------------------ a batch of filters on the device
struct DeviceBatch {
  bssm::FilterBatch fb{};
  DevBuf mats, obs, init, hist, indices, weights, loglik, dead, ws, fparbuf, amatbuf;
  bssm::GridWorkspace gws{};
  bool use_grid = false;
  bool want_factored = false;     // BSSM_PLAN_FACTORED_WEIGHTS: ask the step kernels for the factored weights output
  bool weights_factored = false;  // the last run left `weights` in factored form (GridWorkspace::fw was granted)
  std::vector<double> const_terms;  // per filter, added to the device log-likelihood on the host

  // tables: one per filter (all with the same MP/KP/mats_tstride)
  int upload(const std::vector<bssm::FilterTables>& tabs, int n, int m, int k, int p, int N, int psi, bool keep_hist,
             unsigned path_flags, bool keep_weights = true) {
    std::vector<const bssm::FilterTables*> ptrs(tabs.size());
    for (size_t f = 0; f < tabs.size(); f++) ptrs[f] = &tabs[f];
    return upload(ptrs, n, m, k, p, N, psi, keep_hist, path_flags, keep_weights);
  }
  int upload(const std::vector<const bssm::FilterTables*>& tabs, int n, int m, int k, int p, int N, int psi, bool keep_hist,
             unsigned path_flags, bool keep_weights = true, int copy_threads = 1) {
    const int F = (int)tabs.size();
    const int MP = tabs[0]->MP, KP = tabs[0]->KP;
    fb.n = n; fb.m = m; fb.k = k; fb.p = p; fb.MP = MP; fb.KP = KP; fb.N = N; fb.F = F; fb.psi = psi;
    const size_t ms = tabs[0]->mats.size(), os = tabs[0]->obs.size(), is = tabs[0]->init.size();
    for (int f = 0; f < F; f++)
      if (tabs[f]->mats.size() != ms || tabs[f]->obs.size() != os || tabs[f]->init.size() != is || tabs[f]->MP != MP || tabs[f]->KP != KP)
        return fail(BSSM_ERR_INVALID, "filters of a batch must share padding and time-variation");
    // one pinned staging block [mats | obs | init], filled by a few threads, one asynchronous copy each
    StageBuf stage;
    HIP_TRY(stage.alloc((ms + os + is) * F * sizeof(double)));
    double* hm = (double*)stage.p; double* ho = hm + ms * F; double* hi = ho + os * F;
    parallel_for(F, copy_threads, [&](int f) {
      std::memcpy(hm + ms * f, tabs[f]->mats.data(), ms * sizeof(double));
      std::memcpy(ho + os * f, tabs[f]->obs.data(), os * sizeof(double));
      std::memcpy(hi + is * f, tabs[f]->init.data(), is * sizeof(double));
    });
    HIP_TRY(mats.alloc(ms * F * sizeof(double)));
    HIP_TRY(obs.alloc(os * F * sizeof(double)));
    HIP_TRY(init.alloc(is * F * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(mats.p, hm, ms * F * sizeof(double), hipMemcpyHostToDevice, nullptr));
    HIP_TRY(hipMemcpyAsync(obs.p, ho, os * F * sizeof(double), hipMemcpyHostToDevice, nullptr));
    HIP_TRY(hipMemcpyAsync(init.p, hi, is * F * sizeof(double), hipMemcpyHostToDevice, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));  // the staging block returns to its arena when this function ends
    fb.mats = mats.as<double>(); fb.mats_fstride = ms; fb.mats_tstride = tabs[0]->mats_tstride;
    fb.obs = obs.as<double>(); fb.obs_fstride = os;
    fb.init = init.as<double>(); fb.init_fstride = is;
    return configure(keep_hist, path_flags, keep_weights);
  }
  // tables that are already in device memory (built there by fn_setup_kernel): F filters starting at the given pointers
  int attach(const double* d_mats, size_t ms, int tstride, const double* d_obs, size_t os, const double* d_init, size_t is,
             int F, int n, int m, int k, int p, int N, int psi, int MP, int KP) {
    fb.n = n; fb.m = m; fb.k = k; fb.p = p; fb.MP = MP; fb.KP = KP; fb.N = N; fb.F = F; fb.psi = psi;
    fb.mats = d_mats; fb.mats_fstride = ms; fb.mats_tstride = tstride;
    fb.obs = d_obs; fb.obs_fstride = os;
    fb.init = d_init; fb.init_fstride = is;
    return configure(false, 0, false);
  }
  int configure(bool keep_hist, unsigned path_flags, bool keep_weights) {
    const int F = fb.F, n = fb.n, m = fb.m, N = fb.N, MP = fb.MP;
    HIP_TRY(loglik.alloc(F * sizeof(double)));
    fb.loglik = loglik.as<double>();
    HIP_TRY(dead.alloc(F * sizeof(int)));
    fb.dead_at = dead.as<int>();
    if (bssm::opt().probe_no_weights) keep_weights = false;  // measurement probe, -DBSSM_PROBES builds only (scripts/traffic_probe.sh): what the weights output costs
    if (keep_hist) {
      HIP_TRY(hist.alloc((size_t)F * (n + 1) * m * N * sizeof(double)));
      HIP_TRY(indices.alloc((size_t)F * n * N * sizeof(uint32_t)));
      if (keep_weights) HIP_TRY(weights.alloc((size_t)F * (n + 1) * N * sizeof(double)));
      fb.alpha_hist = hist.as<double>(); fb.indices = indices.as<uint32_t>(); fb.weights = keep_weights ? weights.as<double>() : nullptr;
    }
    // kernel shape: workgroup-resident when the particle system fits one CU's LDS and there are
    // enough filters to fill the chip (or the filter is small); otherwise one launch per time step.
    const int cap = bssm::wg_filter_capacity(MP);
    use_grid = !(N <= cap && (F >= 32 || N <= 2048));
    if (bssm::opt().force_path == 2) use_grid = true;
    if (bssm::opt().force_path == 1) use_grid = false;
    if (path_flags & BSSM_PLAN_FORCE_GRID) use_grid = true;
    if (path_flags & BSSM_PLAN_FORCE_WG) use_grid = false;
    if (!use_grid && N > cap) return fail(BSSM_ERR_INVALID, "nsim exceeds the workgroup-resident capacity");
    if (use_grid) {
      int B, chunk;
      const size_t bytes = bssm::grid_workspace_bytes(fb, &B, &chunk);
      if (B > 256) return fail(BSSM_ERR_UNSUPPORTED, "nsim too large for the grid filter (max 524288)");
      HIP_TRY(ws.alloc(bytes));
      bssm::grid_workspace_carve(fb, ws.p, &gws);
    }
    return BSSM_OK;
  }

  // built-in functor parameter blocks, one FP_SIZE block per filter (or one shared block: stride 0)
  int set_fpar(const std::vector<double>& fp) {
    HIP_TRY(fparbuf.alloc(fp.size() * sizeof(double)));
    HIP_TRY(hipMemcpy(fparbuf.p, fp.data(), fparbuf.bytes, hipMemcpyHostToDevice));
    fb.fpar = fparbuf.as<double>();
    fb.fpar_fstride = fp.size() > (size_t)bssm::FP_SIZE ? bssm::FP_SIZE : 0;
    fb.user_slot = (int)fp[bssm::FP_SLOT];  // run-time compiled user model: its library holds the kernels
    fb.fn_id = (int)fp[bssm::FP_ID];
    return BSSM_OK;
  }

  int set_amats(const std::vector<double>& am) {  // shared by every filter of the batch (stride 0)
    HIP_TRY(amatbuf.alloc(am.size() * sizeof(double)));
    HIP_TRY(hipMemcpy(amatbuf.p, am.data(), amatbuf.bytes, hipMemcpyHostToDevice));
    fb.amats = amatbuf.as<double>();
    fb.amats_fstride = 0;
    return BSSM_OK;
  }

  int run(uint32_t seed, uint32_t stream0, hipStream_t st, bssm::StepTiming* timing = nullptr) {
    fb.seed = seed; fb.stream0 = stream0;
    weights_factored = false;
    if (use_grid) {
      bssm::StepTiming local;
      if (!timing) timing = &local;
      gws.fw = want_factored ? 1 : 0;
      HIP_TRY(bssm::launch_grid_filter(fb, gws, st, timing));
      weights_factored = timing->factored != 0;
    } else if (timing && timing->every > 0) {  // one launch holds the whole filter: time that launch
      hipEvent_t e0, e1;
      HIP_TRY(hipEventCreate(&e0));
      HIP_TRY(hipEventCreate(&e1));
      HIP_TRY(hipEventRecord(e0, st));
      HIP_TRY(bssm::launch_wg_filter(fb, st));
      HIP_TRY(hipEventRecord(e1, st));
      HIP_TRY(hipEventSynchronize(e1));
      float ms = 0;
      HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
      timing->total_ms += ms;
      timing->count++;
      (void)hipEventDestroy(e0);
      (void)hipEventDestroy(e1);
    } else {
      HIP_TRY(bssm::launch_wg_filter(fb, st));
    }
    // the reference zero-fills its outputs (src/R_bootstrap_filter.cpp:136-138): only a degenerate filter leaves
    // slots unwritten, so zero just those instead of memset-ing tens of GB per run
    HIP_TRY(bssm::launch_zero_dead_tail(fb, true, st, weights_factored ? gws.wscale : nullptr, gws.B));
    return BSSM_OK;
  }

  int fetch_loglik(hipStream_t st, double* out) {
    HIP_TRY(hipMemcpyAsync(out, loglik.p, fb.F * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int f = 0; f < fb.F; f++) out[f] += const_terms.empty() ? 0.0 : const_terms[f];
    return BSSM_OK;
  }
};

bool is_gaussian_type_ok(int gaussian, int model_type) {
  if (gaussian) return model_type >= 1 && model_type <= 3;  // ssm_ulg, bsm_lg, ar1_lg (R/model_type.R:2-8)
  return model_type >= 0 && model_type <= 4;                // ssm_mng, ssm_ung, bsm_ng, svm, ar1_ng
}

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
const char* bssm_amd_version(void) { return "bssm_amd 0.1 (gfx950)"; }
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
  if (model && (gaussian != 0) != (model->distribution == BSSM_GAUSSIAN))
    return fail(BSSM_ERR_INVALID, "gaussian flag and model->distribution disagree");
  return run_single_filter(model, nsim, seed, false, false, out);
}

int bssm_bsf_smoother(const bssm_ung_model* model, uint32_t nsim, uint32_t seed, int gaussian, int model_type,
                      bssm_filter_out* out) {
  if (gaussian && model_type == 0)
    return fail(BSSM_ERR_UNSUPPORTED, "Bootstrap particle smoothing for multivariate gaussian model is not yet supported. ");
  if (!is_gaussian_type_ok(gaussian, model_type)) return fail(BSSM_ERR_UNSUPPORTED, "model_type not handled by this entry point");
  if (model && (gaussian != 0) != (model->distribution == BSSM_GAUSSIAN))
    return fail(BSSM_ERR_INVALID, "gaussian flag and model->distribution disagree");
  return run_single_filter(model, nsim, seed, false, true, out);
}

int bssm_psi_smoother(const bssm_ung_model* model, uint32_t nsim, uint32_t seed, int model_type, bssm_filter_out* out) {
  if (!is_gaussian_type_ok(0, model_type)) return fail(BSSM_ERR_UNSUPPORTED, "model_type not handled by this entry point");
  return run_single_filter(model, nsim, seed, true, true, out);
}

// ---------------------------------------------------------------- built-in non-linear / SDE models
namespace {
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
                   double* params_only = nullptr) {
  if (!mod || !mod->y || !mod->theta) return fail(BSSM_ERR_INVALID, "model, y, theta must be non-NULL");
  const int id = mod->model_id, n = mod->n;
  if (id < bssm::FN_NLG_GROWTH || id > bssm::FN_USER_SDE) return fail(BSSM_ERR_UNSUPPORTED, "unknown model id");
  const bool user = id >= bssm::FN_USER_NLG;
  if (n < 1) return fail(BSSM_ERR_INVALID, "n must be positive");
  const int need_theta[] = {0, 3, 4, 2, 1, 3, 3, 0, 0}, need_known[] = {0, 6, 0, 0, 0, 0, 0, 0, 0};
  if (mod->n_theta < need_theta[id] || mod->n_known < need_known[id]) return fail(BSSM_ERR_INVALID, "theta / known_params too short for this model");
  if (mod->n_theta > 8 || mod->n_known > 8) return fail(BSSM_ERR_INVALID, "at most 8 theta / known_params");
  if (user && mod->user_slot < 1) return fail(BSSM_ERR_INVALID, "user-defined model without a loaded library (bssm_user_model_load)");
  if (id == bssm::FN_USER_NLG && (mod->n_states < 1 || mod->n_states > 4)) return fail(BSSM_ERR_UNSUPPORTED, "user-defined ssm_nlg models: n_states must be 1 ... 4");
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
  for (int q = 0; q < mod->n_theta; q++) fp[bssm::FP_THETA + q] = th[q];
  for (int q = 0; q < mod->n_known; q++) fp[bssm::FP_KNOWN + q] = kp[q];
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
    double RR[16] = {0}, Lc[16] = {0}, Li[16] = {0};
    const int kk = ext ? mod->n_etas : m;
    for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) {
      double acc = 0;
      for (int q = 0; q < kk; q++) acc += fp[bssm::FP_RMAT + i + MP * q] * fp[bssm::FP_RMAT + j + MP * q];
      RR[i + 4 * j] = acc;
    }
    int nz[4], cnt = 0;
    for (int j = 0; j < m; j++) if (RR[j + 4 * j] > 2.220446049250313e-16) nz[cnt++] = j;
    bool pd = true;
    for (int a = 0; a < cnt && pd; a++)
      for (int b2 = 0; b2 <= a; b2++) {
        double v = RR[nz[a] + 4 * nz[b2]];
        for (int q = 0; q < b2; q++) v -= Lc[a + 4 * q] * Lc[b2 + 4 * q];
        if (a == b2) { if (!(v > 0)) { pd = false; break; } Lc[a + 4 * a] = std::sqrt(v); }
        else Lc[a + 4 * b2] = v / Lc[b2 + 4 * b2];
      }
    for (int c = 0; c < cnt && pd; c++) {  // forward substitution column by column: Lc Li = I
      for (int a = c; a < cnt; a++) {
        double v = a == c ? 1.0 : 0.0;
        for (int q = c; q < a; q++) v -= Lc[a + 4 * q] * Li[q + 4 * c];
        Li[a + 4 * c] = v / Lc[a + 4 * a];
      }
    }
    for (int a = 0; a < cnt; a++) for (int c = 0; c <= a; c++)
      fp[bssm::FP_RLINV + nz[a] + MP * nz[c]] = pd ? Li[a + 4 * c] : std::nan("");
    if (mod->known_tv_params && mod->tv_rows > 0 && mod->tv_cols > 0 && have_device()) {  // BSSM_USER_TV snippets: matrix on the device
      double* dtv = nullptr;
      if (int rc = device_known_tv(mod->known_tv_params, mod->tv_rows, mod->tv_cols, &dtv)) return rc;
      uint64_t bits = (uint64_t)(uintptr_t)dtv;
      std::memcpy(&fp[bssm::FP_TVPTR], &bits, sizeof bits);
      fp[bssm::FP_TVROWS] = mod->tv_rows; fp[bssm::FP_TVCOLS] = mod->tv_cols;
    }
  }
  if (id >= bssm::FN_NLG_GROWTH && id <= bssm::FN_NLG_LINEAR_GAUSSIAN)  // shipped models: diagonal R, the psi weight reads 1 / R_j
    for (int j = 0; j < m; j++) fp[bssm::FP_RLINV + j + MP * j] = 1.0 / fp[bssm::FP_R + j];
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
    bool nz[4];
    double Lf[16] = {0};
    for (int j = 0; j < m; j++) nz[j] = mod->P1_ext[j + m * j] > thr;
    for (int j = 0; j < m; j++) {
      ft->init[j] = mod->a1_ext[j];
      if (!nz[j]) continue;
      double d = mod->P1_ext[j + m * j];
      for (int q = 0; q < j; q++) d -= Lf[j + 4 * q] * Lf[j + 4 * q];
      if (!(d > 0)) return fail(BSSM_ERR_INVALID, "P1 is not positive semi-definite");
      Lf[j + 4 * j] = std::sqrt(d);
      for (int i = j + 1; i < m; i++) {
        if (!nz[i]) continue;
        double v = mod->P1_ext[i + m * j];
        for (int q = 0; q < j; q++) v -= Lf[i + 4 * q] * Lf[j + 4 * q];
        Lf[i + 4 * j] = v / Lf[j + 4 * j];
      }
    }
    for (int c = 0; c < m; c++) for (int j = 0; j < m; j++) ft->init[MP + j + MP * c] = Lf[j + 4 * c];
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

// max_states: 4 for the psi-APF (any constant R with R R' positive definite on its non-zero diagonal, any P1), 2 for the EKF particle filter (its device code
// factors full 2 x 2 covariances per particle)
int to_fn_model(const bssm_fn_model* mod, bssm::FnModel* f, int max_states = 4) {
  if (!mod || !mod->y || !mod->theta) return fail(BSSM_ERR_INVALID, "model, y, theta must be non-NULL");
  const bool user = mod->model_id == bssm::FN_USER_NLG;
  if (!user && (mod->model_id < bssm::FN_NLG_GROWTH || mod->model_id > bssm::FN_NLG_LINEAR_GAUSSIAN))
    return fail(BSSM_ERR_UNSUPPORTED, "the psi-APF and the EKF particle filter are for ssm_nlg models");
  if (mod->n_theta > 8 || mod->n_known > 8) return fail(BSSM_ERR_INVALID, "at most 8 theta / known_params");
  f->id = mod->model_id; f->n = mod->n; f->m = mod->model_id == bssm::FN_NLG_GROWTH ? 2 : (user ? mod->n_states : 1);
  if (user) {
    if (mod->user_slot < 1 || mod->user_slot > 63) return fail(BSSM_ERR_INVALID, "user-defined model without a loaded library");
    const bssm::UserHostFns& h = g_user_host[mod->user_slot];
    const int ms = mod->n_states;
    if (ms < 1 || ms > max_states)
      return fail(BSSM_ERR_UNSUPPORTED, max_states == 2 ? "EKF particle filter of a user model: at most 2 states" : "psi-APF of a user model: at most 4 states");
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
    for (int q = 0; q < 16; q++) f->uRm[q] = 0.0;
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
  int rc = to_fn_model(mod, &f);
  if (rc) return rc;
  bssm::FilterTables dummy;
  int m = 1, k = 1;
  if ((rc = functor_tables(mod, &dummy, fpar, &m, &k))) return rc;  // fpar block (id, theta, H, R ...)
  for (int q = 0; q < 16; q++)
    if (std::isnan((*fpar)[bssm::FP_RLINV + q])) return fail(BSSM_ERR_INVALID, "psi-APF of a user model: R R' must be positive definite on its non-zero diagonal");
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
}  // namespace

namespace {
// EKF-proposal particle filter of a built-in ssm_nlg model: bootstrap-style tables, the initial proposal (att1, L1) as
// the initial table, and [a1 | P1 | att1 | L1] behind the functor parameters for the time-0 weights
int functor_ekpf_tables(const bssm_fn_model* mod, bssm::FilterTables* ft, std::vector<double>* fpar, int* m_out) {
  bssm::FnModel f;
  int rc = to_fn_model(mod, &f, 4);  // (round 3: up to 4 states, any constant R, any P1 -- as the psi-APF)
  if (rc) return rc;
  int m = 1, k = 1;
  if ((rc = functor_tables(mod, ft, fpar, &m, &k))) return rc;
  const int MP = bssm::pad_dim(m);
  if (mod->model_id == bssm::FN_USER_NLG && m <= 2) {  // the 1- / 2-state device code builds R R' from the diagonal of R
    for (int c = 0; c < m; c++) for (int j = 0; j < m; j++)
      if (j != c && f.uRm[j + m * c] != 0.0) return fail(BSSM_ERR_UNSUPPORTED, "EKF particle filter of a user model with 1 or 2 states: constant diagonal R");
  }
  double a1[4], P1[16], att1[4], L1[16];
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
  const int rc = to_fn_model(model, &f);
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
  if (bssm::opt().force_path == 2) db.use_grid = true;
  if (bssm::opt().force_path == 1 && (int)nsim <= cap) db.use_grid = false;
  if (db.use_grid) {
    int B, chunk;
    const size_t bytes = bssm::grid_workspace_bytes(db.fb, &B, &chunk);
    if (B > 256) return fail(BSSM_ERR_UNSUPPORTED, "nsim too large for the grid filter");
    HIP_TRY(db.ws.alloc(bytes));
    bssm::grid_workspace_carve(db.fb, db.ws.p, &db.gws);
  }
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
  if (n_states < 1 || n_states > 4) return fail(BSSM_ERR_UNSUPPORTED, "user-defined models: n_states must be 1 ... 4");
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
  for (const char* hdr : {"pf_filter_impl.h", "pf_grid_v2.h", "pf_functors.h", "pf_device.h", "pf_kernels.h", "pf_math.h", "pf_math_tables.h", "options.h", "tls_stream.h", "user_model_tu.hip"}) {
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
  std::string cmd = "( ", objs, logs;
  for (int part = 0; part < 6; part++) {
    const std::string obj = tmp + ".p" + std::to_string(part) + ".o", plog = log_tmp + ".p" + std::to_string(part);
    cmd += common + "-DBSSM_USER_PART=" + std::to_string(part) + " -c '" + dir + "/csrc/user_model_tu.hip' -o '" + obj + "' > '" + plog + "' 2>&1 & ";
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

namespace {
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
}  // namespace

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
  DevBuf fpar, ext, modes, y, scratch, mats, obs, init, amats, ok;
  // the set-up kernel runs asynchronously: the host thread that prepared the chunk records `ready` behind it and goes on
  // to the next chunk; the device thread waits for the event before the chunk's filters (finish() below)
  hipEvent_t ready = nullptr;
  StageBuf stage, okst;   // pinned input staging (alive until the copies have run) and the kernel's per-draw ok flags
  DevBuf in_a, in_b;      // ung / mng: packed draws and their transpose
  ~ChunkTables() { if (ready) (void)hipEventDestroy(ready); }
  // device thread: wait for the set-up kernel, mark failed draws, give the work arrays back
  int finish(std::vector<char>* bad) {
    if (!ready) return 0;
    if (hipEventSynchronize(ready) != hipSuccess) return -1;
    const int* okv = (const int*)okst.p;
    for (int q = 0; q < F; q++) if (!okv[q]) (*bad)[lo + q] = 1;  // non-finite Vt / Ct: the filter returns -Inf (src/model_ssm_nlg.cpp:988-990)
    scratch.release(); modes.release(); ext.release(); ok.release(); in_a.release(); in_b.release();
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
  std::vector<double> fpar;    // FP_SIZE per draw, or empty
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
  const std::shared_ptr<ChunkTables> ctabs = J.take_chunk(lo);  // non-null: the tables of [lo, hi) are already in HBM, in draw order
  if (ctabs && ctabs->finish(&J.bad)) return fail(BSSM_ERR_HIP, "psi set-up kernel failed");
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
        db.fb.fpar = ctabs->fpar.as<double>() + pos * bssm::FP_SIZE; db.fb.fpar_fstride = bssm::FP_SIZE;
        db.fb.user_slot = 0;
        db.fb.fn_id = (int)J.fpar[bssm::FP_ID];
        db.fb.amats = ctabs->amats.as<double>() + pos * ctabs->as; db.fb.amats_fstride = ctabs->as;
      } else if (!J.fpar.empty()) {
        HIP_TRY(st_fpar.alloc(F * bssm::FP_SIZE * sizeof(double)));
        double* hb = (double*)st_fpar.p;
        for (size_t f = 0; f < F; f++) std::memcpy(hb + f * bssm::FP_SIZE, J.fpar.data() + (size_t)order[pos + f] * bssm::FP_SIZE, bssm::FP_SIZE * sizeof(double));
        HIP_TRY(fparb.alloc(F * bssm::FP_SIZE * sizeof(double)));
        HIP_TRY(hipMemcpyAsync(fparb.p, hb, fparb.bytes, hipMemcpyHostToDevice, nullptr));
        db.fb.fpar = fparb.as<double>(); db.fb.fpar_fstride = bssm::FP_SIZE;
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
        if (db.use_grid) {
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
          // shard of it at 8 ranks: 292 = 256 + 36) costs a full round.  Those filters go through the step kernels instead (several
          // workgroups per filter, one launch per time step), behind the full rounds on the same stream: 36 filters of the C5 shape
          // 3.7 -> 2.5 ms alone, about 1 ms behind other work (profiles/r03_c5_tail_round.txt).  Not when a path is forced.
          // OPT-IN since round 4 (BSSM_AMD_TAIL_SPLIT=1): the two kernel families agree to reduction-order rounding only (a log-weight
          // to ~1e-15 relative, and through a moved ancestor a summary to 1e-6), and which draws land in the tail depends on the
          // draws per chunk, the shard layout and the CU count -- with the split on, a draw's result would depend on how many GPUs
          // the job ran on.  The default keeps every draw workgroup-resident: outputs bit-identical for any device list
          // (include/bssm_amd.h), at the price of a sparsely filled last round (one rank's 292-draw shard of C5: 8.2 -> 9.8 ms).
          int tail = 0;
          if ((size_t)(sub.MP + 1) * (size_t)N * sizeof(double) > 80 * 1024 && bssm::opt().force_path == 0 && bssm::opt().tail_split) {
            int dev = 0; hipDeviceProp_t prop;
            const int cus = hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
            const int r = sub.F % cus;
            if (r > 0 && r <= cus * 3 / 8) tail = r;
          }
          bssm::FilterBatch head = sub;
          head.F = sub.F - tail;
          if (head.F > 0) {
            HIP_TRY(bssm::launch_wg_filter(head, nullptr));
            HIP_TRY(bssm::launch_zero_dead_tail(head, true, nullptr));
          }
          if (tail > 0) {
            const size_t h0 = (size_t)head.F;
            bssm::FilterBatch tl = sub;
            tl.F = tail;
            tl.mats += db.fb.mats_fstride * h0; tl.obs += db.fb.obs_fstride * h0; tl.init += db.fb.init_fstride * h0;
            tl.loglik += h0; tl.dead_at += h0;
            if (tl.fpar) tl.fpar += tl.fpar_fstride * h0;
            if (tl.amats) tl.amats += tl.amats_fstride * h0;
            if (need_hist) { tl.alpha_hist += h0 * (size_t)(n + 1) * m * N; tl.indices += h0 * (size_t)n * N; }
            tl.stream0 = sub.stream0 + (uint32_t)h0;
            tail_ws.emplace_back();
            int B, chunk;
            HIP_TRY(tail_ws.back().alloc(bssm::grid_workspace_bytes(tl, &B, &chunk)));
            bssm::GridWorkspace gw;
            bssm::grid_workspace_carve(tl, tail_ws.back().p, &gw);
            HIP_TRY(bssm::launch_grid_filter(tl, gw, nullptr));
            HIP_TRY(bssm::launch_zero_dead_tail(tl, true, nullptr));
          }
        }
        f0 = f1;
      }
      // the previous chunk's summary has been running beside these filters: collect it while they finish
      if (J.pending) { const int prc = J.pending->drain(alphahat_i, Vt_i); J.pending.reset(); if (prc) return prc; }
      HIP_TRY(hipMemcpy(ll.data(), db.loglik.p, F * sizeof(double), hipMemcpyDeviceToHost));
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
  close();
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
  bool dev_full = psi && have_device() && m <= 8 && n < 65535 && job->is_type != 1 && (p0 == 1 ? job->models[0].distribution != BSSM_GAUSSIAN : true);
  if (bssm::opt().device_setup >= 0) dev_full = dev_full && bssm::opt().device_setup == 2;
  if (bssm::opt().generic_setup) dev_full = false;
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
      HIP_TRY(ct->scratch.alloc((p0 > 1 ? bssm::mng_setup_scratch_doubles(n, m, p0) : bssm::ung_setup_scratch_doubles(n, m)) * F * sizeof(double)));
      HIP_TRY(ct->mats.alloc(ct->ms * F * sizeof(double))); HIP_TRY(ct->obs.alloc(ct->os * F * sizeof(double)));
      HIP_TRY(ct->init.alloc(ct->is * F * sizeof(double)));
      HIP_TRY(ct->ok.alloc((size_t)F * sizeof(int)));
      const double t0 = now_ms();
      HIP_TRY(hipMemcpyAsync(din.p, h, E * F * sizeof(double), hipMemcpyHostToDevice, st));
      HIP_TRY(bssm::launch_transpose(din.as<double>(), din_t.as<double>(), F, (int)E, st));  // [F][E] -> [E][F]
      bssm::UngSetupDev b{};
      b.F = F; b.n = n; b.m = m; b.p = p0;
      b.Zs = L0.Zs; b.Ts = L0.Ts; b.Rs = L0.Rs; b.Ds = L0.Ds; b.Cs = L0.Cs;
      b.oZ = L0.oZ; b.oT = L0.oT; b.oR = L0.oR; b.oD = L0.oD; b.oC = L0.oC; b.oy = L0.oy; b.oa = L0.oa; b.oP = L0.oP;
      b.ou = L0.ou; b.ox = L0.ox; b.om = L0.om; b.oh = L0.oh;
      b.in = din_t.as<double>(); b.scratch = ct->scratch.as<double>();
      b.mats = ct->mats.as<double>(); b.obs = ct->obs.as<double>(); b.init = ct->init.as<double>(); b.ok = ct->ok.as<int>();
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
      const double per_draw = (double)(p0 > 1 ? bssm::mng_setup_scratch_doubles(n, m, p0) : bssm::ung_setup_scratch_doubles(n, m)) * sizeof(double);
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
  const int n = models[0].n, id = models[0].model_id, m = id == bssm::FN_NLG_GROWTH ? 2 : (id == bssm::FN_USER_NLG ? models[0].n_states : 1);
  IsCore J;
  J.n = n; J.m = m; J.k = m; J.pp = 1; J.psi = psi;
  J.first = first; J.is_type = is_type; J.output_type = output_type; J.nsim = nsim; J.seed = seed; J.counts = counts;
  J.tabs.resize(count);
  J.bad.assign(count, 0);
  J.const_add.assign(count, 0.0); J.sub.assign(count, 0.0);
  J.fpar.assign((size_t)count * bssm::FP_SIZE, 0.0);
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
      std::memcpy(J.fpar.data() + (size_t)i * bssm::FP_SIZE, fp.data(), bssm::FP_SIZE * sizeof(double));
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
      const size_t nf = (size_t)F * bssm::FP_SIZE, ne = (size_t)F * EXT, nm = (size_t)F * m * n;
      HIP_TRY(ct->stage.alloc((nf + ne + nm + n) * sizeof(double)));
      HIP_TRY(ct->okst.alloc((size_t)F * sizeof(int)));
      double* h = (double*)ct->stage.p;
      std::memcpy(h, J.fpar.data() + (size_t)lo * bssm::FP_SIZE, nf * sizeof(double));
      std::memcpy(h + nf, ext.data() + (size_t)lo * EXT, ne * sizeof(double));
      std::memcpy(h + nf + ne, modes + (size_t)m * n * lo, nm * sizeof(double));
      std::memcpy(h + nf + ne + nm, models[0].y, (size_t)n * sizeof(double));
      HIP_TRY(ct->fpar.alloc(nf * sizeof(double))); HIP_TRY(ct->ext.alloc(ne * sizeof(double)));
      HIP_TRY(ct->modes.alloc(nm * sizeof(double))); HIP_TRY(ct->y.alloc((size_t)n * sizeof(double)));
      HIP_TRY(ct->scratch.alloc(bssm::fn_setup_scratch_doubles(n, m) * F * sizeof(double)));
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
      if (count > 4096 || bssm::opt().trace) HIP_TRY(hipStreamSynchronize(st));
      if (bssm::opt().trace)
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
  const bool dev_setup = psi && device_setup_enabled(n_threads) && have_device();
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
        if (to_fn_model(&models[i], &fms[i])) return g_err;
        if (functor_tables(&models[i], &dummy, &fp, &mm_, &kk_)) return g_err;
        std::memcpy(J.fpar.data() + (size_t)i * bssm::FP_SIZE, fp.data(), bssm::FP_SIZE * sizeof(double));
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
    std::memcpy(J.fpar.data() + (size_t)i * bssm::FP_SIZE, fp.data(), bssm::FP_SIZE * sizeof(double));
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
      std::memcpy(J.fpar.data() + (size_t)i * bssm::FP_SIZE, fp.data(), bssm::FP_SIZE * sizeof(double));
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
    if (m > 8) return fail(BSSM_ERR_UNSUPPORTED, "summaries of the plan's particle system: m <= 8");
    DevBuf d_b, d_mean, d_cov;
    HIP_TRY(d_b.alloc((size_t)N * sizeof(uint32_t)));
    HIP_TRY(d_mean.alloc((size_t)(n + 1) * m * sizeof(double)));
    HIP_TRY(d_cov.alloc((size_t)(n + 1) * m * m * sizeof(double)));
    HIP_TRY(bssm::launch_smooth_summary(hist, idx, d_b.as<uint32_t>(), d_mean.as<double>(), d_cov.as<double>(), 1, n, m, N, st));
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
  const int mult = fn == 4 ? 4 : (fn >= 2 ? 2 : 1);
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
