// api_internal.h -- what the translation units of the C ABI share (api_filters.cpp, api_postcorrect.cpp, api_plan.cpp): error
// state, device arenas and buffers, the output pipe, a batch of filters on the device.  Everything here is inline / a type:
// one instance of every variable for the whole library.
// There is no CPU compute path in this library: every filter runs on the HIP device or fails.
#pragma once
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

namespace bssm_api {

inline thread_local std::string g_err;
inline bssm::UserHostFns g_user_host[64];  // host entry points of the loaded user-model libraries, by registry slot
inline thread_local uint32_t g_call_stream0 = 0;  // Philox stream of the next single-filter call on this thread (bssm_fn_smoother_stream)
inline thread_local double g_is_timing[2] = {0.0, 0.0};  // last IS post-correction on this thread: host set-up ms, device phase ms
// the same call split further: per-draw host packing (prepare / post / finish), the per-chunk device set-up call (uploads,
// set-up kernel launch and, where the pipeline waits for it, the wait), the filter thread's busy time, wall time of the call
inline thread_local double g_is_timing4[4] = {0.0, 0.0, 0.0, 0.0};
inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

inline int fail(int code, const std::string& msg) { g_err = msg; return code; }


#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      g_err = std::string(#expr) + ": " + hipGetErrorString(e_);                               \
      return BSSM_ERR_HIP;                                                                     \
    }                                                                                          \
  } while (0)

inline bool have_device() {
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess) return false;
  if (cnt > 0) bssm::prewarm_streams();
  return cnt > 0;
}
inline int no_device() {
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
inline DeviceArena& arena() { static DeviceArena* a = new DeviceArena; return *a; }  // leaked on purpose: no hipFree at exit
// pinned staging for host -> device table uploads (pageable copies run at a few GB/s and block the launch thread)
inline DeviceArena& stage_arena() {
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
inline void parallel_for(int count, int threads, F fn) {
  threads = std::max(1, std::min(threads, count));
  if (threads == 1) { for (int i = 0; i < count; i++) fn(i); return; }
  std::atomic<int> next(0);
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; t++)
    pool.emplace_back([&]() { for (int i = next++; i < count; i = next++) fn(i); });
  for (auto& th : pool) th.join();
}


// Round 6: the descriptor of a structured transition (pf_big.h: sparse_apply; FilterBatch::sp) for a batch of bootstrap filters with
// 17 ... 64 states, or an empty vector when the batch does not qualify: time-invariant T shared by every filter of the batch, all of its
// entries +-1 with at most two per column (the companion matrices bsm_ng / bsm_lg build for trend + seasonal, R/models.R:995-1060), at
// most eight rows that are not a plain copy, one observation series, at most eight noise terms.  BSSM_AMD_NO_SPARSE=1 keeps the dense
// matrix-core transition (comparison switch).
inline std::vector<int> sparse_descriptor(const std::vector<const bssm::FilterTables*>& tabs, int n, int m, int k, int p, int psi, int* cnz_out) {
  std::vector<int> none;
  const int MP = tabs[0]->MP, KP = tabs[0]->KP;
  if (psi != 0 || p != 1 || MP < 32 || MP > 64 || tabs[0]->mats_tstride != 0 || k > 8 || m > 64) return none;
  if (bssm::opt().sparse == 0) return none;
  const double* T0 = tabs[0]->mats.data();
  int row_cnt[64] = {0}, noise[64] = {0}, zflag[64] = {0};
  size_t nnz = 0;
  for (int c = 0; c < m; c++)
    for (int r = 0; r < m; r++) {
      const double v = T0[r + (size_t)MP * c];
      if (v == 0.0) continue;
      if (v != 1.0 && v != -1.0) return none;
      row_cnt[r]++; nnz++;
    }
  if (nnz > (size_t)4 * m) return none;   // (not a structured matrix: the matrix cores do better)
  int cnz = 0;
  const size_t ostride = (size_t)bssm::ROW_HDR + bssm::SER_HDR + MP;
  for (const bssm::FilterTables* ft : tabs) {
    if (std::memcmp(ft->mats.data(), T0, sizeof(double) * (size_t)MP * MP) != 0) return none;   // (bsm_ng: T does not depend on theta)
    const double* R = ft->mats.data() + (size_t)MP * MP;
    const double* Cv = R + (size_t)MP * KP;
    for (int r = 0; r < m; r++) {
      for (int kk = 0; kk < KP; kk++) if (R[r + (size_t)MP * kk] != 0.0) noise[r] = 1;
      if (Cv[r] != 0.0) cnz = 1;
    }
    for (int t = 0; t < n; t++) {
      const double* zr = ft->obs.data() + ostride * t + bssm::ROW_HDR + bssm::SER_HDR;
      for (int r = 0; r < m; r++) if (zr[r] != 0.0) zflag[r] = 1;
    }
  }
  // rows that are a plain copy of the component in front of them: x'[r] = x[r - 1]; the longest run of them is the shift block
  bool shift[64] = {false};
  for (int r = 1; r < m; r++) shift[r] = row_cnt[r] == 1 && !noise[r] && T0[r + (size_t)MP * (r - 1)] == 1.0;
  int best_lo = 1, best_hi = 0;
  for (int r = 1; r < m;) {
    if (!shift[r]) { r++; continue; }
    int e = r;
    while (e + 1 < m && shift[e + 1]) e++;
    if (e - r > best_hi - best_lo) { best_lo = r; best_hi = e; }
    r = e + 1;
  }
  std::vector<int> d((size_t)bssm::SP_WORDS, 0);
  d[0] = best_lo; d[1] = best_hi;
  unsigned long long zmask = 0;
  int n_slots = 0;
  for (int r = 0; r < m; r++) {
    if (r >= best_lo && r <= best_hi) { if (zflag[r]) zmask |= 1ull << r; continue; }
    if (n_slots == bssm::SP_SLOTS) return none;
    unsigned long long P = 0, Nn = 0;
    for (int c = 0; c < m; c++) {
      const double v = T0[r + (size_t)MP * c];
      if (v == 1.0) P |= 1ull << c; else if (v == -1.0) Nn |= 1ull << c;
    }
    int* w = d.data() + bssm::SP_HDR + bssm::SP_SLOT_WORDS * n_slots;
    w[0] = r | 0x100 | (zflag[r] ? 0x200 : 0) | (noise[r] ? 0x400 : 0);
    w[1] = (int)(unsigned)(P & 0xffffffffull); w[2] = (int)(unsigned)(P >> 32);
    w[3] = (int)(unsigned)(Nn & 0xffffffffull); w[4] = (int)(unsigned)(Nn >> 32);
    n_slots++;
  }
  d[2] = (int)(unsigned)(zmask & 0xffffffffull); d[3] = (int)(unsigned)(zmask >> 32);
  *cnz_out = cnz;
  return d;
}

// ------------------------------------------------------------------ a batch of filters on the device
struct DeviceBatch {
  bssm::FilterBatch fb{};
  DevBuf mats, obs, init, hist, indices, weights, loglik, dead, ws, fparbuf, amatbuf, spbuf;
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
    fb.sp = nullptr; fb.sp_cnz = 0;
    {  // a structured transition (bsm_ng with a seasonal component: pf_big.h, sparse_apply)?
      int cnz = 0;
      const std::vector<int> sp = sparse_descriptor(tabs, n, m, k, p, psi, &cnz);
      if (!sp.empty()) {
        HIP_TRY(spbuf.alloc(sp.size() * sizeof(int)));
        HIP_TRY(hipMemcpy(spbuf.p, sp.data(), sp.size() * sizeof(int), hipMemcpyHostToDevice));
        fb.sp = spbuf.as<int>(); fb.sp_cnz = cnz;
      }
    }
    return configure(keep_hist, path_flags, keep_weights);
  }
  // tables that are already in device memory (built there by fn_setup_kernel): F filters starting at the given pointers
  int attach(const double* d_mats, size_t ms, int tstride, const double* d_obs, size_t os, const double* d_init, size_t is,
             int F, int n, int m, int k, int p, int N, int psi, int MP, int KP) {
    fb.n = n; fb.m = m; fb.k = k; fb.p = p; fb.MP = MP; fb.KP = KP; fb.N = N; fb.F = F; fb.psi = psi;
    fb.mats = d_mats; fb.mats_fstride = ms; fb.mats_tstride = tstride;
    fb.obs = d_obs; fb.obs_fstride = os;
    fb.init = d_init; fb.init_fstride = is;
    fb.sp = nullptr; fb.sp_cnz = 0;
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
    use_grid = bssm::prefer_step_shape(fb, use_grid);
    if (bssm::opt().force_path == 2) use_grid = true;
    if (bssm::opt().force_path == 1) use_grid = false;
    if (path_flags & BSSM_PLAN_FORCE_GRID) use_grid = true;
    if (path_flags & BSSM_PLAN_FORCE_WG) use_grid = false;
    if (!use_grid && N > cap) return fail(BSSM_ERR_INVALID, "nsim exceeds the workgroup-resident capacity");
    if (bssm::opt().force_path == 3 && bssm::row_filter_supported(fb)) use_grid = true;  // test hook: the row family (pf_row_step.h)
    if (use_grid) return alloc_step_workspace();
    return BSSM_OK;
  }
  // workspace of the one-launch-per-step families: the row family when BSSM_AMD_FORCE_PATH=row asks for it and the batch has such
  // kernels (pf_row_step.h; otherwise it is reached through the tail split of a post-correction chunk only), else the step kernels
  int alloc_step_workspace() {
    // (states of more than 16 dimensions -- pf_big.h -- have no other step shape than the row family: up to 102 400 particles since round 6;
    // the reference's front-end stops at 1e5, R/check_arguments.R:407-418)
    if (fb.MP >= 32 && !bssm::row_filter_supported(fb)) return fail(BSSM_ERR_UNSUPPORTED, "nsim above 102400 is not supported for states of more than 16 dimensions");
    // (a user model compiled in process takes the row family wherever it reaches -- one kernel variant to compile instead of one per particles-per-lane)
    if ((bssm::opt().force_path == 3 || fb.MP >= 32 || bssm::user_slot_rows_only(fb.user_slot)) && bssm::row_filter_supported(fb)) {
      HIP_TRY(ws.alloc(bssm::row_workspace_bytes(fb)));
      bssm::row_workspace_carve(fb, ws.p, &gws);
      return BSSM_OK;
    }
    int B, chunk;
    const size_t bytes = bssm::grid_workspace_bytes(fb, &B, &chunk);
    if (B > 256) return fail(BSSM_ERR_UNSUPPORTED, "nsim too large for the step kernels (max 262144; the reference's front-end stops at 1e5)");
    HIP_TRY(ws.alloc(bytes));
    bssm::grid_workspace_carve(fb, ws.p, &gws);
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
    // the step-kernel workspace was carved before the batch was known to be a functor model: its particles per lane follow the
    // kernel family (pf_kernels.hip: grid_workspace_bytes brings a functor filter under the round-2 kernel's 128 particle blocks)
    if (use_grid) return alloc_step_workspace();
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

inline bool is_gaussian_type_ok(int gaussian, int model_type) {
  if (gaussian) return model_type >= 1 && model_type <= 3;  // ssm_ulg, bsm_lg, ar1_lg (R/model_type.R:2-8)
  return model_type >= 0 && model_type <= 4;                // ssm_mng, ssm_ung, bsm_ng, svm, ar1_ng
}

}  // namespace bssm_api
