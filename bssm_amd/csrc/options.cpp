// options.cpp -- see options.h
#include "options.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include <hip/hip_runtime.h>
#include <map>
#include <vector>

#include "tls_stream.h"

namespace bssm {
// process-wide pool of idle HIP streams (tls_stream.h): a stream a thread no longer needs waits here for the next thread
namespace {
std::mutex g_stream_pool_mu;
std::map<std::pair<int, int>, std::vector<hipStream_t>>& stream_pool() {
  static auto* pool = new std::map<std::pair<int, int>, std::vector<hipStream_t>>();  // (never destroyed: threads may exit after main)
  return *pool;
}
}  // namespace
hipStream_t stream_pool_take(int dev, int kind) {
  std::lock_guard<std::mutex> lk(g_stream_pool_mu);
  auto& v = stream_pool()[{dev, kind}];
  if (v.empty()) return nullptr;
  hipStream_t s = v.back();
  v.pop_back();
  return s;
}
void stream_pool_give(int dev, int kind, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_stream_pool_mu);
  stream_pool()[{dev, kind}].push_back(s);
}

// The runtime gives a stream its hardware queue at the stream's first use, and which queues of a process run side by side turned out to depend
// on the order in which they came to be.  A post-correction call that was the process's FIRST GPU work overlapped the tail of a chunk with its
// last round (293 draws of C5: 5.8 ms); the same call behind a full-output filter batch -- whose streams came first -- did not (13 ms: slower
// than the tail BEHIND the round, 7.2 ms), whatever the number of hardware queues allowed, and a stream with a queue of its own (CU-masked)
// was worse still (profiles/r05_tail_queue.txt).  So the library brings its streams into being in ONE order, once, at the first entry-point
// call: the null stream, two plain streams (the side launch sequence and the output copies of the filter entry points take them from the pool),
// the tail's stream, the set-up stream (highest priority) -- one tiny operation each.  The order is the one of five tried that keeps BOTH overlaps:
// with the set-up stream or the tail's in front of the plain ones the drop-in filter call (2.6 GB of outputs streaming out beside the steps) went
// 55 -> 63-65 ms and the psi smoother call 14.8 -> 18.2 ms.
// Round 6: once PER DEVICE (it was once per process, on whichever device was current at the first entry-point call: the shard threads of
// a multi-device call -- api_postcorrect.cpp: run_multi_device, suggest_n_multi -- met devices 1 ... 7 with their streams created on demand).
// Every entry point passes here (have_device()), and every shard thread after its hipSetDevice.
//
// Round 6, second half: the order alone is not enough -- a host process that created streams of its own BEFORE the library's first call
// (16 torch streams: profiles/r06_tail_queue.txt) left the tail's stream in an arrangement where it ran BEHIND the null stream's round
// (293 draws of C5: 7.1 ms instead of 5.6).  So the stream reserved for the tail is now MEASURED: while a spinning kernel holds every CU
// on the null stream for ~0.5 ms, a chain of 16 dependent one-thread kernels goes through the candidate; the candidate is kept when the
// chain starts and ends inside the spinner's lifetime.  A candidate that fails joins the plain pool and the next one is created (up to
// eight); when none passes, tail_overlap_available() says so and the post-correction driver steps the tail behind the round (the safe
// 7.2 ms) instead of beside it.  ~1 ms per device, once.  BSSM_AMD_TAIL_CALIBRATE=0 skips the measurement (the order alone, as in round 5).
namespace {
constexpr int kMaxDev = 64;
std::atomic<int> g_tail_overlap[kMaxDev];   // 0 unknown / not calibrated (assume yes), 1 measured yes, 2 measured no
std::atomic<int> g_tail_tried[kMaxDev];

__global__ void prewarm_spin_kernel(unsigned long long ticks, unsigned long long* out) {
  extern __shared__ double spin_lds[];   // (a large dynamic allocation: one workgroup per CU, like a round of workgroup-resident filters)
  const unsigned long long t0 = wall_clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t0;
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
  if (threadIdx.x == 0) { spin_lds[0] = 0.0; atomicMax(&out[1], wall_clock64()); }
}
__global__ void prewarm_chain_kernel(unsigned long long* out, int k) {
  if (threadIdx.x == 0) out[k] = wall_clock64();
}

// does a chain of dependent launches on `cand` run BESIDE a kernel that fills the chip on the null stream?
bool stream_overlaps_null(hipStream_t cand, unsigned long long* d_out, int cus, double clock_khz) {
  constexpr int kChain = 16;
  unsigned long long h[2 + kChain] = {0};
  if (hipMemset(d_out, 0, sizeof h) != hipSuccess) return false;
  const unsigned long long ticks = (unsigned long long)(clock_khz * 0.5);   // 0.5 ms
  hipLaunchKernelGGL(prewarm_spin_kernel, dim3(cus), dim3(256), 96 * 1024, nullptr, ticks, d_out);
  for (int k = 0; k < kChain; k++) hipLaunchKernelGGL(prewarm_chain_kernel, dim3(1), dim3(64), 0, cand, d_out, 2 + k);
  if (hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); return false; }
  if (hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return false;
  const unsigned long long spin_begin = h[0], spin_end = h[1], first = h[2], last = h[2 + kChain - 1];
  return first > spin_begin && last < spin_end && last >= first;
}
}  // namespace

bool tail_overlap_available() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return true;
  return g_tail_overlap[dev].load(std::memory_order_acquire) != 2;
}
int tail_overlap_state(int* candidates_tried) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return 0;
  if (candidates_tried) *candidates_tried = g_tail_tried[dev].load();
  return g_tail_overlap[dev].load();
}

void prewarm_streams() {
  static std::once_flag once[kMaxDev];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return; }
  if (dev < 0 || dev >= kMaxDev) return;
  std::call_once(once[dev], [dev] {
    if (!opt().prewarm_streams) return;
    void* p = nullptr;
    if (hipMalloc(&p, 256) != hipSuccess) { (void)hipGetLastError(); return; }
    // the order: h = set-up stream (highest priority), n = the null stream, t = the tail's stream, p = a plain stream for the pool
    // (side launch sequence / output copies of the filter entry points); BSSM_AMD_PREWARM_ORDER overrides it (measurements only)
    const char* order = std::getenv("BSSM_AMD_PREWARM_ORDER");
    if (!order || !*order) order = "n,p,p,t,h";
    std::vector<TlsStream> made(24);
    int used = 0, tail_at = -1;
    for (const char* c = order; *c && used < 16; c++) {
      hipStream_t st = nullptr;
      bool null_stream = false;
      if (*c == 'h') st = made[used++].get(true);
      else if (*c == 't') { tail_at = used; st = made[used++].get(false, true); }
      else if (*c == 'p') st = made[used++].get(false, false);
      else if (*c == 'n') null_stream = true;
      else continue;
      if (st || null_stream) (void)hipMemsetAsync((char*)p + 64 * (used % 4), 0, 8, st);
    }
    (void)hipDeviceSynchronize();
    const char* cal = std::getenv("BSSM_AMD_TAIL_CALIBRATE");
    if (tail_at >= 0 && made[tail_at].s && !(cal && cal[0] == '0')) {
      int cus = 256, khz = 100000;
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev);
      (void)hipFuncSetAttribute((const void*)prewarm_spin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
      unsigned long long* d_out = nullptr;
      if (hipMalloc((void**)&d_out, 64 * sizeof(unsigned long long)) == hipSuccess) {
        int tried = 1;
        bool ok = stream_overlaps_null(made[tail_at].s, d_out, cus, khz);
        while (!ok && tried < 8 && used < (int)made.size()) {
          made[tail_at].hi = 0;                    // the failed candidate becomes a plain pool stream
          tail_at = used;
          hipStream_t st = made[used++].get(false, true);
          if (!st) break;
          // (a stream taken from the reserved pool of an earlier process state cannot occur here: this runs once per device, first)
          tried++;
          ok = stream_overlaps_null(st, d_out, cus, khz);
        }
        if (!ok && tail_at >= 0) made[tail_at].hi = 0;
        g_tail_tried[dev].store(tried);
        g_tail_overlap[dev].store(ok ? 1 : 2, std::memory_order_release);
        if (opt().trace) std::fprintf(stderr, "[bssm_amd] device %d: tail stream %s after %d candidate(s)\n", dev, ok ? "runs beside the null stream" : "NOT concurrent with the null stream: tails go behind their rounds", tried);
        (void)hipFree(d_out);
      }
      (void)hipGetLastError();
    }
    for (int q = used - 1; q >= 0; q--) made[q].drop();   // (the pool hands out the last one given first: the first plain stream made is the first taken)
    (void)hipFree(p);
    (void)hipGetLastError();
  });
}

namespace {

int env_int(const char* name, int dflt) {
  const char* e = std::getenv(name);
  return (e && *e) ? std::atoi(e) : dflt;
}
bool env_set(const char* name) { return std::getenv(name) != nullptr; }
int env_tri(const char* name) {  // -1 unset, 0 when the value starts with '0', else 1
  const char* e = std::getenv(name);
  return !e ? -1 : (e[0] != '0' ? 1 : 0);
}

Options read_env() {
  Options o;
  if (const char* e = std::getenv("BSSM_AMD_CACHE_GB")) o.cache_gb = std::atof(e);
  o.copy_threads = env_int("BSSM_AMD_COPY_THREADS", 0);
  if (const char* e = std::getenv("BSSM_AMD_USER_CACHE")) o.user_cache = e;
  o.is_chunk = env_int("BSSM_AMD_IS_CHUNK", 0);
  if (const char* e = std::getenv("BSSM_AMD_GATHER")) o.gather = !std::strcmp(e, "copy") ? 1 : (!std::strcmp(e, "host") ? 2 : 0);
  o.trace = env_set("BSSM_AMD_TRACE") ? std::max(1, env_int("BSSM_AMD_TRACE", 1)) : 0;
  { const char* e = std::getenv("BSSM_AMD_TAIL_SPLIT"); o.tail_split = !(e && e[0] == '0'); }  // on unless BSSM_AMD_TAIL_SPLIT=0
  if (const char* e = std::getenv("BSSM_AMD_TAIL_MAX_EIGHTHS")) o.tail_max_eighths = std::max(0, std::min(8, std::atoi(e)));
  if (const char* e = std::getenv("BSSM_AMD_TAIL_CONCURRENT")) o.tail_concurrent = std::atoi(e) != 0;
  if (const char* e = std::getenv("BSSM_AMD_PREWARM_STREAMS")) o.prewarm_streams = std::atoi(e) != 0;
  if (const char* e = std::getenv("BSSM_AMD_FORCE_PATH")) o.force_path = !std::strcmp(e, "wg") ? 1 : (!std::strcmp(e, "grid") ? 2 : (!std::strcmp(e, "row") ? 3 : 0));
  o.grid_ppt = env_int("BSSM_AMD_GRID_PPT", 0);
  o.grid_v2_wx = env_tri("BSSM_AMD_GRID_V2_WX");
  o.grid_v2_exact = env_tri("BSSM_AMD_GRID_V2_EXACT");
  o.grid_streams = env_int("BSSM_AMD_GRID_STREAMS", 0);
  o.xcd_remap = env_int("BSSM_AMD_XCD_REMAP", -1);
  o.wg_wide = env_tri("BSSM_AMD_WG_WIDE");
  if (const char* e = std::getenv("BSSM_AMD_DEVICE_SETUP")) o.device_setup = e[0] == '2' ? 2 : (e[0] != '0' ? 1 : 0);
  o.generic_setup = env_set("BSSM_AMD_GENERIC_SETUP");
  o.mv_sequential = env_set("BSSM_AMD_MV_SEQUENTIAL");
  o.mng_lane_per_draw = env_set("BSSM_AMD_MNG_LANE_PER_DRAW");
  o.defer_summary = env_set("BSSM_AMD_DEFER_SUMMARY");
  if (const char* e = std::getenv("BSSM_AMD_SPARSE")) o.sparse = e[0] == '0' ? 0 : (e[0] == 'a' ? 2 : 1);
  if (const char* e = std::getenv("BSSM_AMD_NO_SPARSE")) { if (e[0] != '0') o.sparse = 0; }
  o.setup_interleaved = env_set("BSSM_AMD_SETUP_INTERLEAVED");
  o.factors_full = env_set("BSSM_AMD_FACTORS_FULL");
  o.setup_rows16 = env_set("BSSM_AMD_SETUP_ROWS16");
  o.single_device_setup = env_int("BSSM_AMD_SINGLE_DEVICE_SETUP", 0) != 0;
  o.host_factor_threads = std::max(0, env_int("BSSM_AMD_HOST_FACTOR_THREADS", 0));
  o.xcd_persist = env_tri("BSSM_AMD_XCD_PERSIST");   // (acts in -DBSSM_XCD_PERSIST builds only)
#ifdef BSSM_PROBES
  o.probe_no_weights = env_set("BSSM_AMD_PROBE_NO_WEIGHTS");
  o.no_outpipe = env_set("BSSM_AMD_NO_OUTPIPE");
  o.side_prio = env_set("BSSM_AMD_SIDE_PRIO");
  o.ss_global = env_set("BSSM_AMD_SS_GLOBAL");
  o.wg_no_exact = env_set("BSSM_AMD_WG_NO_EXACT");
  o.is_big_chunk = env_int("BSSM_AMD_IS_BIG_CHUNK", 0);
  o.xcd_persist_dbg = env_int("BSSM_AMD_XCD_PERSIST_DBG", 0);
#endif
  return o;
}

std::mutex g_mu;
std::atomic<const Options*> g_cur{nullptr};

}  // namespace

const Options& opt() {
  const Options* p = g_cur.load(std::memory_order_acquire);
  if (p) return *p;
  std::lock_guard<std::mutex> lk(g_mu);
  p = g_cur.load(std::memory_order_acquire);
  if (!p) { p = new Options(read_env()); g_cur.store(p, std::memory_order_release); }
  return *p;
}

void reload_options() {
  std::lock_guard<std::mutex> lk(g_mu);
  // snapshots are never freed: a call in flight on another thread may still hold a reference (a few hundred bytes per reload)
  g_cur.store(new Options(read_env()), std::memory_order_release);
}

namespace { struct AtLoad { AtLoad() { (void)opt(); } } g_at_load; }

}  // namespace bssm
