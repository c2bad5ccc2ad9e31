// pf_kernels.hip -- launch dispatch plus the smoothing / summary / layout kernels.
// The filter kernels themselves are templates in pf_filter_impl.h, instantiated per padded
// state dimension in pf_inst_mp*.hip.
#include <algorithm>
#include <map>
#include <mutex>
#include <vector>
#include <atomic>
#include <cstdio>
#include <cstdlib>

// analysis builds (-DBSSM_SETUP_STAMPS): wall-clock stamps (100 MHz) of workgroup 0, lane 0 at the phase boundaries of the set-up
// kernels, printed by the kernel itself (scripts/setup_phase_probe.sh)
#ifdef BSSM_SETUP_STAMPS
#include <hip/hip_runtime.h>
__device__ unsigned long long g_setup_stamps[8];
#define LG_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_setup_stamps[i] = wall_clock64(); } while (0)
#endif
#include "lg_smoother.h"
#include "lg_smoother_rows.h"
#include "lg_factors_rows.h"
#include "lg_smoother_wave.h"
#include "pf_filter_impl.h"
#include "tls_stream.h"

namespace bssm {

static thread_local char g_last_kernel[96] = "";
void note_filter_kernel(const char* name) { std::snprintf(g_last_kernel, sizeof g_last_kernel, "%s", name); }
const char* last_filter_kernel() { return g_last_kernel; }

int wg_filter_capacity(int MP) {
  if (MP >= 32) return big_wg_capacity(MP);  // pf_big.h: one wave per row of 64 particles, at most 512 threads
  // 160 KiB LDS per CU: (MP + 1) doubles per particle + reduction slots; 512 threads x up to 16 particles
  // (8 for padded state dimensions above 2)
  const size_t avail = 160 * 1024 - (2 * CANON_MAX_ROWS + MP * MP * 2 + MP) * sizeof(double);
  const size_t cap = avail / ((size_t)(MP + 1) * sizeof(double));
  for (int c = (MP <= 2 ? 8192 : 4096); c >= 512; c >>= 1) if ((size_t)c <= cap) return c;
  return (int)(cap / 64 * 64);
}

// ONE side stream per host thread for the whole library (streams share a small number of hardware queues: a pair per
// kernel variant ended up serialised against each other)
static hipStream_t library_side_stream() {
  static thread_local TlsStream side;
  static const bool side_hi = opt().side_prio;  // measurement: the side sequence at the highest stream priority
  return side.get(side_hi);
}

// Round 6: a launch BESIDE the sequence of `stream` -- on this thread's side stream, behind what `stream` holds now; `stream` goes on and joins it
// later (side_join).  The events are the thread's own (created once).  Used by the set-up of 16 and more padded states: the t = 0 decomposition of
// every draw (one Jacobi per draw, a 1.3 ms latency chain that fills half a CU's workgroup slots) runs beside the thin factors of the other time points.
namespace {
// (the event pairs live in a process-wide pool keyed by device and are never destroyed: a thread gives its pair back when it ends -- the per-call
// threads of a post-correction come and go -- and nothing calls into the HIP runtime from a thread_local destructor at process exit; as the streams of
// tls_stream.h)
struct SideEvents { hipEvent_t fork = nullptr, join = nullptr; };
std::mutex& side_pool_mu() { static auto* m = new std::mutex; return *m; }
std::map<int, std::vector<SideEvents>>& side_pool() { static auto* p = new std::map<int, std::vector<SideEvents>>(); return *p; }
struct SideFork {
  hipEvent_t fork = nullptr, join = nullptr;
  int dev = -1;   // the device the events belong to (a pool thread may serve another device later: it takes a pair of that device)
  void give_back() {
    if (!fork) return;
    std::lock_guard<std::mutex> lk(side_pool_mu());
    side_pool()[dev].push_back(SideEvents{fork, join});
    fork = join = nullptr; dev = -1;
  }
  ~SideFork() { give_back(); }
  hipError_t init() {
    int cur = 0;
    hipError_t e = hipGetDevice(&cur);
    if (e != hipSuccess) return e;
    if (fork && dev == cur) return hipSuccess;
    give_back();
    {
      std::lock_guard<std::mutex> lk(side_pool_mu());
      auto& v = side_pool()[cur];
      if (!v.empty()) { fork = v.back().fork; join = v.back().join; v.pop_back(); dev = cur; return hipSuccess; }
    }
    hipEvent_t f = nullptr, j = nullptr;
    if ((e = hipEventCreateWithFlags(&f, hipEventDisableTiming)) != hipSuccess) return e;
    if ((e = hipEventCreateWithFlags(&j, hipEventDisableTiming)) != hipSuccess) { (void)hipEventDestroy(f); return e; }
    fork = f; join = j; dev = cur;
    return hipSuccess;
  }
};
SideFork& side_fork() { static thread_local SideFork f; return f; }
// side stream waits for `stream`; returns the side stream (nullptr: no side stream -- the caller launches on `stream` itself)
hipStream_t side_begin(hipStream_t stream) {
  hipStream_t side = library_side_stream();
  if (!side || side == stream || side_fork().init() != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  if (hipEventRecord(side_fork().fork, stream) != hipSuccess || hipStreamWaitEvent(side, side_fork().fork, 0) != hipSuccess) {
    (void)hipGetLastError();   // (no fork: the caller launches on `stream`; the failure must not surface as the launch sequence's error)
    return nullptr;
  }
  return side;
}
hipError_t side_join(hipStream_t side, hipStream_t stream) {
  hipError_t e = hipEventRecord(side_fork().join, side);
  if (e != hipSuccess) return e;
  return hipStreamWaitEvent(stream, side_fork().join, 0);
}
}  // namespace

size_t grid_workspace_bytes(const FilterBatch& b, int* B_out, int* chunk_out) {
  // One particle per thread keeps a single filter's dependent chain shortest and is best until the batch oversubscribes
  // the chip about three times (workgroups of 256 at one particle per thread: w1 = F * ceil(N / 256); the chip holds 1024).
  // Beyond that more particles per thread amortise the per-wave work of a step -- the partial combine and the
  // wave-cooperative ancestor search -- over more particles: two from w1 >= 3072, four from w1 >= 6144 (measured on the
  // metric model and on m = 1, full output and log-likelihood only: scripts/ppt_probe.py; 24 filters of 65 536 particles:
  // 2.79e10 / 2.96e10 / 3.15e10 particle-steps/s at 1 / 2 / 4).
  // (only for states of at most 4 doubles without functor parameters: larger live state costs occupancy, and the
  // fp64-MFMA transition of the 8-dimensional state needs one particle per lane)
  const size_t w1 = (size_t)b.F * ((b.N + GRID_NT - 1) / GRID_NT);
  int ppt = 1;
  if (b.MP <= 4 && !b.fpar) {
    // The round-2 kernel (one ancestor search per wave; at most 128 particle blocks per filter) wins wherever more particles per
    // lane are what make a filter eligible for it.  65 536 particles (256 blocks at one particle per lane, 128 at two), full
    // output, particle-steps/s at 1 / 2 / 4 particles per lane (scripts/ppt_sweep.py): m = 4 -- 4 filters 2.28 / 2.42 / 1.99e10,
    // 8: 3.09 / 3.60 / 3.26e10, 16: 3.66 / 4.93 / 4.75e10, 24: 3.60 / 5.15 / 5.36e10; m = 1 -- 3 filters 1.96 / 2.09 / 1.91e10,
    // 8: 3.56 / 4.89 / 4.51e10, 16: 4.41 / 6.89 / 6.89e10, 20: 4.49 / 7.51 / 7.76e10.
    const size_t b1 = (size_t)(b.N + GRID_NT - 1) / GRID_NT, b2 = (b1 + 1) / 2, b4 = (b1 + 3) / 4;
    const bool small = b.MP <= 2;
    if (b1 > 128 && b2 <= 128) {           // two particles per lane bring the filter into the round-2 kernel
      if (w1 >= (size_t)(small ? 5120 : 6144) && b.N >= 4 * GRID_NT) ppt = 4;
      // (below that only with complete blocks -- the variant without tail guards: a single 65 536-particle call 9.30 -> 8.96 us per
      // step, m = 1 8.73 -> 8.21, two in flight 9.83 -> 9.33 / 9.09 -> 8.40; 40 000 particles, ragged: 9.0 -> 10.1, stays at one;
      // scripts/ppt_single.py, profiles/r02d_ppt_single.txt)
      // (round 5, the step kernel carries 256 blocks: a single m = 4 call of 65 536 particles 9.16 -> 8.57 us per step at ONE per lane, two in
      // flight 9.52 -> 9.17; m = 1 even / 8.54 vs 8.85: stays -- profiles/r05_step_kernel_unified.txt)
      else if (w1 >= (size_t)(small ? 768 : 1024) || (small && b.N % (2 * GRID_NT) == 0)) ppt = 2;
    } else if (b2 > 128 && b4 <= 128) {    // only four do
      // (with complete blocks from one filter on: 131 072 particles, single call 12.8 -> 12.1 us per step, m = 1 11.4 -> 10.2)
      if (w1 >= (size_t)(small ? 2048 : 4096) || b.N % (4 * GRID_NT) == 0) ppt = 4;
      else if (w1 >= 3072) ppt = 2;
    } else {
      // (32 768 particles, full output, 1 / 2 / 4 per lane: m = 4 -- 8 filters 2.63 / 2.54 / 2.06e10, 16: 3.70 / 3.81 / 3.41e10,
      // 24: 4.15 / 4.61 / 3.98e10, 48: 4.38 / 5.30 / 5.41e10; m = 1 -- 16: 4.70 / 5.08 / 4.69e10, 32: 5.87 / 7.40 / 7.24e10,
      // 48: 5.99 / 8.23 / 8.44e10; profiles/r02d_ppt_single.txt)
      if (w1 >= 6144 && b.N >= 4 * GRID_NT) ppt = 4;
      else if (w1 >= 2048 && b.N >= 2 * GRID_NT) ppt = 2;
    }
  }
  // (every other class -- functor models, run-time compiled user models, sixteen-dimensional states: one particle per lane up to the
  // step kernel's 256 blocks = 65 536 particles, two up to 131 072: a single filter of 1e5 particles as 196 workgroups of two per lane,
  // not 98 of four -- scripts/step_kernel_unified_probe.py)
  // 8-dimensional states (round-2 multi-series kernel): two particles per lane once the batch fills the chip twice
  // (C4, 16 384 particles: 32 filters 1.92e10 -> 1.99e10, 64 filters 1.90e10 -> 2.19e10 particle-steps/s; 16 filters 1.58e10 -> 1.37e10)
  if (b.MP == 8 && !b.fpar && w1 >= 2048 && b.N >= 2 * GRID_NT) ppt = 2;
  if (opt().grid_ppt > 0) ppt = opt().grid_ppt >= 4 ? 4 : (opt().grid_ppt >= 2 ? 2 : 1);  // tuning hook (BSSM_AMD_GRID_PPT)
  while ((b.N + GRID_NT * ppt - 1) / (GRID_NT * ppt) > 256 && ppt < 4) ppt *= 2;  // (at most 256 blocks per filter: pf_grid_v2.h; callers refuse N > 262 144)
  const int chunk = GRID_NT * ppt;
  const int B = (b.N + chunk - 1) / chunk;
  if (B_out) *B_out = B;
  if (chunk_out) *chunk_out = chunk;
  size_t bytes = 0;
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  bytes += 2 * al((size_t)b.F * b.m * b.N * sizeof(double));  // state ping-pong
  bytes += al((size_t)2 * b.F * b.N * sizeof(double));        // lcdf
  bytes += al((size_t)2 * b.F * B * (chunk / 16) * sizeof(double));  // coarse
  bytes += al((size_t)2 * b.F * B * 2 * sizeof(double));      // partials
  bytes += al((size_t)2 * b.F * (b.n + 1) * sizeof(double));  // maxhist + llinc (contiguous)
  bytes += al((size_t)b.F * (b.n + 1) * B * sizeof(double));  // wscale (factored weights output)
#ifdef BSSM_XCD_PERSIST
  bytes += al((size_t)kPersistWords * sizeof(unsigned) + sizeof(FilterBatch) + sizeof(GridWorkspace));  // control block + descriptors of the persistent kernel
#endif
  return bytes;
}

void grid_workspace_carve(const FilterBatch& b, void* base, GridWorkspace* ws) {
  int B, chunk;
  grid_workspace_bytes(b, &B, &chunk);
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  char* p = (char*)base;
  ws->state[0] = (double*)p; p += al((size_t)b.F * b.m * b.N * sizeof(double));
  ws->state[1] = (double*)p; p += al((size_t)b.F * b.m * b.N * sizeof(double));
  ws->lcdf = (double*)p; p += al((size_t)2 * b.F * b.N * sizeof(double));
  ws->coarse = (double*)p; p += al((size_t)2 * b.F * B * (chunk / 16) * sizeof(double));
  ws->partials = (double*)p; p += al((size_t)2 * b.F * B * 2 * sizeof(double));
  ws->maxhist = (double*)p; ws->llinc = ws->maxhist + (size_t)b.F * (b.n + 1);
  p += al((size_t)2 * b.F * (b.n + 1) * sizeof(double));
  ws->wscale = (double*)p; p += al((size_t)b.F * (b.n + 1) * B * sizeof(double));
#ifdef BSSM_XCD_PERSIST
  ws->persist = (unsigned*)p; p += al((size_t)kPersistWords * sizeof(unsigned) + sizeof(FilterBatch) + sizeof(GridWorkspace));
#else
  ws->persist = nullptr;  // (the control block of the round-4 persistent-kernel experiment: experiments/xcd_persist.h, `make persist` builds only)
#endif
  ws->fw = 0;
  ws->logw = nullptr;
  ws->dbg = nullptr;
  ws->B = B;
  ws->chunk = chunk;
  ws->f0 = 0;
  ws->xcd_remap = 1;
  ws->wx = 0;
  ws->side = library_side_stream();
}

bool row_filter_supported(const FilterBatch& b) {
  if (b.N < 1) return false;
  if (b.MP >= 32 && !b.fpar && b.N <= CANON_ROW * BIG_WIDE_ROWS) return true;   // (pf_big.h: up to 102 400 particles with the row prefixes in LDS, round 6)
  if (b.N > CANON_ROW * ROW_MAX_ROWS) return false;
  if (b.fpar) return b.MP == b.KP && (b.user_slot > 0 ? b.MP <= 16 : b.MP <= 2);
  if (opt().force_path == 3 && b.MP == 8 && b.KP % 4 == 0) return true;  // EXPERIMENT hook only (pf_filter_impl.h launch_grid_mk): not bit-identical to pf_wg_kernel
  return b.MP <= 4 || b.MP >= 32;  // (eight- and sixteen-dimensional states: pf_wg_kernel's matrix-core transitions sum in another order)
}
bool wg_lean_form_exists(const FilterBatch& b) {
  if (b.MP > 2 || b.N <= 2048 || b.N > 4096 || user_slot_rows_only(b.user_slot)) return false;
  if (opt().wg_wide == 0) return false;           // (tuning switch: the 512-thread kernels have no lean form)
  return b.fpar ? b.MP == b.KP : true;
}
bool prefer_step_shape(const FilterBatch& b, bool use_grid) {
  if (use_grid || b.MP < 32 || b.fpar) return use_grid;
  const long long rows = (long long)b.F * ((b.N + CANON_ROW - 1) / CANON_ROW);
  // (profiles/r05_big_states.txt: m = 53, 1024 x 256 particles 12.8 ms workgroup-resident, 10.6 ms as rows; m = 25: 256 particles per filter 5.9 ms
  // against 5.3 ms, but 512 particles per filter 2.15 ms against 2.6 ms)
  if (b.MP >= 64) return rows >= 1024;
  return b.N <= 256 && rows >= 2048;
}
size_t row_workspace_bytes(const FilterBatch& b) { return row_workspace_bytes_impl(b); }
void row_workspace_carve(const FilterBatch& b, void* base, GridWorkspace* ws) {
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  const int B = (b.N + CANON_ROW - 1) / CANON_ROW;
  char* p = (char*)base;
  ws->state[0] = (double*)p; p += al((size_t)b.F * b.m * b.N * sizeof(double));
  ws->state[1] = (double*)p; p += al((size_t)b.F * b.m * b.N * sizeof(double));
  ws->lcdf = (double*)p; p += al((size_t)2 * b.F * b.N * sizeof(double));
  ws->partials = (double*)p; p += al((size_t)2 * b.F * B * 2 * sizeof(double));
  ws->maxhist = (double*)p; ws->llinc = ws->maxhist + (size_t)b.F * (b.n + 1);
  ws->coarse = nullptr; ws->wscale = nullptr; ws->persist = nullptr; ws->logw = nullptr; ws->dbg = nullptr;
  ws->fw = 0; ws->wx = 0; ws->B = B; ws->chunk = CANON_ROW; ws->f0 = 0; ws->xcd_remap = 0;
  ws->side = library_side_stream();
}

__global__ void zero_dead_tail_kernel(const FilterBatch b, const bool weights_too, double* __restrict__ wscale, const int B) {
  const int f = blockIdx.y;
  const int da = b.dead_at[f];
  if (da == 0) return;
  const int td = da - 1, n = b.n, m = b.m, N = b.N;
  const size_t stride = (size_t)gridDim.x * blockDim.x, i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b.alpha_hist) {  // slots td+1 .. n were never produced
    double* __restrict__ a = b.alpha_hist + ((size_t)f * (n + 1) + (td + 1)) * m * N;
    for (size_t i = i0; i < (size_t)(n - td) * m * N; i += stride) a[i] = 0.0;
  }
  if (b.indices) {     // columns td .. n-1
    uint32_t* __restrict__ x = b.indices + ((size_t)f * n + td) * N;
    for (size_t i = i0; i < (size_t)(n - td) * N; i += stride) x[i] = 0u;
  }
  if (weights_too && b.weights) {  // columns td+1 .. n
    double* __restrict__ w = b.weights + ((size_t)f * (n + 1) + (td + 1)) * N;
    for (size_t i = i0; i < (size_t)(n - td) * N; i += stride) w[i] = 0.0;
    if (wscale) {  // factored weights output: the launches behind the dead step wrote factors from poisoned partials
      double* __restrict__ sc = wscale + ((size_t)f * (n + 1) + (td + 1)) * B;
      for (size_t i = i0; i < (size_t)(n - td) * B; i += stride) sc[i] = 0.0;
    }
  }
}
hipError_t launch_zero_dead_tail(const FilterBatch& b, bool weights_too, hipStream_t stream, double* wscale, int B) {
  if (!b.alpha_hist && !b.indices && !b.weights) return hipSuccess;
  hipLaunchKernelGGL(zero_dead_tail_kernel, dim3(256, b.F), dim3(256), 0, stream, b, weights_too, wscale, B);
  return hipGetLastError();
}

// gaussian_psi_smoother (src/model_ssm_ulg.cpp:644-672): every particle is an independent draw from the joint
// smoothing distribution, alpha_{t+1} = alphahat_{t+1} + C_{t+1}(alpha_t - alphahat_t) + V_{t+1} u: no weights, no
// resampling, so one lane walks one trajectory.  `mats` rows hold (C_{t+1}, V_{t+1}, alphahat_{t+1} - C_{t+1} alphahat_t).
__global__ void __launch_bounds__(256) sim_smoother_kernel(const double* __restrict__ mats, const double* __restrict__ init,
                                                           double* __restrict__ hist, int n, int m, int N, uint32_t seed) {
  math_tab_init();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const RngKey key{seed, 0u};
  const int MS = 2 * m * m + m;
  double x[16], z[16], xn[16];
  rng_normals_rt(key, RNG_INIT, 0u, (uint32_t)i, m, z);
  for (int j = 0; j < m; j++) {
    double s = init[j];
    for (int c = 0; c < m; c++) s += init[m + j + m * c] * z[c];
    x[j] = s;
    hist[(size_t)j * N + i] = s;
  }
  for (int t = 0; t < n; t++) {
    const double* __restrict__ mt = mats + (size_t)t * MS;
    rng_normals_rt(key, RNG_TRANSITION, (uint32_t)t, (uint32_t)i, m, z);
    for (int j = 0; j < m; j++) {
      double s = 0.0, q = 0.0;
      for (int c = 0; c < m; c++) { s += mt[j + m * c] * x[c]; q += mt[m * m + j + m * c] * z[c]; }
      xn[j] = mt[2 * m * m + j] + s + q;
    }
    for (int j = 0; j < m; j++) { x[j] = xn[j]; hist[((size_t)(t + 1) * m + j) * N + i] = xn[j]; }
  }
}
hipError_t launch_sim_smoother(const double* mats, const double* init, double* hist, int n, int m, int N, uint32_t seed,
                               hipStream_t stream) {
  if (m > 14) return hipErrorInvalidValue;
  hipLaunchKernelGGL(sim_smoother_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, mats, init, hist, n, m, N, seed);
  return hipGetLastError();
}

// test hook: evaluate the kernels' elementary functions on an array (see tests/test_gpu_math.py)
__global__ void debug_math_kernel(int fn, const double* __restrict__ in, double* __restrict__ out, int n) {
  math_tab_init();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = in[i];
  if (fn == 0) out[i] = tl_exp(x);
  else if (fn == 1) out[i] = tl_log(x);
  else if (fn == 2) {
    double sn, cs;
    tl_sincos_2pi((uint32_t)(x * 0x1.0p32), sn, cs);  // x in [0,1), a multiple of 2^-32
    out[2 * i] = sn; out[2 * i + 1] = cs;
  } else if (fn == 10) out[i] = fast_exp(x);  // the polynomial forms of round 1, kept for comparison
  else if (fn == 11) out[i] = fast_log(x);
  else if (fn == 12) {
    double sn, cs;
    fast_sincos_2pi((uint32_t)(x * 0x1.0p32), sn, cs);
    out[2 * i] = sn; out[2 * i + 1] = cs;
  } else if (fn == 4) {  // four PACKED normals of (seed = 1, stream = 0), counter (i, 0, 0, purpose 2)
    double z[4];
    rng_normals<4>(RngKey{1u, 0u}, RNG_TRANSITION, 0u, (uint32_t)i, z);
    for (int q = 0; q < 4; q++) out[4 * i + q] = z[q];
  } else if (fn == 5) {  // resampling uniform of slot i at step 3 of a filter with a PACKED step draw
    out[i] = (double)rng_resample_word<4>(RngKey{1u, 0u}, 3u, (uint32_t)i, u32x4{0u, 0u, 0u, 0u}) * 0x1.0p-32;
  } else if (fn == 6) {  // maximum of the wave's 64 values through the integer-key reduction (pf_device.h: wave_max_key)
    out[i] = wave_max_key(x);
  } else if (fn == 3) {  // Box-Muller pair for Philox key (seed = 1, stream = 0), counter (i, 0, 0, purpose 2)
    double z0, z1;
    rng_normal_pair(RngKey{1u, 0u}, RNG_TRANSITION, 0u, (uint32_t)i, 0u, z0, z1);
    out[2 * i] = z0; out[2 * i + 1] = z1;
  }
}
hipError_t launch_debug_math(int fn, const double* in, double* out, int n, hipStream_t stream) {
  hipLaunchKernelGGL(debug_math_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, fn, in, out, n);
  return hipGetLastError();
}

// ---------------------------------------------------------------- psi set-up smoother, batched over theta-draws
// One lane per draw runs the whole Kalman filter / covariance smoother / conditional-factor recursion of its draw
// (sequential in time, O(n m^3)); draws are independent, so a post-correction chunk is one launch.  The same
// template runs on the host (host_model.cpp), which is what the device result is tested against.
template <int M>
__global__ void __launch_bounds__(64) lg_setup_kernel(const LgBatchDev b) {
  const int d = blockIdx.x * 64 + threadIdx.x;
  if (d >= b.F) return;
  // draw-interleaved arrays: element k of draw d lives at base[k * F + d]
  const size_t F = (size_t)b.F;
  const double* in = b.in + d;
  LgModel g;
  g.n = b.n; g.sub = b.sub > 1 ? b.sub : 1;
  g.Z = in + b.oZ * F; g.Zs = b.Zs; g.T = in + b.oT * F; g.Ts = b.Ts; g.RR = in + b.oR * F; g.Rs = b.Rs;
  g.D = in + b.oD * F; g.Ds = b.Ds; g.C = in + b.oC * F; g.Cs = b.Cs; g.y = in + b.oy * F; g.H = in + b.oH * F; g.Hs = b.Hs;
  g.a1 = in + b.oa * F; g.P1 = in + b.oP * F;
  const bool ok = LgSmoother<M>::run(g, b.at + d, b.Pt + d, b.cc + d, b.K + d, b.v + d, b.Fs + d, F);
  b.ok[d] = ok ? 1 : 0;
}
hipError_t launch_lg_setup(const LgBatchDev& b, hipStream_t stream) {
  const dim3 grid((b.F + 63) / 64), block(64);
  switch (b.m) {  // one instantiation per state dimension keeps the small cases in registers
    case 1: hipLaunchKernelGGL(lg_setup_kernel<1>, grid, block, 0, stream, b); break;
    case 2: hipLaunchKernelGGL(lg_setup_kernel<2>, grid, block, 0, stream, b); break;
    case 3: hipLaunchKernelGGL(lg_setup_kernel<3>, grid, block, 0, stream, b); break;
    case 4: hipLaunchKernelGGL(lg_setup_kernel<4>, grid, block, 0, stream, b); break;
    case 5: hipLaunchKernelGGL(lg_setup_kernel<5>, grid, block, 0, stream, b); break;
    case 6: hipLaunchKernelGGL(lg_setup_kernel<6>, grid, block, 0, stream, b); break;
    case 7: hipLaunchKernelGGL(lg_setup_kernel<7>, grid, block, 0, stream, b); break;
    case 8: hipLaunchKernelGGL(lg_setup_kernel<8>, grid, block, 0, stream, b); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
// ---------------------------------------------------------------- psi set-up of ssm_nlg draws, all of it on the device
// What psi_setup / fn_psi_setup_begin + build_fn_psi_tables do per draw on the host (host_model.cpp), as one lane per
// draw: the host then only packs theta (FP_SIZE doubles) and a1 / P1 per draw.  Work arrays are draw-interleaved
// (element k of draw d at base[k * F + d]) like lg_setup_kernel's; the tables go out per draw, contiguous, in the
// layout DeviceBatch hands to the filter kernels.
size_t fn_setup_scratch_doubles(int n, int m) {
  const size_t mm = (size_t)m * m, n1 = (size_t)n + 1;
  //      Z            T        RR   D   C            y   H   a1  P1   at       Pt        cc        K            v   Fs   Vf        Cf
  return (size_t)n * m + n * mm + mm + n + (size_t)n * m + n + 1 + m + mm + n1 * m + n1 * mm + n1 * mm + (size_t)n * m + n + n + n1 * mm + n1 * mm;
}
// Four kernels per chunk, every one with lane = draw and 64 draws per single-wave workgroup:
//   0  linearisation of time point blockIdx.y (and the prior / noise blocks by blockIdx.y == 0)     grid (F / 64, n)
//   1  Kalman filter + covariance smoother, sequential in time -- the only sequential part           grid (F / 64)
//   2  conditional factors of time point blockIdx.y (independent given the smoother's output)        grid (F / 64, n + 1)
//   3  tables of time point blockIdx.y (and the initial state by blockIdx.y == 0)                    grid (F / 64, n)
// As one kernel whose waves shared the draws, a chunk took 1.5 ms on the C5 shape with five busy CUs (292 draws): the tables alone
// 0.5 ms, because each of their store instructions touches 64 cache lines (per-draw contiguous output) and only those CUs issued
// them; spread over the chip by time point the three parallel phases cost a few launches (profiles/r03_setup_phases.txt).
// Work arrays in BLOCKS of 64 draws (round 5, late): element k of draw d at block_base(d / 64)[k * 64 + d % 64] -- the stride between the
// elements of a draw is the compile-time constant 64 instead of the number of draws of the chunk, so every `k * stride` folds into an
// immediate or a shift and the per-lane address is a block base plus the lane's 8 bytes: 1 261 -> 853 instructions in the sequential phase
// (scalar address arithmetic 578 -> 271).  The phase itself only went 665 -> 641 us per 300 time points: one wave per SIMD issues in order,
// and ~450 fp64 VALU instructions at four cycles each per time point are what it costs (not memory latency: a four-step operand prefetch
// changed nothing) -- docs/design_round_5.md.
constexpr size_t FN_ES = 64;
template <int M>
struct FnWork {
  double *sZ, *sT, *sRR, *sD, *sC, *sy, *sH, *sa1, *sP1, *at, *Pt, *cc, *K, *v, *Fs, *Vf, *Cf;
  __device__ FnWork(const FnSetupDev& b, int d) {
    constexpr int MM = M * M;
    constexpr size_t F = FN_ES;
    const int n = b.n;
    const size_t mm = MM, n1 = (size_t)n + 1;
    const size_t total = (size_t)n * M + n * mm + mm + n + (size_t)n * M + n + 1 + M + mm + n1 * M + n1 * mm + n1 * mm + (size_t)n * M + n + n + n1 * mm + n1 * mm;  // fn_setup_scratch_doubles
    double* w = b.scratch + (size_t)(d >> 6) * total * FN_ES + (size_t)(d & 63);
    sZ = w;                 w += (size_t)n * M * F;
    sT = w;                 w += (size_t)n * MM * F;
    sRR = w;                w += (size_t)MM * F;
    sD = w;                 w += (size_t)n * F;
    sC = w;                 w += (size_t)n * M * F;
    sy = w;                 w += (size_t)n * F;
    sH = w;                 w += F;
    sa1 = w;                w += (size_t)M * F;
    sP1 = w;                w += (size_t)MM * F;
    at = w;                 w += (size_t)(n + 1) * M * F;
    Pt = w;                 w += (size_t)(n + 1) * MM * F;
    cc = w;                 w += (size_t)(n + 1) * MM * F;
    K = w;                  w += (size_t)n * M * F;
    v = w;                  w += (size_t)n * F;
    Fs = w;                 w += (size_t)n * F;
    Vf = w;                 w += (size_t)(n + 1) * MM * F;  // V_t^(1/2) and C_t
    Cf = w;
  }
};
template <int M, int PHASE>
__global__ void __launch_bounds__(64) fn_setup_kernel(const FnSetupDev b) {
  constexpr int MM = M * M;
  constexpr size_t F = FN_ES;  // (stride between the elements of a draw in the work arrays: FnWork)
  const int n = b.n;
  if constexpr (PHASE == 0) {
    // ---- linearisation at the stored mode: Z_t = Z'(mode_t), D_t = Z(mode_t) - Z_t mode_t, T_t = T'(mode_t),
    // C_t = T(mode_t) - T_t mode_t  (approximate_for_is, src/model_ssm_nlg.cpp:146-176; scales and approx_loglik stay 0)
    math_tab_init();  // (a workgroup barrier inside: surplus lanes redo the last draw instead of leaving)
    const int d = min((int)(blockIdx.x * 64 + threadIdx.x), b.F - 1), t = blockIdx.y;
    const FnWork<M> W(b, d);
    const double* __restrict__ fp = b.fpar + (size_t)d * fp_size(M <= 1 ? 1 : (M <= 2 ? 2 : (M <= 4 ? 4 : 8)));
    const int id = (int)fp[FP_ID];
    const double* __restrict__ mode = b.modes + (size_t)d * M * n;
    double a[M], Zg[M], Tg[MM], Tf[M];
#pragma unroll
    for (int j = 0; j < M; j++) a[j] = mode[(size_t)M * t + j];
    nlg_Z_gn<M>(id, fp, a, Zg);
    nlg_T_gn<M>(id, fp, a, Tg);
    double za = 0.0;
#pragma unroll
    for (int j = 0; j < M; j++) za += Zg[j] * a[j];
    W.sD[(size_t)t * F] = nlg_Z_fn<M>(id, fp, a) - za;
    nlg_T_fn<M>(id, fp, a, Tf);
#pragma unroll
    for (int i = 0; i < M; i++) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < M; j++) s += Tg[i + M * j] * a[j];
      W.sC[((size_t)t * M + i) * F] = Tf[i] - s;
      W.sZ[((size_t)t * M + i) * F] = Zg[i];
    }
#pragma unroll
    for (int q = 0; q < MM; q++) W.sT[((size_t)t * MM + q) * F] = Tg[q];
    W.sy[(size_t)t * F] = b.y[t];
    if (t == 0) {
#pragma unroll
      for (int q = 0; q < MM; q++) W.sRR[(size_t)q * F] = 0.0;
#pragma unroll
      for (int j = 0; j < M; j++) W.sRR[(size_t)(j + M * j) * F] = fp[FP_R + j] * fp[FP_R + j];  // R is diagonal in every shipped model
      W.sH[0] = fp[FP_H];
      const double* __restrict__ ex = b.ext + (size_t)d * (M + MM);
#pragma unroll
      for (int j = 0; j < M; j++) W.sa1[(size_t)j * F] = ex[j];
#pragma unroll
      for (int q = 0; q < MM; q++) W.sP1[(size_t)q * F] = ex[M + q];
    }
  } else {
    const int d = blockIdx.x * 64 + threadIdx.x;
    if (d >= b.F) return;
    const FnWork<M> W(b, d);
    if constexpr (PHASE == 1) {
      LG_STAMP(1);
      LgModel g;
      g.n = n;
      g.Z = W.sZ; g.Zs = M; g.T = W.sT; g.Ts = MM; g.RR = W.sRR; g.Rs = 0; g.D = W.sD; g.Ds = 1; g.C = W.sC; g.Cs = M;
      g.y = W.sy; g.H = W.sH; g.Hs = 0; g.a1 = W.sa1; g.P1 = W.sP1;
      b.ok[d] = LgSmoother<M>::run(g, W.at, W.Pt, W.cc, W.K, W.v, W.Fs, F, false) ? 1 : 0;  // at = alphahat, Pt = Var, cc = lag-one Cov
#ifdef BSSM_SETUP_STAMPS
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned long long* q = g_setup_stamps;
        printf("fn_setup_kernel<%d, 1> F=%d n=%d us: forward %.1f backward %.1f\n", M, b.F, n, (q[2] - q[1]) * 0.01, (q[3] - q[2]) * 0.01);
      }
#endif
    } else if constexpr (PHASE == 2) {
      if (!b.ok[d]) return;
      if (!LgSmoother<M>::cond_step((int)blockIdx.y, W.Pt, W.cc, W.Vf, W.Cf, F)) b.ok[d] = 0;  // (every writer stores 0)
    } else {
      // ---- tables (build_fn_psi_tables): proposal operands, observation rows, initial state, approximating (T_t, C_t)
      const int t = blockIdx.y;
      const double *at = W.at, *Pt = W.Vf, *cc = W.Cf;
      const double* __restrict__ fp = b.fpar + (size_t)d * fp_size(M <= 1 ? 1 : (M <= 2 ? 2 : (M <= 4 ? 4 : 8)));
      const bool ok = b.ok[d] != 0;
      constexpr int MS = 2 * MM + M, OS = ROW_HDR + SER_HDR + M, AS = MM + M;
      double* __restrict__ mats = b.mats + (size_t)d * n * MS;
      double* __restrict__ obs = b.obs + (size_t)d * n * OS;
      double* __restrict__ ini = b.init + (size_t)d * (M + MM);
      double* __restrict__ am = b.amats + (size_t)d * n * AS;
      const double ih = 1.0 / fp[FP_H];
      double* mt = mats + (size_t)t * MS;
      double ah0[M], ah1[M], Ct1[MM];
#pragma unroll
      for (int j = 0; j < M; j++) { ah0[j] = at[((size_t)t * M + j) * F]; ah1[j] = at[((size_t)(t + 1) * M + j) * F]; }
#pragma unroll
      for (int q = 0; q < MM; q++) { Ct1[q] = ok ? cc[((size_t)(t + 1) * MM + q) * F] : 0.0; mt[q] = Ct1[q]; mt[MM + q] = ok ? Pt[((size_t)(t + 1) * MM + q) * F] : 0.0; }
#pragma unroll
      for (int j = 0; j < M; j++) {
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < M; c++) s += Ct1[j + M * c] * ah0[c];
        mt[2 * MM + j] = ok ? ah1[j] - s : 0.0;
      }
      double* row = obs + (size_t)t * OS;
      const double yt = b.y[t];
      const bool seen = ok && yt == yt && fabs(yt) <= 1.7976931348623157e308;
#pragma unroll
      for (int q = 0; q < OS; q++) row[q] = 0.0;
      if (seen) {
        row[RH_FLAG] = 1.0;  // (RH_SCALE: the scales of approximate_for_is are zero)
        double* ob = row + ROW_HDR;
        ob[OB_FLAG] = 1.0; ob[OB_Y] = yt; ob[OB_D] = W.sD[(size_t)t * F]; ob[OB_PSIIH] = ih;
#pragma unroll
        for (int c = 0; c < M; c++) ob[SER_HDR + c] = W.sZ[((size_t)t * M + c) * F];
      }
      double* a2 = am + (size_t)t * AS;
#pragma unroll
      for (int q = 0; q < MM; q++) a2[q] = ok ? W.sT[((size_t)t * MM + q) * F] : 0.0;
#pragma unroll
      for (int j = 0; j < M; j++) a2[MM + j] = ok ? W.sC[((size_t)t * M + j) * F] : 0.0;
      if (t == 0) {
#pragma unroll
        for (int j = 0; j < M; j++) ini[j] = ok ? at[(size_t)j * F] : 0.0;
#pragma unroll
        for (int q = 0; q < MM; q++) ini[M + q] = ok ? Pt[(size_t)q * F] : 0.0;
      }
    }
  }
}
template <int M>
static hipError_t launch_fn_setup_m(const FnSetupDev& b, hipStream_t stream) {
  const unsigned gx = (unsigned)((b.F + 63) / 64);
  hipLaunchKernelGGL((fn_setup_kernel<M, 0>), dim3(gx, (unsigned)b.n), dim3(64), 0, stream, b);
  // (the row-distributed form was measured here too: two lanes per draw, 0.89 ms against 0.78 ms -- at m = 2 everything fits one lane's registers)
  hipLaunchKernelGGL((fn_setup_kernel<M, 1>), dim3(gx), dim3(64), 0, stream, b);
  hipLaunchKernelGGL((fn_setup_kernel<M, 2>), dim3(gx, (unsigned)b.n + 1), dim3(64), 0, stream, b);
  hipLaunchKernelGGL((fn_setup_kernel<M, 3>), dim3(gx, (unsigned)b.n), dim3(64), 0, stream, b);
  return hipGetLastError();
}
hipError_t launch_fn_setup(const FnSetupDev& b, hipStream_t stream) {
  if (b.n < 1 || b.n >= 65535) return hipErrorInvalidValue;  // (time points ride in gridDim.y)
  switch (b.m) {
    case 1: return launch_fn_setup_m<1>(b, stream);
    case 2: return launch_fn_setup_m<2>(b, stream);
    default: return hipErrorInvalidValue;
  }
}
// ---------------------------------------------------------------- psi set-up of ssm_ung draws, all of it on the device
// What psi_setup_begin (laplace_iter + update_scales) + the smoother + build_psi_tables / fill_obs_rows do per draw on
// the host (host_model.cpp), as one lane per draw.  Elementary functions are the device library's (not the filter
// kernels' table-driven forms): the result is compared with the host set-up at 1e-12.
size_t ung_setup_scratch_doubles(int n, int m) {
  const size_t mm = (size_t)m * m, n1 = (size_t)n + 1;
  //      ay  aH  ayraw scales  at       Pt        cc        K              v   Fs   Vf        Cf
  return (size_t)4 * n + n1 * m + n1 * mm + n1 * mm + (size_t)n * m + n + n + n1 * mm + n1 * mm;
}
// Four kernels per chunk, as for ssm_nlg draws (fn_setup_kernel): PHASE 0 the Laplace step of time point blockIdx.y, 1 the Kalman filter +
// covariance smoother (sequential in time; ROWS: eight lanes per draw for 8-dimensional states, lg_smoother_rows.h), 2 the conditional
// factors of time point blockIdx.y, 3 its tables.
template <int M>
struct UngWork {
  const double* in;
  double *ay, *aH, *ayr, *scl, *at, *Pt, *cc, *K, *v, *Fs, *Vf, *Cf;
  size_t em;   // element stride of the matrix work arrays: F (draw-interleaved) or 1 (draw-major, UngSetupDev::dm)
  __device__ UngWork(const UngSetupDev& b, int d) {
    constexpr int MM = M * M;
    const size_t F = (size_t)b.F;
    const int n = b.n;
    const bool dm = b.dm != 0;
    em = dm ? 1 : F;
    in = b.in + d;
    double* w = b.scratch;   // region bases; the vectors are always draw-interleaved (+ d), a matrix array's draw slice sits at d x its per-draw size when draw-major
    auto mat = [&](size_t per_draw) { double* p = dm ? w + (size_t)d * per_draw : w + d; w += per_draw * F; return p; };
    ay = w + d;     w += (size_t)n * F;   // approximating y minus x'beta: what the smoother sees
    aH = w + d;     w += (size_t)n * F;
    ayr = w + d;    w += (size_t)n * F;   // approximating y
    scl = w + d;    w += (size_t)n * F;
    at = mat((size_t)(n + 1) * M);
    Pt = mat((size_t)(n + 1) * MM);
    cc = mat((size_t)(n + 1) * MM);
    K = mat((size_t)n * M);
    v = w + d;      w += (size_t)n * F;
    Fs = w + d;     w += (size_t)n * F;
    Vf = mat((size_t)(n + 1) * MM);
    Cf = mat((size_t)(n + 1) * MM);
  }
};
template <int M, int PHASE, bool ROWS>
__global__ void __launch_bounds__(64) ung_setup_kernel(const UngSetupDev b) {
  constexpr int MM = M * M, MP = M <= 1 ? 1 : (M <= 2 ? 2 : (M <= 4 ? 4 : (M <= 8 ? 8 : (M <= 16 ? 16 : (M <= 32 ? 32 : 64)))));
  constexpr int DPW = (PHASE == 1 && ROWS) ? 64 / M : 64;  // draws per workgroup
  constexpr int kU = M >= 32 ? 1 : M;                      // (the copy loops of the tables stay loops for the padded big states)
  const int d0 = blockIdx.x * DPW + ((PHASE == 1 && ROWS) ? (int)threadIdx.x / M : (int)threadIdx.x);
  if (!(PHASE == 1 && ROWS) && d0 >= b.F) return;
  const int d = min(d0, b.F - 1);  // (rows: surplus groups redo the last draw -- every lane takes part in the LDS exchanges)
  const size_t F = (size_t)b.F;
  const int n = b.n;
  const UngWork<M> W(b, d);
  const double* in = W.in;
  const int dist = (int)in[b.oh * F];
  const double phi = in[(b.oh + 1) * F];
  if constexpr (PHASE == 0) {
    // ---- approximate_for_is: one Laplace step at the stored mode (src/model_ssm_ung.cpp:302-342) and the scales (:241-292)
    const int t = blockIdx.y;
    const double y = in[(b.oy + t) * F], sg = in[(b.om + t) * F], u = in[(b.ou + t) * F];
    double HH = 0.0, yy = 0.0, sv = 0.0;
    switch (dist) {
      case DIST_SVM: {
        double tmp = y;
        if (fabs(tmp) < 1e-4) tmp = 1e-4;
        HH = 2.0 * exp(sg) / ((tmp / phi) * (tmp / phi));
        yy = sg + 1.0 - 0.5 * HH;
        sv = -0.5 * (sg + (y / phi) * (y / phi) * exp(-sg));
      } break;
      case DIST_POISSON: HH = 1.0 / (exp(sg) * u); yy = y * HH + sg - 1.0; sv = y * sg - u * exp(sg); break;
      case DIST_BINOMIAL: { const double e = exp(sg); HH = (1.0 + e) * (1.0 + e) / (u * e); yy = y * HH + sg - 1.0 - e; sv = y * sg - u * log1p(e); } break;
      case DIST_NEGBIN: { const double e = exp(sg) * u; HH = (phi + e) * (phi + e) / (phi * e * (y + phi)); yy = sg + (phi + e) * (y - e) / ((y + phi) * e);
                          sv = y * sg - (y + phi) * log(phi + u * exp(sg)); } break;
      case DIST_GAMMA: { const double e = exp(sg) * u; HH = e / (y * phi); yy = sg - e / y + 1.0; sv = -phi * (sg + (y * exp(-sg) / u)); } break;
      default: break;
    }
    const double sd = sqrt(HH);
    const bool seen = y == y && fabs(y) <= 1.7976931348623157e308;
    const double q = (yy - sg) / sd;
    W.ay[(size_t)t * F] = yy - in[(b.ox + t) * F];
    W.ayr[(size_t)t * F] = yy;
    W.aH[(size_t)t * F] = sd;
    W.scl[(size_t)t * F] = seen ? sv + 0.5 * q * q : 0.0;
  } else if constexpr (PHASE == 1) {
    LgModel g;
    g.n = n;
    g.Z = in + b.oZ * F; g.Zs = b.Zs; g.T = in + b.oT * F; g.Ts = b.Ts; g.RR = in + b.oR * F; g.Rs = b.Rs;
    g.D = in + b.oD * F; g.Ds = b.Ds; g.C = in + b.oC * F; g.Cs = b.Cs; g.y = W.ay; g.H = W.aH; g.Hs = 1;
    g.a1 = in + b.oa * F; g.P1 = in + b.oP * F;
    bool ok;
    if constexpr (ROWS && M >= 32) {  // (17 ... 64 states, padded: ung_smooth_wave_kernel)
      ok = false;
    } else if constexpr (ROWS) {
      __shared__ double lds[LgRows<M>::kLdsDoubles];
      LgRows<M> rows(lds);
      ok = rows.run(g, W.at, W.Pt, W.cc, W.K, W.v, W.Fs, F);
    } else {
      ok = LgSmoother<M>::run(g, W.at, W.Pt, W.cc, W.K, W.v, W.Fs, F, false);  // at = alphahat, Pt = Var, cc = lag-one Cov
    }
    b.ok[d] = ok ? 1 : 0;
  } else if constexpr (PHASE == 2) {
    if constexpr (M < 16) {  // (16 and more states: ung_factors_rows_kernel)
      if (!b.ok[d]) return;
      if (!LgSmoother<M>::cond_step((int)blockIdx.y, W.Pt, W.cc, W.Vf, W.Cf, F)) b.ok[d] = 0;  // (every writer stores 0)
    }
  } else {
    // ---- tables: psi proposal operands (src/model_ssm_ung.cpp:585-586), observation rows, initial state (:543)
    const int t = blockIdx.y;
    const double *at = W.at, *Pt = W.Vf, *cc = W.Cf;
    const bool ok = b.ok[d] != 0;
    constexpr int MS = 2 * MP * MP + MP, OS = ROW_HDR + SER_HDR + MP;
    double* __restrict__ mats = b.mats + (size_t)d * n * MS;
    double* __restrict__ obs = b.obs + (size_t)d * n * OS;
    double* __restrict__ ini = b.init + (size_t)d * (MP + MP * MP);
    double* mt = mats + (size_t)t * MS;
    if (!(M >= 16 && b.dm)) { for (int q = 0; q < MS; q++) mt[q] = 0.0; }
    if constexpr (M >= 32) {  // (no work arrays of a lane: a 32 x 32 matrix is 8 KB of scratch memory; MP == M here)
      // (draw-major work arrays, UngSetupDev::dm: setup_tables_big_kernel copies the matrices, a workgroup per (draw, time point), coalesced)
      if (ok && !b.dm) {
        const double* c1 = cc + (size_t)(t + 1) * MM * F;
        const double* p1 = Pt + (size_t)(t + 1) * MM * F;
        for (int q = 0; q < MM; q++) { mt[q] = c1[(size_t)q * F]; mt[MM + q] = p1[(size_t)q * F]; }
        for (int j = 0; j < M; j++) {
          double s = 0.0;
          for (int c = 0; c < M; c++) s += c1[(size_t)(j + M * c) * F] * at[((size_t)t * M + c) * F];
          mt[2 * MM + j] = at[((size_t)(t + 1) * M + j) * F] - s;
        }
      }
    } else if (ok && !(M == 16 && b.dm)) {   // (16 padded states with draw-major work arrays, late round 6: setup_tables_big_kernel as well)
      double ah0[M], ah1[M], Ct1[MM];
#pragma unroll
      for (int j = 0; j < M; j++) { ah0[j] = at[((size_t)t * M + j) * F]; ah1[j] = at[((size_t)(t + 1) * M + j) * F]; }
#pragma unroll
      for (int q = 0; q < MM; q++) Ct1[q] = cc[((size_t)(t + 1) * MM + q) * F];
#pragma unroll
      for (int c = 0; c < M; c++)
#pragma unroll
        for (int j = 0; j < M; j++) {
          mt[j + MP * c] = Ct1[j + M * c];
          mt[MP * MP + j + MP * c] = Pt[((size_t)(t + 1) * MM + j + M * c) * F];
        }
#pragma unroll
      for (int j = 0; j < M; j++) {
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < M; c++) s += Ct1[j + M * c] * ah0[c];
        mt[2 * MP * MP + j] = ah1[j] - s;
      }
    }
    double* row = obs + (size_t)t * OS;
    for (int q = 0; q < OS; q++) row[q] = 0.0;
    double* ob = row + ROW_HDR;
    const double y = in[(b.oy + t) * F], u = in[(b.ou + t) * F];
    ob[OB_DIST] = (double)dist;
    ob[OB_PHI] = phi;
    const bool seen = y == y && fabs(y) <= 1.7976931348623157e308;
    if (seen) {
      ob[OB_FLAG] = 1.0; ob[OB_Y] = y; ob[OB_U] = u;
      if (dist == DIST_SVM) {  // signal = first state; Z, D, x'beta are ignored (src/model_ssm_ung.cpp:456)
        ob[SER_HDR] = 1.0;
        ob[OB_AUX] = (y / phi) * (y / phi);
      } else {
        ob[OB_D] = in[(b.oD + (size_t)t * b.Ds) * F] + in[(b.ox + t) * F];
#pragma unroll kU
        for (int c = 0; c < M; c++) ob[SER_HDR + c] = in[(b.oZ + (size_t)t * b.Zs + c) * F];
        ob[OB_AUX] = dist == DIST_NEGBIN ? y + phi : (dist == DIST_GAMMA ? y / u : 0.0);
      }
      ob[OB_PSIY] = W.ayr[(size_t)t * F];
      ob[OB_PSIIH] = 1.0 / W.aH[(size_t)t * F];
      row[RH_FLAG] = 1.0;
      row[RH_SCALE] = W.scl[(size_t)t * F];
    }
    if (t == 0) {
      for (int q = 0; q < MP + MP * MP; q++) ini[q] = 0.0;
      if (ok) {
        const size_t em = W.em;
#pragma unroll kU
        for (int j = 0; j < M; j++) ini[j] = at[(size_t)j * em];
#pragma unroll kU
        for (int c = 0; c < M; c++)
#pragma unroll kU
          for (int j = 0; j < M; j++) ini[MP + j + MP * c] = Pt[(size_t)(j + M * c) * em];
      }
    }
  }
}
// Round 6, draw-major work arrays (UngSetupDev::dm, 32 / 64 padded states): the proposal operands of time point blockIdx.y of draw blockIdx.x --
// C_{t+1}, V_{t+1}^(1/2) and alphahat_{t+1} - C_{t+1} alphahat_t (build_psi_tables; src/model_ssm_ung.cpp:585-586) -- copied by a workgroup, coalesced on
// both sides (one lane per draw copying 2 x 64 x 64 strided doubles per time point took 8.6 ms per 256 draws of 53 states).  `sub` = virtual steps per
// time point (1, or p for ssm_mng draws).
template <int M>
__global__ void __launch_bounds__(256) setup_tables_big_kernel(const double* __restrict__ at_all, const double* __restrict__ Vf_all, const double* __restrict__ Cf_all,
                                                               const size_t at_draw, const size_t mat_draw, const int sub, double* __restrict__ mats, const int n,
                                                               const int* __restrict__ okv) {
  constexpr int MM = M * M, MS = 2 * MM + M;
  const int d = blockIdx.x, t = blockIdx.y, tid = threadIdx.x;
  double* __restrict__ mt = mats + ((size_t)d * n + t) * MS;
  if (!okv[d]) { for (int q = tid; q < MS; q += 256) mt[q] = 0.0; return; }
  const size_t i0 = (size_t)t * sub, i1 = (size_t)(t + 1) * sub;
  const double* __restrict__ c1 = Cf_all + (size_t)d * mat_draw + i1 * MM;
  const double* __restrict__ p1 = Vf_all + (size_t)d * mat_draw + i1 * MM;
  const double* __restrict__ a0 = at_all + (size_t)d * at_draw + i0 * M;
  const double* __restrict__ a1 = at_all + (size_t)d * at_draw + i1 * M;
  for (int q = tid; q < MM; q += 256) { mt[q] = c1[q]; mt[MM + q] = p1[q]; }
  if (tid < M) {
    double sm = 0.0;
    for (int c = 0; c < M; c++) sm += c1[tid + M * c] * a0[c];
    mt[2 * MM + tid] = a1[tid] - sm;
  }
}
// Kalman filter + covariance smoother of 17 ... 64 states (padded to 32 / 64): one workgroup of M / 16 waves per 64 / M draws, the matrices of the
// recursion in LDS (lg_smoother_wave.h) -- phase 1 of the set-up for these dimensions
template <int M>
__global__ void __launch_bounds__(LgWave<M>::kThreads, (M <= 32 ? 2 : 1)) ung_smooth_wave_kernel(const UngSetupDev b) {
  __shared__ double lds[LgWave<M>::kLdsDoubles];
  const int d = min((int)(blockIdx.x * (64 / M) + (threadIdx.x & 63) / M), b.F - 1);  // (surplus groups redo the last draw: every lane takes part in the barriers)
  const size_t F = (size_t)b.F;
  const UngWork<M> W(b, d);
  const double* in = W.in;
  LgModel g;
  g.n = b.n;
  g.Z = in + b.oZ * F; g.Zs = b.Zs; g.T = in + b.oT * F; g.Ts = b.Ts; g.RR = in + b.oR * F; g.Rs = b.Rs;
  g.D = in + b.oD * F; g.Ds = b.Ds; g.C = in + b.oC * F; g.Cs = b.Cs; g.y = W.ay; g.H = W.aH; g.Hs = 1;
  g.a1 = in + b.oa * F; g.P1 = in + b.oP * F;
  LgWave<M> wave(lds);
  wave.mr = (b.mr > 0 && b.mr < M) ? b.mr : M;
  const bool ok = wave.run(g, W.at, W.Pt, W.cc, W.K, W.v, W.Fs, F, W.em);
  if (wave.r == 0 && wave.w == 0) b.ok[d] = ok ? 1 : 0;
}
// conditional factors of states of sixteen and more (padded) dimensions: M lanes per (draw, time point), matrices in LDS (lg_factors_rows.h)
template <int M>
// (32 and 64 states: two waves per SIMD -- 256 registers, a few of them spilt: the CU's LDS holds two (64) / four (32) workgroups, the registers of
// the first build only half of that; 256 draws of 53 states 244 -> 148 ms, 512 draws of 25 states 39.6 -> 29.5 ms)
__global__ void __launch_bounds__(LgFactorRows<M>::kThreads, (M >= 32 ? 2 : 1)) ung_factors_rows_kernel(const UngSetupDev b) {
  constexpr int G = 64 / M;
  __shared__ double lds[LgFactorRows<M>::kLdsDoubles];
  const int d = min((int)(blockIdx.x * G + (threadIdx.x & 63) / M), b.F - 1);  // (surplus groups redo the last draw: every lane takes part in the barriers)
  const UngWork<M> W(b, d);
  LgFactorRows<M> rows(lds);
  rows.skip = b.ok[d] == 0;
  rows.mr = (b.mr > 0 && b.mr < M) ? b.mr : M;
  const int t = (int)blockIdx.y + b.t_off;
  const bool fine = rows.cond_step(t, W.Pt, W.cc, W.Vf, W.Cf, W.em, 0, (b.flags && t > 0) ? b.flags + (size_t)d * (b.n + 1) + t : nullptr);
  if (!fine && !rows.skip && rows.r == 0 && rows.w == 0) b.ok[d] = 0;  // (every writer stores 0)
}
// the thin path of time point blockIdx.y + 1 (lg_factors_rows.h: cond_step_thin): a kernel of its own, so that its registers are its own
template <int M>
__global__ void __launch_bounds__(LgFactorRows<M>::kThreads, (M >= 32 ? 2 : 1)) ung_factors_thin_kernel(const UngSetupDev b) {
  constexpr int G = 64 / M;
  __shared__ double lds[LgFactorRows<M>::kLdsDoubles];
  const int d = min((int)(blockIdx.x * G + (threadIdx.x & 63) / M), b.F - 1);
  const int t = (int)blockIdx.y + 1;
  const UngWork<M> W(b, d);
  LgFactorRows<M> rows(lds);
  rows.skip = b.ok[d] == 0;
  rows.mr = (b.mr > 0 && b.mr < M) ? b.mr : M;
  const bool fine = rows.cond_step_thin(t, W.Pt, W.cc, W.Vf, W.Cf, W.em, 0, b.flags + (size_t)d * (b.n + 1) + t, b.cond ? b.cond + (size_t)d * (b.n + 1) + t : nullptr);
  if (!fine && !rows.skip && rows.r == 0 && rows.w == 0) b.ok[d] = 0;
}
template <int M>
static hipError_t launch_ung_setup_m(const UngSetupDev& b, hipStream_t stream) {
  const unsigned gx = (unsigned)((b.F + 63) / 64);
  if (b.phases & 1) hipLaunchKernelGGL((ung_setup_kernel<M, 0, false>), dim3(gx, (unsigned)b.n), dim3(64), 0, stream, b);
  if (opt().trace >= 3) { const hipError_t de = hipStreamSynchronize(stream); std::fprintf(stderr, "[bssm_amd] ung set-up M=%d phase %d: %s\n", M, 0, hipGetErrorString(de)); }
  const bool lane_form = opt().mng_lane_per_draw;  // (comparison switch, shared with the ssm_mng set-up)
  constexpr int MR = (M == 8 || M == 4) ? M : 1;  // (as the ssm_mng set-up: rows for 8 and 4 states; 16 and 32: rows only, above)
  if (!(b.phases & 2)) {
  } else if constexpr (M >= 32) {
    hipLaunchKernelGGL((ung_smooth_wave_kernel<M>), dim3((unsigned)((b.F + 64 / M - 1) / (64 / M))), dim3(LgWave<M>::kThreads), 0, stream, b);
  } else if constexpr (M >= 16) {  // (no one-lane-per-draw form: five 16 x 16 matrices do not fit a lane)
    // late round 6: the matrix-core smoother for 16 padded states as well (2 048 draws of 13 states: 9.2 -> ... ms); BSSM_AMD_SETUP_ROWS16: the rows form
    if (!opt().setup_rows16) hipLaunchKernelGGL((ung_smooth_wave_kernel<M>), dim3((unsigned)((b.F + 64 / M - 1) / (64 / M))), dim3(LgWave<M>::kThreads), 0, stream, b);
    else hipLaunchKernelGGL((ung_setup_kernel<M, 1, true>), dim3((unsigned)((b.F + 64 / M - 1) / (64 / M))), dim3(64), 0, stream, b);
  } else {
    if ((M == 8 || M == 4) && !lane_form) hipLaunchKernelGGL((ung_setup_kernel<MR, 1, true>), dim3((unsigned)((b.F + 64 / MR - 1) / (64 / MR))), dim3(64), 0, stream, b);
    else hipLaunchKernelGGL((ung_setup_kernel<M, 1, false>), dim3(gx), dim3(64), 0, stream, b);
  }
  if (opt().trace >= 3) { const hipError_t de = hipStreamSynchronize(stream); std::fprintf(stderr, "[bssm_amd] ung set-up M=%d phase %d: %s\n", M, 1, hipGetErrorString(de)); }
  if (!(b.phases & 4)) {
  } else if constexpr (M >= 16) {
    const unsigned gg = (unsigned)((b.F + 64 / M - 1) / (64 / M));
    if (b.flags) {
      // t = 0 (always the full decomposition) beside the thin kernel; what the thin kernel flagged behind it
      hipStream_t side = opt().trace >= 3 ? nullptr : side_begin(stream);
      hipLaunchKernelGGL((ung_factors_rows_kernel<M>), dim3(gg, 1u), dim3(LgFactorRows<M>::kThreads), 0, side ? side : stream, b);
      hipLaunchKernelGGL((ung_factors_thin_kernel<M>), dim3(gg, (unsigned)b.n), dim3(LgFactorRows<M>::kThreads), 0, stream, b);
      UngSetupDev b1 = b;
      b1.t_off = 1;
      hipLaunchKernelGGL((ung_factors_rows_kernel<M>), dim3(gg, (unsigned)b.n), dim3(LgFactorRows<M>::kThreads), 0, stream, b1);
      if (side) { const hipError_t je = side_join(side, stream); if (je != hipSuccess) return je; }
    } else {
      hipLaunchKernelGGL((ung_factors_rows_kernel<M>), dim3(gg, (unsigned)b.n + 1), dim3(LgFactorRows<M>::kThreads), 0, stream, b);
    }
  }
  else hipLaunchKernelGGL((ung_setup_kernel<M, 2, false>), dim3(gx, (unsigned)b.n + 1), dim3(64), 0, stream, b);
  if (opt().trace >= 3) { const hipError_t de = hipStreamSynchronize(stream); std::fprintf(stderr, "[bssm_amd] ung set-up M=%d phase %d: %s\n", M, 2, hipGetErrorString(de)); }
  if (!(b.phases & 8)) return hipGetLastError();
  hipLaunchKernelGGL((ung_setup_kernel<M, 3, false>), dim3(gx, (unsigned)b.n), dim3(64), 0, stream, b);
  if constexpr (M >= 16) {
    if (b.dm) {   // the matrices of the tables from the draw-major work arrays (the region offsets of UngWork)
      const size_t F = (size_t)b.F, n = (size_t)b.n, n1 = n + 1, MMs = (size_t)M * M;
      double* w = b.scratch + 4 * n * F;
      const double* at_all = w; w += n1 * M * F + 2 * n1 * MMs * F + n * M * F + 2 * n * F;   // at | Pt, cc | K | v, Fs
      const double* Vf_all = w; w += n1 * MMs * F;
      const double* Cf_all = w;
      hipLaunchKernelGGL((setup_tables_big_kernel<M>), dim3((unsigned)b.F, (unsigned)b.n), dim3(256), 0, stream, at_all, Vf_all, Cf_all, n1 * M, n1 * MMs, 1, b.mats, b.n, b.ok);
    }
  }
  if (opt().trace >= 3) { const hipError_t de = hipStreamSynchronize(stream); std::fprintf(stderr, "[bssm_amd] ung set-up M=%d phase %d: %s\n", M, 3, hipGetErrorString(de)); }
  return hipGetLastError();
}
// diagnostic: (workgroups of the factor kernels that took the full path, all of them) since the last call; resets the counters
hipError_t debug_thin_counts(unsigned long long out[2]) {
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(lgf_thin_counts), 2 * sizeof(unsigned long long));
  if (e != hipSuccess) return e;
  const unsigned long long z[2] = {0ull, 0ull};
  return hipMemcpyToSymbol(HIP_SYMBOL(lgf_thin_counts), z, sizeof z);
}
hipError_t launch_ung_setup(const UngSetupDev& b, hipStream_t stream) {
  if (b.n < 1 || b.n >= 65535) return hipErrorInvalidValue;  // (time points ride in gridDim.y)
  switch (b.m) {
    case 1: return launch_ung_setup_m<1>(b, stream);
    case 2: return launch_ung_setup_m<2>(b, stream);
    case 3: return launch_ung_setup_m<3>(b, stream);
    case 4: return launch_ung_setup_m<4>(b, stream);
    case 5: return launch_ung_setup_m<5>(b, stream);
    case 6: return launch_ung_setup_m<6>(b, stream);
    case 7: return launch_ung_setup_m<7>(b, stream);
    case 8: return launch_ung_setup_m<8>(b, stream);
    case 16: return launch_ung_setup_m<16>(b, stream);  // (9 ... 16 states arrive padded to 16: ung_pack)
    case 32: return launch_ung_setup_m<32>(b, stream);  // (17 ... 32: padded to 32)
    case 64: return launch_ung_setup_m<64>(b, stream);  // (33 ... 64: padded to 64)
    default: return hipErrorInvalidValue;
  }
}
// ---------------------------------------------------------------- psi set-up of ssm_mng draws, all of it on the device
// The same for p > 1 independent observation series per time point (src/model_ssm_mng.cpp:117-133 approximate_for_is,
// :269-315 Laplace step, :218-255 scales): pseudo-observations per (time point, series), one scale per time point, the
// smoother over n p virtual steps (LgModel::sub: one series at a time, moments of time point s at index s p), tables
// with rows of p series.  What psi_setup_begin + lg_setup_kernel + psi_setup_compact + build_psi_tables do between host
// and device otherwise.
size_t mng_setup_scratch_doubles(int n, int m, int p) {
  const size_t mm = (size_t)m * m, nv = (size_t)n * p, n1 = nv + 1;
  //      ay  aH    scales  at       Pt        cc        K        v    Fs   Vf        Cf
  return 2 * nv + n + n1 * m + n1 * mm + n1 * mm + nv * m + nv + nv + n1 * mm + n1 * mm;
}
// Three kernels per chunk: (A) Laplace step + Kalman filter + covariance smoother, one lane per draw, sequential in time -- the only
// sequential part; (B) the conditional factors, independent across time points given the smoother's output: one wave per (64 draws,
// time point); (C) the tables, likewise.  As ONE kernel the factors (two 8 x 8 Jacobi eigen-decompositions per time point) and the last
// sweep were 60 % of 229 ms on the C4 shape (512 draws, m = 8, p = 4, T = 200: profiles/r03_setup_phases.txt).
template <int M>
struct MngWork {
  const double* in;
  double *ay, *aH, *scl, *at, *Pt, *cc, *K, *v, *Fs, *Vf, *Cf;
  size_t em;   // element stride of the matrix work arrays: F (draw-interleaved) or 1 (draw-major, UngSetupDev::dm)
  __device__ MngWork(const UngSetupDev& b, int d) {
    constexpr int MM = M * M;
    const size_t F = (size_t)b.F;
    const int n = b.n, nv = n * b.p;
    const bool dm = b.dm != 0;
    em = dm ? 1 : F;
    in = b.in + d;
    double* w = b.scratch;
    auto mat = [&](size_t per_draw) { double* p = dm ? w + (size_t)d * per_draw : w + d; w += per_draw * F; return p; };
    ay = w + d;     w += (size_t)nv * F;   // approximating y
    aH = w + d;     w += (size_t)nv * F;
    scl = w + d;    w += (size_t)n * F;
    at = mat((size_t)(nv + 1) * M);
    Pt = mat((size_t)(nv + 1) * MM);
    cc = mat((size_t)(nv + 1) * MM);
    K = mat((size_t)nv * M);
    v = w + d;      w += (size_t)nv * F;
    Fs = w + d;     w += (size_t)nv * F;
    Vf = mat((size_t)(nv + 1) * MM);  // V_t^(1/2) and C_t at the time points (virtual steps s p)
    Cf = mat((size_t)(nv + 1) * MM);
  }
};
// (L) Laplace step at the stored mode for time point blockIdx.y: approximating y / H of its p series and the scale (src/model_ssm_mng.cpp:117-133, 218-255, 269-315)
template <int M>
__global__ void __launch_bounds__(64) mng_laplace_kernel(const UngSetupDev b) {
  const int d = blockIdx.x * 64 + threadIdx.x;
  if (d >= b.F) return;
  const size_t F = (size_t)b.F;
  const int p = b.p, t = blockIdx.y;
  const MngWork<M> W(b, d);
  const double* in = W.in;
  double sc = 0.0;
  for (int j = 0; j < p; j++) {
    const size_t q = (size_t)t * p + j;
    const int dist = (int)in[(b.oh + j) * F];
    const double phi = in[(b.oh + p + j) * F];
    const double y = in[(b.oy + q) * F], sg = in[(b.om + q) * F], u = in[(b.ou + q) * F];
    double HH = 0.0, yy = 0.0, sv = 0.0;
    switch (dist) {
      case DIST_POISSON: HH = 1.0 / (exp(sg) * u); yy = y * HH + sg - 1.0; sv = y * sg - u * exp(sg); break;
      case DIST_BINOMIAL: { const double e = exp(sg); HH = (1.0 + e) * (1.0 + e) / (u * e); yy = y * HH + sg - 1.0 - e; sv = y * sg - u * log1p(e); } break;
      case DIST_NEGBIN: { const double e = exp(sg) * u; HH = (phi + e) * (phi + e) / (phi * e * (y + phi)); yy = sg + (phi + e) * (y - e) / ((y + phi) * e);
                          sv = y * sg - (y + phi) * log(phi + u * exp(sg)); } break;
      case DIST_GAMMA: { const double e = exp(sg) * u; HH = e / (y * phi); yy = sg - e / y + 1.0; sv = -phi * sg - (y * phi * exp(-sg) / u); } break;
      case DIST_GAUSSIAN: { HH = phi * phi; yy = y; const double r = (y - sg) / phi; sv = -0.5 * r * r; } break;
      default: break;
    }
    const double sd = sqrt(HH);
    const bool seen = y == y && fabs(y) <= 1.7976931348623157e308;
    const double r = (yy - sg) / sd;
    W.ay[q * F] = yy;
    W.aH[q * F] = sd;
    if (seen) sc += sv + 0.5 * r * r;
  }
  W.scl[(size_t)t * F] = sc;
}
// (A) Kalman filter + covariance smoother over the n p virtual steps, sequential in time.  ROWS = false: one lane per draw
// (LgSmoother<M>::run); ROWS = true: M lanes per draw, one matrix row each (lg_smoother_rows.h) -- 64 / M draws per single-wave workgroup
template <int M, bool ROWS>
__global__ void __launch_bounds__(64) mng_setup_kernel(const UngSetupDev b) {
  constexpr int DPW = ROWS ? 64 / M : 64;  // draws per workgroup
  const int lane = threadIdx.x;
  const int d0 = blockIdx.x * DPW + (ROWS ? lane / M : lane);
  if (!ROWS && d0 >= b.F) return;
  const int d = min(d0, b.F - 1);  // (rows: the surplus groups of the last workgroup redo the last draw -- every lane takes part in the LDS exchanges)
  const size_t F = (size_t)b.F;
  const int n = b.n, p = b.p, nv = n * p;
  const MngWork<M> W(b, d);
  const double* in = W.in;
  LgModel g;
  g.n = nv; g.sub = p;
  g.Z = in + b.oZ * F; g.Zs = b.Zs; g.T = in + b.oT * F; g.Ts = b.Ts; g.RR = in + b.oR * F; g.Rs = b.Rs;
  g.D = in + b.oD * F; g.Ds = b.Ds; g.C = in + b.oC * F; g.Cs = b.Cs; g.y = W.ay; g.H = W.aH; g.Hs = 1;
  g.a1 = in + b.oa * F; g.P1 = in + b.oP * F;
  bool ok;
  if constexpr (ROWS) {
    __shared__ double lds[LgRows<M>::kLdsDoubles];
    LgRows<M> rows(lds);
    ok = rows.run(g, W.at, W.Pt, W.cc, W.K, W.v, W.Fs, F);
  } else {
    LG_STAMP(1);
    ok = LgSmoother<M>::run(g, W.at, W.Pt, W.cc, W.K, W.v, W.Fs, F, false);  // at = alphahat, Pt = Var, cc = lag-one Cov
#ifdef BSSM_SETUP_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      const unsigned long long* q = g_setup_stamps;
      printf("mng_setup_kernel<%d> F=%d n=%d p=%d us: forward %.1f backward %.1f\n", M, b.F, n, p, (q[2] - q[1]) * 0.01, (q[3] - q[2]) * 0.01);
    }
#endif
  }
  b.ok[d] = ok ? 1 : 0;
}
// (B) conditional factors of time point blockIdx.y (virtual step blockIdx.y * p) for the 64 draws of blockIdx.x
template <int M>
__global__ void __launch_bounds__(64) mng_factors_kernel(const UngSetupDev b) {
  const int d = blockIdx.x * 64 + threadIdx.x;
  if (d >= b.F || !b.ok[d]) return;
  const MngWork<M> W(b, d);
  if (!LgSmoother<M>::cond_step((int)blockIdx.y * b.p, W.Pt, W.cc, W.Vf, W.Cf, (size_t)b.F)) b.ok[d] = 0;  // (every writer stores 0)
}
// (C) tables of time point blockIdx.y (build_psi_tables, fill_obs_rows): the moments of time point s sit at virtual step s p
template <int M>
__global__ void __launch_bounds__(64) mng_tables_kernel(const UngSetupDev b) {
  const int d = blockIdx.x * 64 + threadIdx.x;
  if (d >= b.F) return;
  constexpr int MM = M * M, MP = M <= 1 ? 1 : (M <= 2 ? 2 : (M <= 4 ? 4 : (M <= 8 ? 8 : M)));   // (16 / 32 / 64: the padded dimension itself, round 6)
  constexpr int kU = M >= 16 ? 1 : M;   // (the copy loops stay loops for the padded big states)
  const size_t F = (size_t)b.F;
  const int n = b.n, p = b.p, t = blockIdx.y;
  const MngWork<M> W(b, d);
  const double* in = W.in;
  const double *at = W.at, *Pt = W.Vf, *cc = W.Cf, *ay = W.ay, *aH = W.aH, *scl = W.scl;
  const bool ok = b.ok[d] != 0;
  const int sstride = SER_HDR + MP, OS = ROW_HDR + p * sstride;
  constexpr int MS = 2 * MP * MP + MP;
  double* __restrict__ mats = b.mats + (size_t)d * n * MS;
  double* __restrict__ obs = b.obs + (size_t)d * n * OS;
  double* __restrict__ ini = b.init + (size_t)d * (MP + MP * MP);
  if constexpr (M >= 16) {   // (no work arrays of a lane: a 16 x 16 matrix is 2 KB of scratch memory; MP == M here)
    double* mt = mats + (size_t)t * MS;
    if (!b.dm) { for (int q = 0; q < MS; q++) mt[q] = 0.0; }
    const size_t i0 = (size_t)t * p, i1 = (size_t)(t + 1) * p;
    if (ok && !b.dm) {   // (draw-major work arrays: setup_tables_big_kernel)
      const double* c1 = cc + i1 * MM * F;
      const double* p1 = Pt + i1 * MM * F;
      for (int q = 0; q < MM; q++) { mt[q] = c1[(size_t)q * F]; mt[MM + q] = p1[(size_t)q * F]; }
      for (int j = 0; j < M; j++) {
        double sm = 0.0;
        for (int c = 0; c < M; c++) sm += c1[(size_t)(j + M * c) * F] * at[(i0 * M + c) * F];
        mt[2 * MM + j] = at[(i1 * M + j) * F] - sm;
      }
    }
  } else {
    double* mt = mats + (size_t)t * MS;
    for (int q = 0; q < MS; q++) mt[q] = 0.0;
    const size_t i0 = (size_t)t * p, i1 = (size_t)(t + 1) * p;
    double ah0[M], ah1[M], Ct1[MM];
#pragma unroll
    for (int j = 0; j < M; j++) { ah0[j] = at[(i0 * M + j) * F]; ah1[j] = at[(i1 * M + j) * F]; }
#pragma unroll
    for (int q = 0; q < MM; q++) Ct1[q] = ok ? cc[(i1 * MM + q) * F] : 0.0;
    if (ok) {
#pragma unroll
      for (int c = 0; c < M; c++)
#pragma unroll
        for (int j = 0; j < M; j++) {
          mt[j + MP * c] = Ct1[j + M * c];
          mt[MP * MP + j + MP * c] = Pt[(i1 * MM + j + M * c) * F];
        }
#pragma unroll
      for (int j = 0; j < M; j++) {
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < M; c++) s += Ct1[j + M * c] * ah0[c];
        mt[2 * MP * MP + j] = ah1[j] - s;
      }
    }
  }
  {
    double* row = obs + (size_t)t * OS;
    for (int q = 0; q < OS; q++) row[q] = 0.0;
    bool any = false;
    for (int j = 0; j < p; j++) {
      double* ob = row + ROW_HDR + j * sstride;
      const size_t q = (size_t)t * p + j;
      const int dist = (int)in[(b.oh + j) * F];
      const double phi = in[(b.oh + p + j) * F];
      const double y = in[(b.oy + q) * F], u = in[(b.ou + q) * F];
      ob[OB_DIST] = (double)dist;
      ob[OB_PHI] = phi;
      const bool seen = y == y && fabs(y) <= 1.7976931348623157e308;
      if (!seen) continue;
      any = true;
      ob[OB_FLAG] = 1.0; ob[OB_Y] = y; ob[OB_U] = u;
      ob[OB_D] = in[(b.oD + (size_t)t * b.Ds + j) * F];
#pragma unroll kU
      for (int c = 0; c < M; c++) ob[SER_HDR + c] = in[(b.oZ + (size_t)t * b.Zs + (size_t)j * M + c) * F];
      ob[OB_AUX] = dist == DIST_NEGBIN ? y + phi : (dist == DIST_GAMMA ? y / u : (dist == DIST_GAUSSIAN ? 1.0 / (phi * phi) : 0.0));
      ob[OB_PSIY] = ay[q * F];
      ob[OB_PSIIH] = 1.0 / aH[q * F];
    }
    row[RH_FLAG] = any ? 1.0 : 0.0;
    row[RH_SCALE] = any ? scl[(size_t)t * F] : 0.0;
  }
  if (t == 0) {
    for (int q = 0; q < MP + MP * MP; q++) ini[q] = 0.0;
    if (ok) {
      const size_t em = W.em;
#pragma unroll kU
      for (int j = 0; j < M; j++) ini[j] = at[(size_t)j * em];
#pragma unroll kU
      for (int c = 0; c < M; c++)
#pragma unroll kU
        for (int j = 0; j < M; j++) ini[MP + j + MP * c] = Pt[(size_t)(j + M * c) * em];
    }
  }
}
// ---- round 6: ssm_mng draws of 9 ... 64 states (padded to 16 / 32 / 64 by mng_pack), the kernels of the big univariate states over the
// n p virtual steps: the smoother with its matrices in LDS (lg_smoother_wave.h, series at a time), the conditional factors M lanes per
// (draw, time point) at the time points' virtual steps (lg_factors_rows.h; src/model_ssm_mlg.cpp:310-396, src/conditional_dist.cpp:4-31)
template <int M>
__global__ void __launch_bounds__(LgWave<M>::kThreads, (M <= 32 ? 2 : 1)) mng_smooth_wave_kernel(const UngSetupDev b) {
  __shared__ double lds[LgWave<M>::kLdsDoubles];
  const int d = min((int)(blockIdx.x * (64 / M) + (threadIdx.x & 63) / M), b.F - 1);  // (surplus groups redo the last draw: every lane takes part in the barriers)
  const size_t F = (size_t)b.F;
  const MngWork<M> W(b, d);
  const double* in = W.in;
  LgModel g;
  g.n = b.n * b.p; g.sub = b.p;
  g.Z = in + b.oZ * F; g.Zs = b.Zs; g.T = in + b.oT * F; g.Ts = b.Ts; g.RR = in + b.oR * F; g.Rs = b.Rs;
  g.D = in + b.oD * F; g.Ds = b.Ds; g.C = in + b.oC * F; g.Cs = b.Cs; g.y = W.ay; g.H = W.aH; g.Hs = 1;
  g.a1 = in + b.oa * F; g.P1 = in + b.oP * F;
  LgWave<M> wave(lds);
  wave.mr = (b.mr > 0 && b.mr < M) ? b.mr : M;
  const bool ok = wave.run(g, W.at, W.Pt, W.cc, W.K, W.v, W.Fs, F, W.em);
  if (wave.r == 0 && wave.w == 0) b.ok[d] = ok ? 1 : 0;
}
template <int M>
__global__ void __launch_bounds__(LgFactorRows<M>::kThreads, (M >= 32 ? 2 : 1)) mng_factors_rows_kernel(const UngSetupDev b) {
  constexpr int G = 64 / M;
  __shared__ double lds[LgFactorRows<M>::kLdsDoubles];
  const int d = min((int)(blockIdx.x * G + (threadIdx.x & 63) / M), b.F - 1);
  const MngWork<M> W(b, d);
  LgFactorRows<M> rows(lds);
  rows.skip = b.ok[d] == 0;
  rows.mr = (b.mr > 0 && b.mr < M) ? b.mr : M;
  const int t = (int)blockIdx.y + b.t_off;
  const bool fine = rows.cond_step(t * b.p, W.Pt, W.cc, W.Vf, W.Cf, W.em, 0, (b.flags && t > 0) ? b.flags + (size_t)d * (b.n + 1) + t : nullptr);
  if (!fine && !rows.skip && rows.r == 0 && rows.w == 0) b.ok[d] = 0;  // (every writer stores 0)
}
template <int M>
__global__ void __launch_bounds__(LgFactorRows<M>::kThreads, (M >= 32 ? 2 : 1)) mng_factors_thin_kernel(const UngSetupDev b) {
  constexpr int G = 64 / M;
  __shared__ double lds[LgFactorRows<M>::kLdsDoubles];
  const int d = min((int)(blockIdx.x * G + (threadIdx.x & 63) / M), b.F - 1);
  const int t = (int)blockIdx.y + 1;
  const MngWork<M> W(b, d);
  LgFactorRows<M> rows(lds);
  rows.skip = b.ok[d] == 0;
  rows.mr = (b.mr > 0 && b.mr < M) ? b.mr : M;
  const bool fine = rows.cond_step_thin(t * b.p, W.Pt, W.cc, W.Vf, W.Cf, W.em, 0, b.flags + (size_t)d * (b.n + 1) + t, b.cond ? b.cond + (size_t)d * (b.n + 1) + t : nullptr);
  if (!fine && !rows.skip && rows.r == 0 && rows.w == 0) b.ok[d] = 0;
}
template <int M>
static hipError_t launch_mng_setup_big(const UngSetupDev& b, hipStream_t stream) {
  const unsigned gx = (unsigned)((b.F + 63) / 64), gg = (unsigned)((b.F + 64 / M - 1) / (64 / M));
  hipLaunchKernelGGL(mng_laplace_kernel<M>, dim3(gx, (unsigned)b.n), dim3(64), 0, stream, b);
  if (M >= 32 || !opt().setup_rows16) hipLaunchKernelGGL((mng_smooth_wave_kernel<M>), dim3(gg), dim3(LgWave<M>::kThreads), 0, stream, b);
  else hipLaunchKernelGGL((mng_setup_kernel<M, true>), dim3(gg), dim3(64), 0, stream, b);
  if (b.flags) {   // (as launch_ung_setup_m: t = 0 beside the thin kernel)
    hipStream_t side = side_begin(stream);
    hipLaunchKernelGGL((mng_factors_rows_kernel<M>), dim3(gg, 1u), dim3(LgFactorRows<M>::kThreads), 0, side ? side : stream, b);
    hipLaunchKernelGGL((mng_factors_thin_kernel<M>), dim3(gg, (unsigned)b.n), dim3(LgFactorRows<M>::kThreads), 0, stream, b);
    UngSetupDev b1 = b;
    b1.t_off = 1;
    hipLaunchKernelGGL((mng_factors_rows_kernel<M>), dim3(gg, (unsigned)b.n), dim3(LgFactorRows<M>::kThreads), 0, stream, b1);
    if (side) { const hipError_t je = side_join(side, stream); if (je != hipSuccess) return je; }
  } else {
    hipLaunchKernelGGL((mng_factors_rows_kernel<M>), dim3(gg, (unsigned)b.n + 1), dim3(LgFactorRows<M>::kThreads), 0, stream, b);
  }
  hipLaunchKernelGGL(mng_tables_kernel<M>, dim3(gx, (unsigned)b.n), dim3(64), 0, stream, b);
  if constexpr (M >= 16) {
    if (b.dm) {   // (the region offsets of MngWork)
      const size_t F = (size_t)b.F, n = (size_t)b.n, nv = n * b.p, n1 = nv + 1, MMs = (size_t)M * M;
      double* w = b.scratch + (2 * nv + n) * F;
      const double* at_all = w; w += n1 * M * F + 2 * n1 * MMs * F + nv * M * F + 2 * nv * F;
      const double* Vf_all = w; w += n1 * MMs * F;
      const double* Cf_all = w;
      hipLaunchKernelGGL((setup_tables_big_kernel<M>), dim3((unsigned)b.F, (unsigned)b.n), dim3(256), 0, stream, at_all, Vf_all, Cf_all, n1 * M, n1 * MMs, b.p, b.mats, b.n, b.ok);
    }
  }
  return hipGetLastError();
}
template <int M>
static hipError_t launch_mng_setup_m(const UngSetupDev& b, hipStream_t stream) {
  const unsigned gx = (unsigned)((b.F + 63) / 64);
  hipLaunchKernelGGL(mng_laplace_kernel<M>, dim3(gx, (unsigned)b.n), dim3(64), 0, stream, b);
  // (BSSM_AMD_MNG_LANE_PER_DRAW=1 keeps the one-lane form for comparison)
  const bool lane_form = opt().mng_lane_per_draw;
  // row-distributed for 8 and 4 states (M lanes per draw must tile the wave): m = 8 cannot be held by one lane (49 -> 5.2 ms on the C4
  // shape); m = 4 fits one lane but gains from four times as many waves (1 024 draws, p = 3, T = 300: the call 0.0485 -> 0.0436 s)
  constexpr int MR = (M == 8 || M == 4) ? M : 1;
  if ((M == 8 || M == 4) && !lane_form) hipLaunchKernelGGL((mng_setup_kernel<MR, true>), dim3((unsigned)((b.F + 64 / MR - 1) / (64 / MR))), dim3(64), 0, stream, b);
  else hipLaunchKernelGGL((mng_setup_kernel<M, false>), dim3(gx), dim3(64), 0, stream, b);
  hipLaunchKernelGGL(mng_factors_kernel<M>, dim3(gx, (unsigned)b.n + 1), dim3(64), 0, stream, b);
  hipLaunchKernelGGL(mng_tables_kernel<M>, dim3(gx, (unsigned)b.n), dim3(64), 0, stream, b);
  return hipGetLastError();
}
hipError_t launch_mng_setup(const UngSetupDev& b, hipStream_t stream) {
  if (b.n < 1 || b.n >= 65535) return hipErrorInvalidValue;  // (time points ride in gridDim.y)
  switch (b.m) {
    case 1: return launch_mng_setup_m<1>(b, stream);
    case 2: return launch_mng_setup_m<2>(b, stream);
    case 3: return launch_mng_setup_m<3>(b, stream);
    case 4: return launch_mng_setup_m<4>(b, stream);
    case 5: return launch_mng_setup_m<5>(b, stream);
    case 6: return launch_mng_setup_m<6>(b, stream);
    case 7: return launch_mng_setup_m<7>(b, stream);
    case 8: return launch_mng_setup_m<8>(b, stream);
    case 16: return launch_mng_setup_big<16>(b, stream);  // (9 ... 16 states arrive padded to 16: mng_pack, round 6)
    case 32: return launch_mng_setup_big<32>(b, stream);  // (17 ... 32)
    case 64: return launch_mng_setup_big<64>(b, stream);  // (33 ... 64)
    default: return hipErrorInvalidValue;
  }
}
// out[c * rows + r] = in[r * cols + c], 32 x 32 tiles through LDS (both sides coalesced): per-draw contiguous <->
// draw-interleaved layouts of the smoother batch
__global__ void __launch_bounds__(256) transpose_kernel(const double* __restrict__ in, double* __restrict__ out, int rows, int cols) {
  __shared__ double tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) {
    const int r = r0 + k, c = c0 + tx;
    if (r < rows && c < cols) tile[k][tx] = in[(size_t)r * cols + c];
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k, r = r0 + tx;
    if (r < rows && c < cols) out[(size_t)c * rows + r] = tile[tx][k];
  }
}
hipError_t launch_transpose(const double* in, double* out, int rows, int cols, hipStream_t stream) {
  hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, stream, in, out, rows, cols);
  return hipGetLastError();
}

// known-traffic kernel for calibrating the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on this library's own access
// pattern (8 bytes per lane, coalesced): reads n doubles and writes n doubles, nothing else
__global__ void __launch_bounds__(256) calib_copy_kernel(const double* __restrict__ in, double* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i] + 1.0;
}
hipError_t launch_calib_copy(const double* in, double* out, size_t n, hipStream_t stream) {
  hipLaunchKernelGGL(calib_copy_kernel, dim3(8192), dim3(256), 0, stream, in, out, n);
  return hipGetLastError();
}

// ---------------------------------------------------------------- stratified_sample on its own
// src/stratified_sample.cpp:9-28 as the exported helper `_bssm_stratified_sample`: p <- cumsum(p), p[last] = 1,
// sample j -> first k with (r_j + j) / N <= p_k.  The cumulative sum is a SEQUENTIAL fp64 recurrence in the
// reference (arma::cumsum); one lane reproduces it term by term so the returned indices are bit-for-bit the
// reference's (a tree scan rounds differently and can move a boundary).  The N searches then run in parallel.
__global__ void cumsum_sequential_kernel(double* __restrict__ p, int len) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  double s = 0.0;
  for (int k = 0; k < len; k++) { s += p[k]; p[k] = s; }
  p[len - 1] = 1.0;
}
__global__ void __launch_bounds__(256) stratified_search_kernel(const double* __restrict__ cdf, const double* __restrict__ r,
                                                                uint32_t* __restrict__ out, int len, int N) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  const double v = (r[j] + (double)j) * (1.0 / (double)N);  // (r(j) + j) * alpha, alpha = 1.0 / N (:17,:19)
  int lo = 0, hi = len - 1;  // first k with v <= cdf[k]; none (v > 1) -> len - 1 (:24-27)
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (v <= cdf[mid]) hi = mid; else lo = mid + 1;
  }
  out[j] = (uint32_t)lo;
}
hipError_t launch_stratified_sample(double* p, const double* r, uint32_t* out, int len, int N, hipStream_t stream) {
  hipLaunchKernelGGL(cumsum_sequential_kernel, dim3(1), dim3(64), 0, stream, p, len);
  hipLaunchKernelGGL(stratified_search_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, p, r, out, len, N);
  return hipGetLastError();
}

// run-time compiled user models: launchers exported by the user's library (user_model_tu.hip), by slot (1-based)
namespace {
struct UserLaunchers { user_wg_launcher_t wg; user_grid_launcher_t grid; };
UserLaunchers g_user[64];
int g_user_count = 0;
}  // namespace
static bool g_user_rows_only[64];
void mark_user_slot_rows_only(int slot) { if (slot >= 1 && slot <= 63) g_user_rows_only[slot] = true; }
bool user_slot_rows_only(int slot) { return slot >= 1 && slot <= 63 && g_user_rows_only[slot]; }
hipError_t launch_loglik_finalize(const FilterBatch& b, const GridWorkspace& ws, hipStream_t stream) {
  hipLaunchKernelGGL(loglik_finalize_kernel, dim3(b.F), dim3(64), 0, stream, b.loglik, ws.llinc, b.dead_at, b.n, b.F);
  return hipGetLastError();
}
int register_user_launchers(user_wg_launcher_t wg, user_grid_launcher_t grid) {
  if (g_user_count >= 63) return -1;
  g_user[++g_user_count] = UserLaunchers{wg, grid};
  return g_user_count;
}

// process-wide counts of filter batches launched per kernel family (any host thread; post-correction filters on a worker thread)
static std::atomic<unsigned long long> g_family_launches[3];
void family_launches(unsigned long long (&out)[3]) { for (int i = 0; i < 3; i++) out[i] = g_family_launches[i].load(); }

hipError_t launch_wg_filter(const FilterBatch& b, hipStream_t stream) {
  g_family_launches[0]++;
  if (b.user_slot > 0) return b.user_slot <= g_user_count ? g_user[b.user_slot].wg(&b, stream) : hipErrorInvalidValue;
  switch (b.MP) {
    case 1: return launch_wg_mp<1>(b, stream);
    case 2: return launch_wg_mp<2>(b, stream);
    case 4: return launch_wg_mp<4>(b, stream);
    case 8: return launch_wg_mp<8>(b, stream);
    case 16: return launch_wg_mp<16>(b, stream);
    case 32: return launch_wg_mp<32>(b, stream);
    case 64: return launch_wg_mp<64>(b, stream);
    default: return hipErrorInvalidValue;
  }
}
hipError_t launch_grid_filter(const FilterBatch& b, const GridWorkspace& ws, hipStream_t stream, StepTiming* timing) {
  g_family_launches[ws.chunk == CANON_ROW ? 2 : 1]++;
  if (b.user_slot > 0) return b.user_slot <= g_user_count ? g_user[b.user_slot].grid(&b, &ws, stream, timing) : hipErrorInvalidValue;
  switch (b.MP) {
    case 1: return launch_grid_mp<1>(b, ws, stream, timing);
    case 2: return launch_grid_mp<2>(b, ws, stream, timing);
    case 4: return launch_grid_mp<4>(b, ws, stream, timing);
    case 8: return launch_grid_mp<8>(b, ws, stream, timing);
    case 16: return launch_grid_mp<16>(b, ws, stream, timing);
    case 32: return launch_grid_mp<32>(b, ws, stream, timing);
    case 64: return launch_grid_mp<64>(b, ws, stream, timing);
    default: return hipErrorInvalidValue;
  }
}

// ------------------------------------------------------------------------------------------
// smoothing and summaries
// ------------------------------------------------------------------------------------------
// src/filter_smoother.cpp:5-18: every trajectory is an independent pointer chase through the
// ancestor table, so it parallelises over particles with no synchronisation at all.
__global__ void backtrace_kernel(const double* __restrict__ hist, const uint32_t* __restrict__ indices,
                                 double* __restrict__ out, int n, int m, int N) {
  const int f = blockIdx.y;
  const double* __restrict__ h = hist + (size_t)f * (n + 1) * m * N;
  const uint32_t* __restrict__ idx = indices + (size_t)f * n * N;
  double* __restrict__ o = out + (size_t)f * (n + 1) * m * N;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    for (int c = 0; c < m; c++) o[((size_t)n * m + c) * N + i] = h[((size_t)n * m + c) * N + i];
    uint32_t bi = (uint32_t)i;
    for (int t = n - 1; t >= 0; t--) {
      bi = idx[(size_t)t * N + bi];
      for (int c = 0; c < m; c++) o[((size_t)t * m + c) * N + i] = h[((size_t)t * m + c) * N + bi];
    }
  }
}

hipError_t launch_backtrace(const double* hist, const uint32_t* indices, double* out, int F, int n, int m, int N,
                            hipStream_t stream) {
  const int bx = (N + 255) / 256;
  hipLaunchKernelGGL(backtrace_kernel, dim3(bx, F), dim3(256), 0, stream, hist, indices, out, n, m, N);
  return hipGetLastError();
}

__device__ __forceinline__ double block_sum_256(double v, double* s_red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) s_red[wave] = v;
  __syncthreads();
  return ((s_red[0] + s_red[1]) + s_red[2]) + s_red[3];
}

// summary(): mean and population covariance over particles, two-pass (the reference's Welford
// recursion, src/summary.cpp:4-17, is a serial formulation of the same quantities).
__global__ void __launch_bounds__(256) summary_kernel(const double* __restrict__ hist, double* __restrict__ mean,
                                                      double* __restrict__ cov, int n1, int m, int N) {
  __shared__ double s_red[4];
  __shared__ double s_mean[64];
  const int t = blockIdx.x, f = blockIdx.y;
  const double* __restrict__ h = hist + ((size_t)f * n1 + t) * m * N;
  for (int j = 0; j < m; j++) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) acc += h[(size_t)j * N + i];
    const double tot = block_sum_256(acc, s_red);
    if (threadIdx.x == 0) { s_mean[j] = tot / N; mean[((size_t)f * n1 + t) * m + j] = tot / N; }
  }
  __syncthreads();
  for (int c = 0; c < m; c++)
    for (int j = 0; j < m; j++) {
      const double mj = s_mean[j], mc = s_mean[c];
      double acc = 0.0;
      for (int i = threadIdx.x; i < N; i += 256) acc += (h[(size_t)j * N + i] - mj) * (h[(size_t)c * N + i] - mc);
      const double tot = block_sum_256(acc, s_red);
      if (threadIdx.x == 0) cov[(((size_t)f * n1 + t) * m + c) * m + j] = tot / N;
    }
}

hipError_t launch_summary(const double* hist, double* mean, double* cov, int F, int n1, int m, int N,
                          hipStream_t stream) {
  if (m > 64) return hipErrorInvalidValue;
  hipLaunchKernelGGL(summary_kernel, dim3(n1, F), dim3(256), 0, stream, hist, mean, cov, n1, m, N);
  return hipGetLastError();
}

// ---------------------------------------------------------------- IS post-correction outputs without a smoothed copy
// output_type 1 (src/approx_mcmc.cpp:389-393): ONE back-traced trajectory per draw is kept, so only that one is
// traced (filter_smoother.cpp:7-16 restricted to particle `pick`): out [f][t][c], t in [0, n], slot n left as is.
__global__ void pick_trajectory_kernel(const double* __restrict__ hist, const uint32_t* __restrict__ indices,
                                       const uint32_t* __restrict__ picks, double* __restrict__ out, int F, int n, int m,
                                       int N) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  const double* __restrict__ h = hist + (size_t)f * (n + 1) * m * N;
  const uint32_t* __restrict__ idx = indices + (size_t)f * n * N;
  double* __restrict__ o = out + (size_t)f * (n + 1) * m;
  uint32_t b = picks[f];
  for (int c = 0; c < m; c++) o[(size_t)n * m + c] = h[((size_t)n * m + c) * N + b];
  for (int t = n - 1; t >= 0; t--) {
    b = idx[(size_t)t * N + b];
    for (int c = 0; c < m; c++) o[(size_t)t * m + c] = h[((size_t)t * m + c) * N + b];
  }
}
hipError_t launch_pick_trajectory(const double* hist, const uint32_t* indices, const uint32_t* picks, double* out, int F,
                                  int n, int m, int N, hipStream_t stream) {
  hipLaunchKernelGGL(pick_trajectory_kernel, dim3((F + 63) / 64), dim3(64), 0, stream, hist, indices, picks, out, F, n, m, N);
  return hipGetLastError();
}

// output_type 2 (src/approx_mcmc.cpp:394-400): filter_smoother + summary fused.  One workgroup per draw walks the
// ancestor table backwards with all N trajectories in step (b_i <- idx(b_i, t), filter_smoother.cpp:11-15) and
// reduces mean / population covariance of the gathered states at every time point (summary.cpp:4-17), so the
// back-traced particle system is never written: traffic per particle-step is one index gather + the state gathers
// instead of a full smoothed copy (write 8m) and its re-reads.  bcur [F][N] holds the current ancestors.
constexpr int SS_NT = 1024;
#ifndef BSSM_SS_NT
#define BSSM_SS_NT 1   // history and ancestor table read with non-temporal loads (each entry is read once)
#endif
constexpr int SS_RPT = 4;  // trajectories per thread of the register-resident variant (N <= 4096 at 1024 threads)
template <int M, bool REG = false>
__global__ void __launch_bounds__(SS_NT) smooth_summary_kernel(const double* __restrict__ hist,
                                                               const uint32_t* __restrict__ indices,
                                                               uint32_t* __restrict__ bcur_all, double* __restrict__ mean,
                                                               double* __restrict__ cov, int n, int m, int N) {
  constexpr int NC = M * (M + 1) / 2, NW = SS_NT / 64;
  __shared__ double s_part[NW][NC > M ? NC : M];
  __shared__ double s_mean[M];
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n1 = n + 1;
  const double* __restrict__ h = hist + (size_t)f * n1 * m * N;
  const uint32_t* __restrict__ idx = indices + (size_t)f * n * N;
  uint32_t* __restrict__ bcur = bcur_all + (size_t)f * N;
  const double dN = (double)N;
  uint32_t rb[REG ? SS_RPT : 1];
  double rv[REG ? SS_RPT : 1][M];
  for (int t = n; t >= 0; t--) {
    const double* __restrict__ ht = h + (size_t)t * m * N;
    // ---- advance the ancestors and accumulate the sums
    double acc[M];
#pragma unroll
    for (int c = 0; c < M; c++) acc[c] = 0.0;
    if constexpr (REG) {
      // N <= SS_RPT * SS_NT: a thread keeps the current ancestors of its trajectories and the values it gathered in
      // registers -- no ancestor array in global memory, one gather per trajectory and time point (same summation order)
#pragma unroll
      for (int r = 0; r < SS_RPT; r++) {
        const int i = tid + r * SS_NT;
        if (i < N) {
          const uint32_t b = (t == n) ? (uint32_t)i : (BSSM_SS_NT ? __builtin_nontemporal_load(&idx[(size_t)t * N + rb[r]]) : idx[(size_t)t * N + rb[r]]);
          rb[r] = b;
#pragma unroll
          for (int c = 0; c < M; c++) { rv[r][c] = c < m ? (BSSM_SS_NT ? __builtin_nontemporal_load(&ht[(size_t)c * N + b]) : ht[(size_t)c * N + b]) : 0.0; acc[c] += rv[r][c]; }
        }
      }
    } else {
    for (int i = tid; i < N; i += SS_NT) {
      const uint32_t b = (t == n) ? (uint32_t)i : idx[(size_t)t * N + bcur[i]];
      bcur[i] = b;
#pragma unroll
      for (int c = 0; c < M; c++) if (c < m) acc[c] += ht[(size_t)c * N + b];
    }
    }
#pragma unroll
    for (int c = 0; c < M; c++) { const double v = wave_sum(acc[c]); if (lane == 0) s_part[wave][c] = v; }
    __syncthreads();
    if (tid < m) {
      double tot = 0.0;
      for (int w = 0; w < NW; w++) tot += s_part[w][tid];
      s_mean[tid] = tot / dN;
      mean[((size_t)f * n1 + t) * m + tid] = tot / dN;
    }
    __syncthreads();
    // ---- second pass: centred cross products (lower triangle)
    double mu[M], cc[NC];
#pragma unroll
    for (int c = 0; c < M; c++) mu[c] = c < m ? s_mean[c] : 0.0;
#pragma unroll
    for (int q = 0; q < NC; q++) cc[q] = 0.0;
    if constexpr (REG) {
#pragma unroll
      for (int r = 0; r < SS_RPT; r++) {
        if (tid + r * SS_NT < N) {
          double d[M];
#pragma unroll
          for (int c = 0; c < M; c++) d[c] = c < m ? rv[r][c] - mu[c] : 0.0;
          int q = 0;
#pragma unroll
          for (int c = 0; c < M; c++)
#pragma unroll
            for (int j = 0; j <= c; j++) cc[q++] += d[c] * d[j];
        }
      }
    } else {
    for (int i = tid; i < N; i += SS_NT) {
      const uint32_t b = bcur[i];
      double d[M];
#pragma unroll
      for (int c = 0; c < M; c++) d[c] = c < m ? ht[(size_t)c * N + b] - mu[c] : 0.0;
      int q = 0;
#pragma unroll
      for (int c = 0; c < M; c++)
#pragma unroll
        for (int j = 0; j <= c; j++) cc[q++] += d[c] * d[j];
    }
    }
#pragma unroll
    for (int q = 0; q < NC; q++) { const double v = wave_sum(cc[q]); if (lane == 0) s_part[wave][q] = v; }
    __syncthreads();
    if (tid < NC) {
      int c = 0;
      while ((c + 1) * (c + 2) / 2 <= tid) c++;
      const int j = tid - c * (c + 1) / 2;
      if (c < m) {
        double tot = 0.0;
        for (int w = 0; w < NW; w++) tot += s_part[w][tid];
        cov[(((size_t)f * n1 + t) * m + c) * m + j] = tot / dN;
        cov[(((size_t)f * n1 + t) * m + j) * m + c] = tot / dN;
      }
    }
    __syncthreads();
  }
}
hipError_t launch_smooth_summary(const double* hist, const uint32_t* indices, uint32_t* bcur, double* mean, double* cov, int F,
                                 int n, int m, int N, hipStream_t stream) {
  // register-resident variant (no ancestor array in memory, one gather per trajectory and time point): C5 output 2
  const bool reg = N <= SS_RPT * SS_NT && m <= 4 && !opt().ss_global;
  if (reg && m <= 1) hipLaunchKernelGGL((smooth_summary_kernel<1, true>), dim3(F), dim3(SS_NT), 0, stream, hist, indices, bcur, mean, cov, n, m, N);
  else if (reg && m <= 2) hipLaunchKernelGGL((smooth_summary_kernel<2, true>), dim3(F), dim3(SS_NT), 0, stream, hist, indices, bcur, mean, cov, n, m, N);
  else if (reg) hipLaunchKernelGGL((smooth_summary_kernel<4, true>), dim3(F), dim3(SS_NT), 0, stream, hist, indices, bcur, mean, cov, n, m, N);
  else if (m <= 1) hipLaunchKernelGGL(smooth_summary_kernel<1>, dim3(F), dim3(SS_NT), 0, stream, hist, indices, bcur, mean, cov, n, m, N);
  else if (m <= 2) hipLaunchKernelGGL(smooth_summary_kernel<2>, dim3(F), dim3(SS_NT), 0, stream, hist, indices, bcur, mean, cov, n, m, N);
  else if (m <= 4) hipLaunchKernelGGL(smooth_summary_kernel<4>, dim3(F), dim3(SS_NT), 0, stream, hist, indices, bcur, mean, cov, n, m, N);
  else if (m <= 8) hipLaunchKernelGGL(smooth_summary_kernel<8>, dim3(F), dim3(SS_NT), 0, stream, hist, indices, bcur, mean, cov, n, m, N);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// filter_summary(): src/summary.cpp:42-79
__global__ void __launch_bounds__(256) filter_summary_kernel(const double* __restrict__ hist,
                                                             const double* __restrict__ weights,
                                                             double* __restrict__ at, double* __restrict__ att,
                                                             double* __restrict__ Pt, double* __restrict__ Ptt, int n,
                                                             int m, int N) {
  __shared__ double s_red[4];
  __shared__ double s_at[64], s_att[64];
  const int t = blockIdx.x, f = blockIdx.y, n1 = n + 1;
  const double* __restrict__ h = hist + ((size_t)f * n1 + t) * m * N;
  const double* __restrict__ w = weights + ((size_t)f * n1 + t) * N;
  const bool filt = t < n;
  double wsum = 1.0;
  if (filt) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) acc += w[i];
    wsum = block_sum_256(acc, s_red);
  }
  for (int j = 0; j < m; j++) {
    double acc = 0.0, accw = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) {
      const double xv = h[(size_t)j * N + i];
      acc += xv;
      if (filt) accw += xv * (w[i] / wsum);
    }
    const double tot = block_sum_256(acc, s_red);
    const double totw = filt ? block_sum_256(accw, s_red) : 0.0;
    if (threadIdx.x == 0) {
      s_at[j] = tot / N;
      at[((size_t)f * n1 + t) * m + j] = tot / N;
      if (filt) { s_att[j] = totw; att[((size_t)f * n + t) * m + j] = totw; }
    }
  }
  __syncthreads();
  for (int c = 0; c < m; c++)
    for (int j = 0; j < m; j++) {
      double acc = 0.0, accw = 0.0;
      for (int i = threadIdx.x; i < N; i += 256) {
        const double xj = h[(size_t)j * N + i], xc = h[(size_t)c * N + i];
        acc += (xj - s_at[j]) * (xc - s_at[c]);
        if (filt) accw += (w[i] / wsum) * (xj - s_att[j]) * (xc - s_att[c]);
      }
      const double tot = block_sum_256(acc, s_red);
      const double totw = filt ? block_sum_256(accw, s_red) : 0.0;
      if (threadIdx.x == 0) {
        Pt[(((size_t)f * n1 + t) * m + c) * m + j] = tot / N;
        if (filt) Ptt[(((size_t)f * n + t) * m + c) * m + j] = totw;
      }
    }
}

hipError_t launch_filter_summary(const double* hist, const double* weights, double* at, double* att, double* Pt,
                                 double* Ptt, int F, int n, int m, int N, hipStream_t stream) {
  if (m > 64) return hipErrorInvalidValue;
  hipLaunchKernelGGL(filter_summary_kernel, dim3(n + 1, F), dim3(256), 0, stream, hist, weights, at, att, Pt, Ptt, n, m,
                     N);
  return hipGetLastError();
}

// particle-minor history -> Armadillo cube m x (n+1) x N.  LDS-tiled transpose of the
// [(t,c)][i] matrix so both the reads and the writes are coalesced.
__global__ void __launch_bounds__(256) to_cube_kernel(const double* __restrict__ hist, double* __restrict__ cube,
                                                      int rows /* (n+1)*m */, int N) {
  __shared__ double tile[32][33];
  const int f = blockIdx.z;
  const double* __restrict__ h = hist + (size_t)f * rows * N;
  double* __restrict__ o = cube + (size_t)f * rows * N;
  const int i0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int k = ty; k < 32; k += 8) {
    const int r = r0 + k, i = i0 + tx;
    if (r < rows && i < N) tile[k][tx] = h[(size_t)r * N + i];
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int i = i0 + k, r = r0 + tx;
    if (r < rows && i < N) o[(size_t)i * rows + r] = tile[tx][k];
  }
}

hipError_t launch_to_cube(const double* hist, double* cube, int F, int n1, int m, int N, hipStream_t stream) {
  const int rows = n1 * m;
  hipLaunchKernelGGL(to_cube_kernel, dim3((N + 31) / 32, (rows + 31) / 32, F), dim3(256), 0, stream, hist, cube, rows,
                     N);
  return hipGetLastError();
}

}  // namespace bssm
