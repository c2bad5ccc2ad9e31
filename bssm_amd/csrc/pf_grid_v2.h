// pf_grid_v2.h -- round-2 step kernel of the multi-workgroup filter (univariate-observation, linear-state models:
// ssm_ung / bsm_ng / svm / ar1_ng / ssm_ulg, bootstrap and psi-APF; src/model_ssm_ung.cpp:511-712).
//
// Same exchange between launches as pf_grid_step (block-local scans `lcdf`, row ends `coarse`, (max, sum) partials; the
// kernel boundary is the grid-wide synchronisation) and the same random stream, so results are interchangeable.
// What changed, and why (PMC of round 1: a wave executes VALU during only ~21 % of its lifetime, the launch pair is
// VALU-busy 66 %: the step is bound by its dependent chain -- 7 workgroup barriers and, per particle of a lane, three
// dependent global round trips -- as much as by instruction count):
//   * the partial combine is WAVE-LOCAL: a filter has at most 128 particle blocks here, so every wave keeps one or two block
//     prefix per lane in registers (max via DPP, one exp, one DPP scan) -- no LDS exchange, no barrier; the source block of
//     a wave's slots comes from two ballots, its prefix and scale from v_readlane;
//   * block statistics use the WAVE maximum for the exponentials and rescale by exp(max_wave - max_block) after a single
//     exchange: one barrier for max + scan instead of three;  => 2 barriers per launch (+ the table publish) instead of 7;
//   * the particles of a lane go through the search PHASE BY PHASE (all coarse-row loads, then all window loads, then all
//     ancestor gathers), each phase's loads in flight together: 3 dependent global rounds per wave-step instead of 3 per
//     particle;  every particle row has its own LDS window, so no wave-level fences serialise them;
//   * one generic search (`v2_search_generic`, a loop over the source blocks a wave's slots touch and over 128-entry
//     windows of their local scans) replaces the unrolled 16-ary per-lane fallbacks: a wave that straddles a block
//     boundary, a ragged wave or a wide window costs a second trip through the same cooperative code;
//   * the observation density is selected by a wave-uniform switch OUTSIDE the particle loop (one density body runs);
//   * table-driven log / exp / sincos (pf_math.h);
//   * ONE SEARCH PER WAVE.  A wave owns 64 * PPT CONSECUTIVE output slots (row r of a lane is slot wave * 64 PPT + 64 r +
//     lane, so every load / store instruction still covers 64 consecutive slots).  For the search the resampling
//     uniforms are transposed through the wave's LDS window so that a lane holds PPT consecutive slots: the wave's
//     targets are then one contiguous stratum range, and source block (two ballots), coarse row and window are found
//     ONCE per wave instead of once per particle row (an experiment build that merely reused row 0's search for the
//     other rows ran +16 %); the PPT bisections of a lane advance together, branch-free, over a window padded with
//     +inf; the ancestors go back through LDS to the row layout.  A wave whose range crosses ONE block boundary (63 of a
//     65 536-particle filter's 256 waves at four particles per lane) stays on the fast path: its window is the tail of
//     one block's local scan followed by the head of the next one's, and each particle bisects its own segment.
#pragma once

// analysis builds (-DBSSM_V2_MARKS): comment markers in the ISA listing at the phase boundaries (scripts/isa_phases.py)
#ifdef BSSM_V2_MARKS
#define V2_MARK(name) __asm__ volatile("; V2MARK " name ::: "memory")
#else
#define V2_MARK(name)
#endif
// analysis builds (-DBSSM_V2_STAMPS): shader-clock stamps of one wave at the phase boundaries (scripts/v2_phase_probe.py
// through bssm_plan_debug_timeline)
#ifdef BSSM_V2_STAMPS
#define V2_STAMP(k)                                                                                  \
  do {                                                                                               \
    __asm__ volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                   \
    if (ws.dbg && blockIdx.x == 3 && blockIdx.y == 0 && threadIdx.x == 64 && t >= 1)                   \
      ws.dbg[(size_t)t * 16 + (k)] = (long long)__builtin_readcyclecounter();                        \
    __asm__ volatile("" ::: "memory");                                                               \
  } while (0)
#else
#define V2_STAMP(k)
#endif

namespace bssm {

constexpr int V2_WIN = 128;  // entries of a source block's local scan staged per wave and particle of a lane

constexpr int v2_log2(int x) { return x <= 1 ? 0 : 1 + v2_log2(x >> 1); }
// slot (inside its block) of row r of thread tid: a wave owns 64 * PPT consecutive slots, a row 64 consecutive ones
template <int PPT> __device__ __forceinline__ int v2_li(int r, int tid) { return (tid >> 6) * (64 * PPT) + r * 64 + (tid & 63); }

// two consecutive doubles (16-byte aligned) of an array whose base is wave-uniform
typedef double v2_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2_d2 v2_ld32_d2(const double* uniform_base, uint32_t index) {
  typedef __attribute__((address_space(1))) const v2_d2 gD2;
  typedef __attribute__((address_space(1))) const char gC;
  return *(gD2*)((gC*)(uint64_t)uniform_base + (uint32_t)(index * 8u));
}
__device__ __forceinline__ v2_d2 v2_ld32_d2_nt(const double* uniform_base, uint32_t index) {
  typedef __attribute__((address_space(1))) const v2_d2 gD2;
  typedef __attribute__((address_space(1))) const char gC;
  return __builtin_nontemporal_load((gD2*)((gC*)(uint64_t)uniform_base + (uint32_t)(index * 8u)));
}
// Loads of data that ANOTHER workgroup of the same launch has written (the XCD-resident persistent kernel below): `sc1` loads
// -- relaxed agent-scope atomic loads -- are served by the L2 of this XCD and never by the CU's vector L1, which another CU's
// stores do not refresh (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility").  COH = false:
// the plain load of the launch-per-step kernels, where the kernel boundary is the synchronisation.
template <class T>
__device__ __forceinline__ T ld32_sc1(const T* uniform_base, uint32_t index) {
  typedef __attribute__((address_space(1))) T gT;
  typedef __attribute__((address_space(1))) char gC;
  return __hip_atomic_load((gT*)((gC*)(uint64_t)uniform_base + (uint32_t)(index * (uint32_t)sizeof(T))), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool COH, class T>
__device__ __forceinline__ T v2_ldx(const T* uniform_base, uint32_t index) {
  if constexpr (COH) return ld32_sc1(uniform_base, index);
  else return ld32(uniform_base, index);
}
// value of lane `src` (any lane, per-lane choice): ds_bpermute on the two halves (LDS crossbar, no LDS memory)
__device__ __forceinline__ double v2_lane_gather(double v, int src) {
  const int lo = __builtin_amdgcn_ds_bpermute(src << 2, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(src << 2, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
// Weights exchange (WX): blocks hand over their block-scaled WEIGHTS instead of the scan of them (the `lcdf` array then
// holds exp(w - max_b)), and a consumer rebuilds the scan values of the entries it stages: rows of 16 entries sit in
// rows of 16 lanes, so the prefix inside the row is four DPP row shifts, and the base of the row is the block's coarse
// entry of the row before (held by lane `row - 1` of `cv`).  One array is written per particle-step instead of two.
//   x: block-scaled weight of this lane's entry;  row: the entry's row inside its block;  cv: coarse row of that block
__device__ __forceinline__ double v2_entry_scan(double x, const double cv, const int row) {
  x += dpp_shr_zero<DPP_ROW_SHR1>(x);
  x += dpp_shr_zero<DPP_ROW_SHR2>(x);
  x += dpp_shr_zero<DPP_ROW_SHR4>(x);
  x += dpp_shr_zero<DPP_ROW_SHR8>(x);
  const double base = v2_lane_gather(cv, max(row - 1, 0));
  return row > 0 ? base + x : x;
}

// number of entries of sw[0, wlen) below vp (entries non-decreasing), wlen <= 128; exact when STEPS = 8, min(count, 127)
// when STEPS = 7
template <int STEPS>
__device__ __forceinline__ int v2_count_below(const double* sw, int wlen, double vp) {
  int lo = 0;
#pragma unroll
  for (int step = 1 << (STEPS - 1); step >= 1; step >>= 1) {
    const int idx = lo + step - 1;
    if (idx < wlen && sw[idx] < vp) lo += step;
  }
  return lo;
}

// Block prefixes of a filter as the lanes of a wave hold them: lane l has the inclusive, scaled prefix (and 1 / scale) of
// block l and of block 64 + l.  Blocks 128 ... 255 of a filter of more than 128 blocks (round 5: 256 blocks = 65 536 particles at
// one per lane, 262 144 at four) stay in LDS (`up`: prefixes [0, 128), 1 / scale [128, 256) of the combine's upper half) and are
// read where a wave's targets reach them -- no register of the kernels that never see such a filter
constexpr int V2_MAX_BLOCKS = 256;
struct V2Prefix {
  double Qa, Qb, ia, ib;
  int B;
  const double* up;
  __device__ __forceinline__ int count_below(const double v, const int lane) const {  // number of block prefixes below v
    int c = __builtin_popcountll(__builtin_amdgcn_ballot_w64(lane < B && Qa < v));
    if (B > 64) c += __builtin_popcountll(__builtin_amdgcn_ballot_w64(lane + 64 < B && Qb < v));  // wave-uniform
    if (B > 128) {
      c += __builtin_popcountll(__builtin_amdgcn_ballot_w64(lane + 128 < B && up[lane] < v));
      if (B > 192) c += __builtin_popcountll(__builtin_amdgcn_ballot_w64(lane + 192 < B && up[64 + lane] < v));
    }
    return c;
  }
  __device__ __forceinline__ double q(const int e) const { return e < 64 ? lane_broadcast(Qa, e) : (e < 128 ? lane_broadcast(Qb, e - 64) : up[e - 128]); }   // e wave-uniform
  __device__ __forceinline__ double is(const int e) const { return e < 64 ? lane_broadcast(ia, e) : (e < 128 ? lane_broadcast(ib, e - 64) : up[e]); }
};

// Ancestor of the wave's active slots (a prefix of its lanes), any shape: loops over the source blocks the slots fall
// into and over 128-entry windows of each block's local scan.  Every loop bound is wave-uniform.
//   Q / iscale: this lane's block prefix (inclusive, scaled) and 1 / scale of block `lane`
template <int PPT, bool WX = false, bool COH = false>
__device__ __forceinline__ int v2_search_generic(const double v, const bool act, const V2Prefix& px, const int B,
                                                 const int N, const double* __restrict__ coarse, const double* __restrict__ lcdf,
                                                 double* sw, const int lane) {
  constexpr int chunk = GRID_NT * PPT, crow = chunk / 16;
  const unsigned long long amask = __builtin_amdgcn_ballot_w64(act);
  if (amask == 0ull) return 0;
  const int l_last = 63 - __builtin_clzll(amask);
  const double v_lo = lane_broadcast(v, 0), v_hi = lane_broadcast(v, l_last);
  int c_lo = px.count_below(v_lo, lane);
  int c_hi = px.count_below(v_hi, lane);
  c_lo = min(c_lo, B - 1); c_hi = min(c_hi, B - 1);
  int sb = c_lo;
  for (int e = c_lo; e < c_hi; e++) sb += (px.q(e) < v) ? 1 : 0;
  int anc = 0;
  for (int e = c_lo; e <= c_hi; e++) {
    const bool mine = act && sb == e;
    const unsigned long long mm = __builtin_amdgcn_ballot_w64(mine);
    if (mm == 0ull) continue;
    const int f0 = __builtin_ctzll(mm), f1 = 63 - __builtin_clzll(mm);
    const double Pm1 = e > 0 ? px.q(e - 1) : 0.0;
    const double vp = (v - Pm1) * px.is(e);
    const double vp_lo = lane_broadcast(vp, f0), vp_hi = lane_broadcast(vp, f1);
    const int slen = min(chunk, N - e * chunk), ncr = (slen + 15) >> 4;
    const double cv = v2_ldx<COH>(coarse, (uint32_t)(e * crow + (lane < ncr ? lane : ncr - 1)));
    const int k_lo = min(__builtin_popcountll(__builtin_amdgcn_ballot_w64(lane < ncr && cv < vp_lo)), ncr - 1);
    const int k_hi = min(__builtin_popcountll(__builtin_amdgcn_ballot_w64(lane < ncr && cv < vp_hi)), ncr - 1);
    const int wend = min(16 * (k_hi + 1), slen);
    const double* __restrict__ src = lcdf + (size_t)e * chunk;
    int pos = 16 * k_lo;
    for (int w0 = 16 * k_lo; w0 < wend; w0 += V2_WIN) {
      const int wlen = min(V2_WIN, wend - w0);
      __builtin_amdgcn_wave_barrier();  // the previous window's reads are done (same wave, LDS is in order)
      double x0 = v2_ldx<COH>(src, (uint32_t)(w0 + (lane < wlen ? lane : wlen - 1))), x1 = v2_ldx<COH>(src, (uint32_t)(w0 + (lane + 64 < wlen ? lane + 64 : wlen - 1)));
      if constexpr (WX) {  // (w0 is a multiple of 16: rows of entries sit in rows of lanes)
        x0 = v2_entry_scan(x0, cv, (w0 + lane) >> 4);
        x1 = v2_entry_scan(x1, cv, (w0 + 64 + lane) >> 4);
      }
      sw[lane] = x0;
      sw[lane + 64] = x1;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      pos += v2_count_below<8>(sw, wlen, vp);
    }
    if (mine) anc = e * chunk + min(pos, slen - 1);
  }
  return anc;
}

// unnormalised log-density of the observation given the signals of a lane's PPT particles; the distribution is
// wave-uniform, so ONE body runs (src/model_ssm_ung.cpp:447-493; gaussian: src/model_ssm_ulg.cpp:582-586)
template <int PPT>
__device__ __forceinline__ void v2_log_weights(const cdbl* row, const bool psi, const double (&s)[PPT], double (&w)[PPT]) {
  const cdbl* ob = row + ROW_HDR;
  const int dist = (int)ob[OB_DIST];
  const double y = ob[OB_Y], u = ob[OB_U], aux = ob[OB_AUX], phi = ob[OB_PHI];
  switch (dist) {
    case DIST_SVM:
#pragma unroll
      for (int r = 0; r < PPT; r++) { w[r] = -0.5 * (s[r] + aux * tl_exp(-s[r])); __builtin_amdgcn_sched_barrier(0); }
      break;
    case DIST_POISSON:
#pragma unroll
      for (int r = 0; r < PPT; r++) { w[r] = y * s[r] - u * tl_exp(s[r]); __builtin_amdgcn_sched_barrier(0); }
      break;
    case DIST_BINOMIAL:
#pragma unroll
      for (int r = 0; r < PPT; r++) { w[r] = y * s[r] - u * log1p(tl_exp(s[r])); __builtin_amdgcn_sched_barrier(0); }
      break;
    case DIST_NEGBIN:
#pragma unroll
      for (int r = 0; r < PPT; r++) { w[r] = y * s[r] - aux * tl_log(phi + u * tl_exp(s[r])); __builtin_amdgcn_sched_barrier(0); }
      break;
    case DIST_GAMMA:
#pragma unroll
      for (int r = 0; r < PPT; r++) { w[r] = -phi * (s[r] + aux * tl_exp(-s[r])); __builtin_amdgcn_sched_barrier(0); }
      break;
    default:  // DIST_GAUSSIAN
#pragma unroll
      for (int r = 0; r < PPT; r++) { const double d = y - s[r]; w[r] = -0.5 * d * d * aux; }
      break;
  }
  if (psi) {  // psi-APF weight: minus the approximating Gaussian density and scales_t (src/model_ssm_ung.cpp:385-439, :550, :590)
    const double py = ob[OB_PSIY], pih = ob[OB_PSIIH], rs = row[RH_SCALE];
#pragma unroll
    for (int r = 0; r < PPT; r++) { const double q = (py - s[r]) * pih; w[r] = __builtin_fma(0.5 * q, q, w[r]) - rs; }
  }
}

template <int MP>
__device__ __forceinline__ double v2_signal(const cdbl* ob, const double (&x)[MP]) {
  double za = ob[OB_D];
#pragma unroll
  for (int c = 0; c < MP; c++) za = __builtin_fma(ob[SER_HDR + c], x[c], za);
  return za;
}

// the same for a row of p observation series (ssm_mng: src/model_ssm_mng.cpp:357-442): the log-weight is the sum over the
// observed series, each with its own distribution -- a wave-uniform loop, one density body per series
template <int MP, int PPT>
__device__ __forceinline__ void v2_log_weights_multi(const int p, const cdbl* row, const bool psi, const double (&x)[PPT][MP], double (&w)[PPT]) {
#pragma unroll
  for (int r = 0; r < PPT; r++) w[r] = 0.0;
  for (int j = 0; j < p; j++) {
    const cdbl* ob = row + ROW_HDR + j * (SER_HDR + MP);
    if (ob[OB_FLAG] == 0.0) continue;  // this series is missing at t (src/model_ssm_mng.cpp:414)
    double sj[PPT], dj[PPT];
#pragma unroll
    for (int r = 0; r < PPT; r++) sj[r] = v2_signal<MP>(ob, x[r]);
    const int dist = (int)ob[OB_DIST];
    const double y = ob[OB_Y], u = ob[OB_U], aux = ob[OB_AUX], phi = ob[OB_PHI];
    switch (dist) {
      case DIST_SVM:
#pragma unroll
        for (int r = 0; r < PPT; r++) { dj[r] = -0.5 * (sj[r] + aux * tl_exp(-sj[r])); __builtin_amdgcn_sched_barrier(0); }
        break;
      case DIST_POISSON:
#pragma unroll
        for (int r = 0; r < PPT; r++) { dj[r] = y * sj[r] - u * tl_exp(sj[r]); __builtin_amdgcn_sched_barrier(0); }
        break;
      case DIST_BINOMIAL:
#pragma unroll
        for (int r = 0; r < PPT; r++) { dj[r] = y * sj[r] - u * log1p(tl_exp(sj[r])); __builtin_amdgcn_sched_barrier(0); }
        break;
      case DIST_NEGBIN:
#pragma unroll
        for (int r = 0; r < PPT; r++) { dj[r] = y * sj[r] - aux * tl_log(phi + u * tl_exp(sj[r])); __builtin_amdgcn_sched_barrier(0); }
        break;
      case DIST_GAMMA:
#pragma unroll
        for (int r = 0; r < PPT; r++) { dj[r] = -phi * (sj[r] + aux * tl_exp(-sj[r])); __builtin_amdgcn_sched_barrier(0); }
        break;
      default:  // DIST_GAUSSIAN
#pragma unroll
        for (int r = 0; r < PPT; r++) { const double d = y - sj[r]; dj[r] = -0.5 * d * d * aux; }
        break;
    }
    if (psi) {
      const double py = ob[OB_PSIY], pih = ob[OB_PSIIH];
#pragma unroll
      for (int r = 0; r < PPT; r++) { const double q = (py - sj[r]) * pih; dj[r] = __builtin_fma(0.5 * q, q, dj[r]); }
    }
#pragma unroll
    for (int r = 0; r < PPT; r++) w[r] += dj[r];
  }
  if (psi) {
    const double rs = row[RH_SCALE];
#pragma unroll
    for (int r = 0; r < PPT; r++) w[r] -= rs;
  }
}

// One particle's log-weight over a row of p series, evaluated right behind its transition while its state is still in
// registers (the four-particles-per-lane multi-series variant: nothing of the state is kept for a density phase).  Same
// operations in the same order as v2_log_weights_multi, so the two give identical bits.
template <int MP>
__device__ __forceinline__ double v2_log_weight_multi_one(const int p, const cdbl* row, const bool psi, const double (&x)[MP]) {
  double w = 0.0;
  for (int j = 0; j < p; j++) {
    const cdbl* ob = row + ROW_HDR + j * (SER_HDR + MP);
    if (ob[OB_FLAG] == 0.0) continue;
    const double sj = v2_signal<MP>(ob, x);
    const int dist = (int)ob[OB_DIST];
    const double y = ob[OB_Y], u = ob[OB_U], aux = ob[OB_AUX], phi = ob[OB_PHI];
    double dj;
    switch (dist) {
      case DIST_SVM: dj = -0.5 * (sj + aux * tl_exp(-sj)); break;
      case DIST_POISSON: dj = y * sj - u * tl_exp(sj); break;
      case DIST_BINOMIAL: dj = y * sj - u * log1p(tl_exp(sj)); break;
      case DIST_NEGBIN: dj = y * sj - aux * tl_log(phi + u * tl_exp(sj)); break;
      case DIST_GAMMA: dj = -phi * (sj + aux * tl_exp(-sj)); break;
      default: { const double d = y - sj; dj = -0.5 * d * d * aux; } break;
    }
    if (psi) { const double q = (ob[OB_PSIY] - sj) * ob[OB_PSIIH]; dj = __builtin_fma(0.5 * q, q, dj); }
    w += dj;
  }
  if (psi) w -= row[RH_SCALE];
  return w;
}

// Block statistics of the new weights and their hand-over to the next launch: block-local inclusive scan of
// exp(w - max_b) (`lcdf`), row ends (`coarse`), (max_b, sum_b) (`partials`); optional block-scaled weights output.
// One workgroup barrier.  s_x: 4 + 4 * PPT doubles of LDS.
template <int PPT, bool WX>
__device__ __forceinline__ void v2_block_stats(const FilterBatch& b, const GridWorkspace& ws, const int f, const int pb, const int base,
                                               const int len, const int tn, const double (&w)[PPT], double* s_x) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int N = b.N, n = b.n, B = ws.B;
  constexpr int crow = GRID_NT * PPT / 16;
  double mx = w[0];
#pragma unroll
  for (int r = 1; r < PPT; r++) mx = fmax(mx, w[r]);
  const double Mw = wave_max(mx);
  const double Mws = Mw == -INFINITY ? 0.0 : Mw;  // a wave without mass: exp(-inf - 0) = 0 instead of exp(nan)
  const bool fw = WX && ws.fw != 0;  // factored weights output: the weights column of time tn is the exchange array
  double* __restrict__ lcdf_new = uniform_ptr(fw ? b.weights + ((size_t)f * (n + 1) + tn) * N + base : ws.lcdf + ((size_t)(tn & 1) * b.F + f) * N + base);
  double* __restrict__ coarse_new = uniform_ptr(ws.coarse + (((size_t)(tn & 1) * b.F + f) * B + pb) * crow);
  if constexpr (WX) {
    // Weights exchange: the next launch needs the block-scaled weights, the scan value at the end of every row of 16
    // entries (`coarse`) and the block's (max, sum) -- not the scan of every entry.  So: the sum of every row of 16 lanes
    // (four DPP steps), the 16 PPT row sums of the block through LDS, and ONE wave scans them (a single coalesced store
    // of the coarse row) while the others only scale and store their weights.  The output weights of time tn are written
    // by launch tn, once, in the global scale; the last time point has no launch of its own and its weights (ones:
    // nothing is observed behind the series) leave here.
    double e[PPT];
#pragma unroll
    for (int r = 0; r < PPT; r++) {
      e[r] = tl_exp(w[r] - Mws);  // inactive slots carry w = -inf: 0
      double rs = e[r];
      rs += dpp_shr_zero<DPP_ROW_SHR1>(rs);
      rs += dpp_shr_zero<DPP_ROW_SHR2>(rs);
      rs += dpp_shr_zero<DPP_ROW_SHR4>(rs);
      rs += dpp_shr_zero<DPP_ROW_SHR8>(rs);
      if ((lane & 15) == 15) s_x[4 + (wave * PPT + r) * 4 + (lane >> 4)] = rs;  // row (li >> 4) of the block
      __builtin_amdgcn_sched_barrier(0);
    }
    if (lane == 0) s_x[wave] = Mw;
    lds_barrier();
    const double m0 = s_x[0], m1 = s_x[1], m2 = s_x[2], m3 = s_x[3];
    const double Mb = fmax(fmax(m0, m1), fmax(m2, m3));
    const double Mbs = Mb == -INFINITY ? 0.0 : Mb;
    const double swv = tl_exp(Mws - Mbs) * (Mw == -INFINITY ? 0.0 : 1.0);  // exp(max_wave - max_block)
    double* __restrict__ wout = (b.weights && tn == n && !fw) ? uniform_ptr(b.weights + ((size_t)f * (n + 1) + tn) * N + base) : nullptr;
#pragma unroll
    for (int r = 0; r < PPT; r++) {
      const int li = v2_li<PPT>(r, tid);
      if (li < len) {
        st32(lcdf_new, li, e[r] * swv);
        if (wout) st32(wout, li, e[r] * swv);
      }
    }
    if (wave == 0) {  // rows of the block in lanes: row i belongs to wave i / (4 PPT)
      const int wq = lane / (4 * PPT);
      const double Mq = wq == 0 ? m0 : (wq == 1 ? m1 : (wq == 2 ? m2 : m3));
      const double scq = Mq == -INFINITY ? 0.0 : tl_exp(Mq - Mbs);
      const double rt = lane < crow ? s_x[4 + lane] : 0.0;
      const double c = wave_scan_incl(rt * scq);
      const int nrow = (len + 15) >> 4;
      if (lane < nrow) st32(coarse_new, lane, c);
      if (lane == 63) {
        double* __restrict__ part = ws.partials + ((size_t)(tn & 1) * b.F + f) * B * 2;
        part[2 * pb] = Mb;
        part[2 * pb + 1] = c;
      }
    }
  } else {
    double e[PPT], sc[PPT];
#pragma unroll
    for (int r = 0; r < PPT; r++) {
      e[r] = tl_exp(w[r] - Mws);  // inactive slots carry w = -inf: 0
      sc[r] = wave_scan_incl(e[r]);
      if (lane == 63) s_x[4 + wave * PPT + r] = sc[r];
      __builtin_amdgcn_sched_barrier(0);
    }
    if (lane == 0) s_x[wave] = Mw;
    lds_barrier();
    // every wave combines the 4 wave maxima and the 4 * PPT chunk totals (chunks in particle order: q = PPT wave + r)
    const double m0 = s_x[0], m1 = s_x[1], m2 = s_x[2], m3 = s_x[3];
    const double Mb = fmax(fmax(m0, m1), fmax(m2, m3));
    const double Mbs = Mb == -INFINITY ? 0.0 : Mb;
    const int qw = (lane / PPT) & 3;  // chunk `lane` belongs to wave lane / PPT
    const double Mq = qw == 0 ? m0 : (qw == 1 ? m1 : (qw == 2 ? m2 : m3));
    const double scq = tl_exp(Mq - Mbs);                       // lane q: exp(max of chunk q's wave - max_block)
    const double tq = lane < 4 * PPT ? s_x[4 + lane] : 0.0;
    const double incl = wave_scan_incl(tq * scq);              // lanes >= 4 PPT add zeros
    const double total = lane_broadcast(incl, 4 * PPT - 1);
    const double swv = lane_broadcast(scq, wave * PPT);
    double* __restrict__ wout = b.weights ? uniform_ptr(b.weights + ((size_t)f * (n + 1) + tn) * N + base) : nullptr;
#pragma unroll
    for (int r = 0; r < PPT; r++) {
      const int q = wave * PPT + r, li = v2_li<PPT>(r, tid);
      const double off = q == 0 ? 0.0 : lane_broadcast(incl, (q - 1) & 63);
      const double lc = __builtin_fma(sc[r], swv, off);
      if (li < len) {
        st32(lcdf_new, li, lc);
        if ((li & 15) == 15 || li == len - 1) st32(coarse_new, li >> 4, lc);  // last entry of each row of 16
        if (wout) st32(wout, li, e[r] * swv);  // block-scaled; rescaled to the global maximum by launch tn
      }
    }
    if (tid == 0) {
      double* __restrict__ part = ws.partials + ((size_t)(tn & 1) * b.F + f) * B * 2;
      part[2 * pb] = Mb;
      part[2 * pb + 1] = total;
    }
  }
}

// 16 live state doubles per lane (m = 4 at four particles per lane: the headline shape).  Five waves per SIMD since round 4: with
// the PACKED random stream the kernel fits 96 VGPRs without a spill (round 2 needed three tricks to get there and gained 0.6 %);
// same-box A/B 5.58 -> 5.74e10 particle-steps/s, +3.0 % (profiles/r04_ab_occ5.txt).  LDS: 5 x 27 KB per CU.
#ifndef BSSM_V2_OCC16
#define BSSM_V2_OCC16 5
#endif
#ifndef BSSM_V2_OCC_MS16
#define BSSM_V2_OCC_MS16 4   // MP = 8, two particles per lane: 128 VGPRs with 5 spilled beat 132 at three waves (C4, 64 filters: +4 ... +7 %)
#endif
#ifndef BSSM_V2_OCC_MS32
#define BSSM_V2_OCC_MS32 3   // MP = 8, four particles per lane (weights fused into the particle loop)
#endif
#ifndef BSSM_V2_OCC_MS8
#define BSSM_V2_OCC_MS8 4    // MP = 8, one particle per lane
#endif
constexpr int v2_occupancy(int mp, int ppt, bool ms = false) {
  if (mp >= 16) return ppt == 1 ? 3 : 2;  // sixteen-dimensional states: ancestor, new state, normals and the MFMA fragments of one particle: 90 ... 138 VGPRs at one per lane
  if (ms && mp * ppt == 16) return BSSM_V2_OCC_MS16;
  if (ms && mp * ppt == 8) return BSSM_V2_OCC_MS8;
  if (ms && ppt >= 4) return mp * ppt >= 32 ? BSSM_V2_OCC_MS32 : 4;
  const int live = mp * ppt * (ms && mp * ppt > 8 ? 2 : 1);  // the multi-series variants keep the new states until the density
  return live <= 4 ? 5 : (live <= 8 ? 4 : (live <= 16 ? BSSM_V2_OCC16 : (live <= 32 ? 3 : 2)));
}

__device__ __forceinline__ void v2_block_of(const GridWorkspace& ws, int& pb, int& f) {
  int lin = blockIdx.y * gridDim.x + blockIdx.x;
  const int total = gridDim.x * gridDim.y;
  if ((total & 7) == 0 && ws.xcd_remap) lin = (lin & 7) * (total >> 3) + (lin >> 3);  // XCD-contiguous (filter, block) ranges
  // the quotient is computed on the VALU (float reciprocal); without the readfirstlane the compiler keeps it in a VGPR and
  // every "uniform" table address derived from it becomes a per-lane address: vector loads instead of s_load
  pb = __builtin_amdgcn_readfirstlane(lin % ws.B);
  f = __builtin_amdgcn_readfirstlane(ws.f0 + lin / ws.B);
}

// t = -1: initial particles a1 + L u (src/model_ssm_ung.cpp:626; psi: alphahat_0 + V_0 u, :543) and their weights
template <int MP, int PPT, bool MS = false, int FN = FN_NONE, int PM = -1>
__global__ void __launch_bounds__(GRID_NT) pf_grid_init_v2(const FilterBatch b, const GridWorkspace ws) {
  constexpr int FID = FN > 0 ? FN : 0;
  const int psi_mode = PM >= 0 ? PM : b.psi;
  __shared__ double s_x[4 + 16 * PPT];
  int pb, f;
  v2_block_of(ws, pb, f);
  math_tab_init();
  const int tid = threadIdx.x, N = b.N, n = b.n, m = b.m;
  constexpr int chunk = GRID_NT * PPT;
  const int base = pb * chunk, len = min(chunk, N - base);
  const RngKey key{b.seed, b.stream0 + (uint32_t)f};
  const cdbl* ini = as_uniform_table(b.init + (size_t)f * b.init_fstride);
  const cdbl* row = as_uniform_table(b.obs + (size_t)f * b.obs_fstride);
  double* __restrict__ st_new = (b.alpha_hist ? b.alpha_hist + (size_t)f * (n + 1) * m * N : ws.state[0] + (size_t)f * m * N) + base;
  const bool observed = n > 0 && row[RH_FLAG] != 0.0;
  double s[PPT], w[PPT], xs[MS ? PPT : 1][MP];
#pragma unroll
  for (int r = 0; r < PPT; r++) {
    const int li = v2_li<PPT>(r, tid);
    double z[MP], x[MP];
    if constexpr (FN != FN_NONE) {  // functor models (ssm_nlg / ssm_sde): pf_functors.h
      const double* __restrict__ fpar = b.fpar + (size_t)f * b.fpar_fstride;
      if (!psi_mode) functor_initial<MP, FID>(fpar, b.init + (size_t)f * b.init_fstride, key, (uint32_t)(base + li), x);
      else { rng_normals<MP, false>(key, RNG_INIT, 0, (uint32_t)(base + li), z); initial_state<MP>(ini, z, x); }
      const double* __restrict__ amats = b.amats ? b.amats + (size_t)f * b.amats_fstride : nullptr;
      w[r] = n > 0 ? weight_dispatch<MP, FN, PM>(b, fpar, b.obs + (size_t)f * b.obs_fstride, amats, 0, x, x) : 0.0;
    } else {
      rng_normals<MP>(key, RNG_INIT, 0, (uint32_t)(base + li), z);
      initial_state<MP>(ini, z, x);
    }
    if (li < len) {
#pragma unroll
      for (int c = 0; c < MP; c++) if (c < m) st_new[(size_t)c * N + li] = x[c];
    }
    if constexpr (FN != FN_NONE) {
    } else if constexpr (MS) {
#pragma unroll
      for (int c = 0; c < MP; c++) xs[r][c] = x[c];
    } else {
      s[r] = v2_signal<MP>(row + ROW_HDR, x);
    }
  }
  if constexpr (FN == FN_NONE) {
  if (observed) {
    if constexpr (MS) v2_log_weights_multi<MP, PPT>(b.p, row, b.psi != 0, xs, w);
    else v2_log_weights<PPT>(row, b.psi != 0, s, w);
  } else {
#pragma unroll
    for (int r = 0; r < PPT; r++) w[r] = 0.0;
  }
  }
#pragma unroll
  for (int r = 0; r < PPT; r++) if (v2_li<PPT>(r, tid) >= len) w[r] = -INFINITY;
  if (ws.wx) v2_block_stats<PPT, true>(b, ws, f, pb, base, len, 0, w, s_x);
  else v2_block_stats<PPT, false>(b, ws, f, pb, base, len, 0, w, s_x);
}

// Ancestors of a wave's 64 * PPT consecutive slots whose targets fall into one source block (TWO = false) or into two
// neighbouring ones (TWO = true): the window staged in `sw` holds rows [k_lo, ...) of block e1's local scan and, with
// TWO, the first rows of block e2's behind them, then +inf up to 128 * PPT entries; every particle bisects its own block's
// segment.  The bisections of a lane's particles advance together (PPT independent LDS reads in flight per step) and
// are branch-free (the loads are pinned: the compiler otherwise turns the length test into a branch around each read).
//   vp[r]: the target in its source block's local scale;  in2 bit r: particle r's source block is e2
template <int PPT, bool TWO>
__device__ __forceinline__ void v2_window_ancestors(const double* sw, const double (&vp)[PPT], const unsigned in2, const int e1, const int e2,
                                                    const int w0, const int len1, const int len2, const int slen1, const int slen2,
                                                    int (&anc)[PPT]) {
  constexpr int chunk = GRID_NT * PPT, STEPS = 7 + v2_log2(PPT);
  int lo[PPT], off[PPT], seg[PPT];
#pragma unroll
  for (int r = 0; r < PPT; r++) {
    lo[r] = 0;
    if constexpr (TWO) { const bool second = (in2 >> r) & 1u; off[r] = second ? len1 : 0; seg[r] = second ? max(len2, 1) : len1; }
  }
#pragma unroll
  for (int step = 1 << (STEPS - 1); step >= 1; step >>= 1) {
    double x[PPT];
#pragma unroll
    for (int r = 0; r < PPT; r++) {
      if constexpr (TWO) x[r] = sw[off[r] + min(lo[r] + step - 1, seg[r] - 1)];
      else x[r] = sw[lo[r] + step - 1];  // (entries beyond the window are +inf)
    }
#pragma unroll
    for (int r = 0; r < PPT; r++) __asm__ volatile("" : "+v"(x[r]));
#pragma unroll
    for (int r = 0; r < PPT; r++) {
      bool below = x[r] < vp[r];
      if constexpr (TWO) below = below && lo[r] + step - 1 < seg[r];
      lo[r] += below ? step : 0;
    }
  }
#pragma unroll
  for (int r = 0; r < PPT; r++) {
    if constexpr (TWO) {
      const bool second = (in2 >> r) & 1u;
      anc[r] = second ? e2 * chunk + min(lo[r], slen2 - 1) : e1 * chunk + min(w0 + lo[r], slen1 - 1);
    } else {
      anc[r] = e1 * chunk + min(w0 + lo[r], slen1 - 1);
    }
  }
}

// t in [0, n): resample on the time-t weights, propagate to t+1, weigh at t+1 (ones when t+1 == n or y_{t+1} is missing)
// MS: a row of p >= 1 observation series (ssm_mng, and every model with an 8-dimensional state); with MP = 8 and one
// particle per lane the transition runs on the matrix cores (propagate_mfma8), whose per-wave operand tile doubles as the
// wave's search window
// WX: weights exchange (see v2_entry_scan): with the weights output on, the owner's read-modify-write becomes "read my
// exchanged weights, write the output once", and the `lcdf` write (8 of the 109 bytes a particle-step moved) disappears
// EXACT: every block is complete (N a multiple of 256 * PPT) and the state has no padding components (m == MP): the tail
// guards of every load and store fold away (-18 % SALU instructions; C2-m4 +4.4 %)
// The body of one step for workgroup (filter f, particle block pb).  COH: the data other workgroups wrote (partials, coarse
// rows, windows, ancestor states) is read with sc1 loads -- the persistent kernel, where they were written in the SAME launch.
// load_tables: the first step of a launch publishes the math tables through barrier (1).  Returns false when the filter has
// degenerated at this step (every workgroup of the filter sees the same S and returns the same).
// FN / PM: a functor model (ssm_nlg / ssm_sde, pf_functors.h) and its sampler (0 bootstrap, 1 psi-APF, 2 EKF particle filter) -- round 5:
// filters of more than 8192 particles (the row family's limit) no longer fall back to the round-1 step kernel
template <int MP, int KP, int PPT, bool MS, bool WX, bool EXACT, bool COH, int FN = FN_NONE, int PM = -1>
__device__ __forceinline__ bool v2_step_body(const FilterBatch& b, const GridWorkspace& ws, const int t, const int pb, const int f, const bool load_tables) {
  constexpr int FID = FN > 0 ? FN : 0;
  constexpr bool kFn = FN != FN_NONE;
  const int psi_mode = PM >= 0 ? PM : b.psi;
  #ifndef BSSM_V2_MFMA_PPT
#define BSSM_V2_MFMA_PPT 4
#endif
  constexpr bool kMfma = MS && MP == 8 && PPT <= BSSM_V2_MFMA_PPT;
  // sixteen-dimensional states (m = 9 ... 16, round 5): the transition on the matrix cores through a 4 x 64 tile per wave
  // (propagate_mfma16_q, pf_filter_impl.h) that doubles as the search window
  constexpr bool kMfma16 = MS && MP == 16 && FN == FN_NONE;
  constexpr int kWinWave = kMfma ? 8 * 64 : ((kMfma16 && PPT * V2_WIN < 256) ? 256 : PPT * V2_WIN);  // doubles of LDS per wave
  __shared__ double s_win[4 * kWinWave];
  static_assert(!kMfma || PPT * V2_WIN <= 8 * 64, "the MFMA operand tile doubles as the search window");
  __shared__ double s_x[4 + 16 * PPT];
  __shared__ double s_comb[3 * 128 + 1];  // combine of wave 0: block prefixes, 1 / scale, scale (128 blocks each), global maximum
  __shared__ double s_comb_up[3 * 128];   // the same three arrays for blocks 128 ... 255 (filters of more than 128 blocks)
  // T_t, R_t, C_t of this step.  Read from LDS (broadcast ds_read, same address in every lane) and not through scalar
  // loads: 36 doubles are 72 SGPRs, which with the kernel arguments and the observation row overflowed the scalar file --
  // 58 SGPRs spilled to VGPR lanes and came back through ~250 v_readlane / v_writelane per wave-step (10 % of the VALU work)
  constexpr int NMAT = MP * MP + MP * KP + MP;
  __shared__ double s_mat[NMAT];
  typedef double d2 __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  MathTabRegs mtab{};
  if (load_tables) mtab = math_tab_load_begin256();
  const int B = ws.B, N = b.N, n = b.n, m = EXACT ? MP : b.m, tn = t + 1;
  constexpr int chunk = GRID_NT * PPT, crow = chunk / 16;
  const int base = pb * chunk, len = EXACT ? chunk : min(chunk, N - base);
  const RngKey key{b.seed, b.stream0 + (uint32_t)f};

  V2_MARK("loads"); V2_STAMP(0);
  // ---- exchange loads: the (max, sum) partials of the time-t weights, one block per lane -- combined by wave 0 only
  // (global maximum, scaled block sums, their inclusive prefix, total; the polynomial exp needs no table) while every
  // wave draws the step blocks of its particles (integer work, independent of the exchange); barrier (1) then
  // publishes both the math tables and the combine
#ifdef BSSM_V2_NO_RMW
  double* __restrict__ wt = nullptr;
#else
  const bool fw = WX && ws.fw != 0;  // factored weights output (GridWorkspace::fw): no rescaling pass over the time-t weights
  double* __restrict__ wt = (b.weights && !fw) ? uniform_ptr(b.weights + ((size_t)f * (n + 1) + t) * N + base) : nullptr;
#endif
  // the read-modify-write that rescales the stored time-t weights from the block maximum to the global one; LATE: behind
  // the transition, where few registers are live (BSSM_V2_RMW_LATE, the five-waves-per-SIMD builds), otherwise loaded
  // here and stored right behind the barrier
#ifndef BSSM_V2_RMW_LATE
#define BSSM_V2_RMW_LATE 0
#endif
  // BSSM_V2_NT: non-temporal (`nt`, streaming) cache policy -- stores of the outputs the device never reads again: bit 0
  // ancestor indices, bit 1 output weights; bit 3: the ancestor gathers as non-temporal loads (each line is wanted by
  // one or two neighbouring waves, then never again; only where kNtGather below says).  Default 11: C2-m4 +2.9 %
  // (profiles/r02c_ab_nontemporal.txt).
  // Measured and not kept: bit 2 the new states (read again by the next launch's gathers: -1.3 %), bit 4 the owner's read of
  // its exchanged weights (-7 %: the consumers' windows want those lines), bit 5 the window loads (-1.3 %)
#ifndef BSSM_V2_NT
#define BSSM_V2_NT 11
#endif
  const double* __restrict__ wsrc = WX ? uniform_ptr(ws.lcdf + ((size_t)(t & 1) * b.F + f) * N + base) : wt;
  auto rmw_load = [&](double (&wo)[PPT]) {
#pragma unroll
    for (int r = 0; r < PPT; r++) { const int li = v2_li<PPT>(r, tid); wo[r] = li < len ? (((BSSM_V2_NT) & 16) ? ld32_nt(wsrc, li) : ld32(wsrc, li)) : 0.0; }
  };
  auto rmw_store = [&](const double (&wo)[PPT]) {
    const double myscale = pb < 128 ? s_comb[256 + pb] : s_comb_up[128 + pb];
#pragma unroll
    for (int r = 0; r < PPT; r++) {
      const int li = v2_li<PPT>(r, tid);
      if (li < len) { if constexpr ((BSSM_V2_NT) & 2) st32_nt(wt, li, wo[r] * myscale); else st32(wt, li, wo[r] * myscale); }
    }
  };
  double wt_old[PPT];
  if (!BSSM_V2_RMW_LATE && wt) rmw_load(wt_old);
  // BSSM_V2_EARLY_Z (rows of several series, PACKED step draw; round 4): the transition normals are drawn UNDER the round trips
  // of the exchange and the search instead of behind the gathers -- they depend on nothing but (seed, t, slot), and the
  // Box-Muller code no longer overlaps the live ranges of the gathered states (C4, two particles per lane: 127 VGPRs with two
  // spilt -> 121 with none).  1: the pairs of the step block while the coarse rows travel; 2: the second block there too;
  // 3 (default): the math tables are published by a barrier of their own, so that waves 1-3 draw the first block while
  // wave 0 combines the partials; wave 0's first block and every wave's second block while the coarse rows travel.
  // Same-box A/B (profiles/r04_c4_earlyz_ab.txt): C4 at 64 / 32 filters in flight +3.0 % / +6.5 %
#ifndef BSSM_V2_EARLY_Z
#define BSSM_V2_EARLY_Z 3
#endif
  constexpr int kEarlyZ = (MS && KP > 2 && KP <= 8 && PPT <= 2) ? (BSSM_V2_EARLY_Z) : 0;
  double zq[kEarlyZ ? PPT : 1][kEarlyZ ? KP : 1];
  auto early_block = [&](const int r, const int j0, const u32x4 w) {
    double z0, z1;
    box_muller_packed(w.x, w.y, z0, z1);
    zq[r][j0] = z0;
    if (j0 + 1 < KP) zq[r][j0 + 1] = z1;
    if (j0 + 2 < KP) {
      box_muller_packed(w.z, w.w, z0, z1);
      zq[r][j0 + 2] = z0;
      if (j0 + 3 < KP) zq[r][j0 + 3] = z1;
    }
#pragma unroll
    for (int j = j0; j < j0 + 4 && j < KP; j++) __asm__ volatile("" : "+v"(zq[r][j]));
  };
  auto early_second = [&]() {
    if constexpr (KP > 4) {
#pragma unroll
      for (int r = 0; r < PPT; r++)
        early_block(r, 4, philox4x32_10((uint32_t)(base + v2_li<PPT>(r, tid)), (uint32_t)t, 1u, RNG_TRANSITION, key.seed, key.stream));
    }
  };
  V2_MARK("combine"); V2_STAMP(1);
  double pmax = -INFINITY, psum = 0.0, pmaxb = -INFINITY, psumb = 0.0, pmaxc = -INFINITY, psumc = 0.0, pmaxd = -INFINITY, psumd = 0.0;
  if (wave == 0) {
    const double* __restrict__ part = ws.partials + ((size_t)(t & 1) * b.F + f) * B * 2;
    if (B > 128) {  // wave-uniform
      if (lane + 128 < B) { pmaxc = v2_ldx<COH>(part, (uint32_t)(2 * (lane + 128))); psumc = v2_ldx<COH>(part, (uint32_t)(2 * (lane + 128) + 1)); }
      if (lane + 192 < B) { pmaxd = v2_ldx<COH>(part, (uint32_t)(2 * (lane + 192))); psumd = v2_ldx<COH>(part, (uint32_t)(2 * (lane + 192) + 1)); }
    }
    if constexpr (COH) {
      if (lane < B) { pmax = ld32_sc1(part, (uint32_t)(2 * lane)); psum = ld32_sc1(part, (uint32_t)(2 * lane + 1)); }
      if (lane + 64 < B) { pmaxb = ld32_sc1(part, (uint32_t)(2 * (lane + 64))); psumb = ld32_sc1(part, (uint32_t)(2 * (lane + 64) + 1)); }
    } else {
      if (lane < B) { const d2 pp = *(const d2*)(part + 2 * lane); pmax = pp.x; psum = pp.y; }
      if (lane + 64 < B) { const d2 pp = *(const d2*)(part + 2 * (lane + 64)); pmaxb = pp.x; psumb = pp.y; }
    }
  }
  if constexpr (kEarlyZ == 3) {
    if (load_tables) { math_tab_load_end256(mtab); lds_barrier(); }
  }
  if (wave == 0) {
    const double M0 = wave_max(B > 128 ? fmax(fmax(pmax, pmaxb), fmax(pmaxc, pmaxd)) : fmax(pmax, pmaxb));
    const double scale0 = fast_exp(pmax - M0);  // lanes >= B: 0; a poisoned (NaN) partial makes S NaN below
    const double Q0 = wave_scan_incl(psum * scale0);
    s_comb[lane] = Q0;
    s_comb[128 + lane] = 1.0 / scale0;
    s_comb[256 + lane] = scale0;
    double Q1 = lane_broadcast(Q0, 63);
    if (B > 64) {  // wave-uniform: blocks 64 ... 127
      const double scale1 = fast_exp(pmaxb - M0);
      Q1 += wave_scan_incl(psumb * scale1);
      s_comb[128 + 64 + lane] = 1.0 / scale1;
      s_comb[256 + 64 + lane] = scale1;
    }
    s_comb[64 + lane] = Q1;  // (B <= 64: the total in every lane)
    if (B > 128) {  // blocks 128 ... 255
      const double scale2 = fast_exp(pmaxc - M0);
      const double Q2 = lane_broadcast(Q1, 63) + wave_scan_incl(psumc * scale2);
      s_comb_up[lane] = Q2;
      s_comb_up[128 + lane] = 1.0 / scale2;
      s_comb_up[256 + lane] = scale2;
      double Q3 = lane_broadcast(Q2, 63);
      if (B > 192) {
        const double scale3 = fast_exp(pmaxd - M0);
        Q3 += wave_scan_incl(psumd * scale3);
        s_comb_up[128 + 64 + lane] = 1.0 / scale3;
        s_comb_up[256 + 64 + lane] = scale3;
      }
      s_comb_up[64 + lane] = Q3;
    }
    if (lane == 0) s_comb[384] = M0;
  }
  if (!kFn || psi_mode == 1) {  // (functor models: only the psi-APF proposal is linear)
#pragma unroll
    for (int i = tid; i < NMAT; i += GRID_NT) s_mat[i] = (b.mats + (size_t)f * b.mats_fstride + (size_t)t * b.mats_tstride)[i];
  }
  u32x4 blk[PPT];
#pragma unroll
  for (int r = 0; r < PPT; r++) blk[r] = rng_step_block(key, (uint32_t)t, (uint32_t)(base + v2_li<PPT>(r, tid)));
  if constexpr (kEarlyZ == 3) {
    if (wave != 0) {
#pragma unroll
      for (int r = 0; r < PPT; r++) early_block(r, 0, blk[r]);
    }
  }
  if (load_tables && kEarlyZ != 3) math_tab_load_end256(mtab);
  lds_barrier();  // (1)
  const V2Prefix px{s_comb[lane], s_comb[64 + lane], s_comb[128 + lane], s_comb[128 + 64 + lane], B, s_comb_up};
  const double S = B > 128 ? s_comb_up[127] : s_comb[127], M = s_comb[384];
  if (fw && pb == 0 && tid < B) ws.wscale[((size_t)f * (n + 1) + t) * B + tid] = tid < 128 ? s_comb[256 + tid] : s_comb_up[128 + tid];  // exp(max_block - max_t)
  V2_MARK("rmw"); V2_STAMP(2);
  // stored weights of time t are exp(w - max_block): rescale to exp(w - M) (also on the degenerate step, :676-682)
  if (!BSSM_V2_RMW_LATE && wt) rmw_store(wt_old);
  if (!(S > 0.0)) {
    if (BSSM_V2_RMW_LATE && wt) { rmw_load(wt_old); rmw_store(wt_old); }
    // degenerate weights (src/model_ssm_ung.cpp:678-682): stop and poison this block's slot of the next parity, so every
    // later launch sees a non-finite S too (no flag load on healthy steps); later output slots stay zero like the reference's
    if (tid == 0) {
      double* __restrict__ pn = ws.partials + ((size_t)(tn & 1) * b.F + f) * B * 2;
      pn[2 * pb] = __builtin_nan("");
      pn[2 * pb + 1] = __builtin_nan("");
      if (pb == 0) atomicCAS(&b.dead_at[f], 0, t + 1);
    }
    return false;
  }
  if (pb == B - 1 && tid == GRID_NT - 1) ws.llinc[(size_t)f * (n + 1) + t] = M + tl_log(S / (double)N);  // :683
  const double c_scale = S / (double)N;

  const double* __restrict__ lcdf = uniform_ptr(fw ? b.weights + ((size_t)f * (n + 1) + t) * N : ws.lcdf + ((size_t)(t & 1) * b.F + f) * N);
  const double* __restrict__ coarse = uniform_ptr(ws.coarse + ((size_t)(t & 1) * b.F + f) * B * crow);
  const double* __restrict__ st_old = uniform_ptr(b.alpha_hist ? b.alpha_hist + ((size_t)f * (n + 1) + t) * m * N : ws.state[t & 1] + (size_t)f * m * N);
  double* __restrict__ st_new = uniform_ptr((b.alpha_hist ? b.alpha_hist + ((size_t)f * (n + 1) + tn) * m * N : ws.state[tn & 1] + (size_t)f * m * N) + base);
  uint32_t* __restrict__ idx_out = b.indices ? uniform_ptr(b.indices + ((size_t)f * n + t) * N + base) : nullptr;
  double* sw_base = s_win + wave * kWinWave;

#ifndef BSSM_V2_PRIO
#define BSSM_V2_PRIO 0
#endif
  // BSSM_V2_PRIO: wave priority (s_setprio) raised from here until the ancestor gathers have been requested (1), or only
  // around the search's global rounds and the gathers (2): measurement option, profiles/r02d_ab_setprio.txt
  if constexpr ((BSSM_V2_PRIO) == 1 || (BSSM_V2_PRIO) == 3) __builtin_amdgcn_s_setprio(2);
  V2_MARK("phaseA"); V2_STAMP(3);
  // ---- phase A (once per wave): the resampling uniforms go through the wave's LDS window so that a lane holds PPT
  // CONSECUTIVE slots (wave * 64 PPT + PPT lane + q); targets; the source blocks of the wave's first and last target (two
  // ballots over the block prefixes); coarse rows of those blocks in flight
  constexpr int WIN = PPT * V2_WIN;  // window entries per wave
  uint32_t* sw_u32 = (uint32_t*)sw_base;
  // a PACKED step draw (more than two noise terms, pf_device.h) leaves no spare word in the step blocks: the uniforms of four
  // consecutive slots share one RNG_RESAMPLE block, which the lane that holds those slots for the search draws directly --
  // no transposition through LDS
  constexpr bool kSharedU = !kFn && KP > 2;  // (the functor filters keep WIDE blocks whatever their count: pf_device.h)
  if constexpr (PPT > 1 && !kSharedU) {
#pragma unroll
    for (int r = 0; r < PPT; r++) sw_u32[r * 64 + lane] = blk[r].w;
    lds_wave_sync();
  }
  double vp[PPT];
  unsigned in2 = 0;  // bit q: particle q draws from the second of the wave's two source blocks
  bool slow = !EXACT && (wave + 1) * 64 * PPT > len;  // wave-uniform: a ragged wave goes through the generic search
  int e1, e2, slen1, slen2, ncr1, ncr2;
  double cv1, cv2;
  {
    double v[PPT];
    uint32_t uw[PPT];
    if constexpr (kSharedU) {
      const uint32_t s0 = (uint32_t)(base + wave * (64 * PPT) + PPT * lane);  // first of this lane's PPT consecutive slots
      if constexpr (PPT >= 4) {
#pragma unroll
        for (int q = 0; q < PPT; q += 4) {
          const u32x4 ub = rng_resample_block(key, (uint32_t)t, (s0 + q) >> 2);
          uw[q] = ub.x; uw[q + 1] = ub.y; uw[q + 2] = ub.z; uw[q + 3] = ub.w;
        }
      } else {
        const u32x4 ub = rng_resample_block(key, (uint32_t)t, s0 >> 2);
#pragma unroll
        for (int q = 0; q < PPT; q++) uw[q] = u32x4_word(ub, (s0 + q) & 3u);
      }
    } else {
#pragma unroll
      for (int q = 0; q < PPT; q++) uw[q] = PPT > 1 ? sw_u32[PPT * lane + q] : blk[0].w;
    }
#pragma unroll
    for (int q = 0; q < PPT; q++) v[q] = ((double)uw[q] * 0x1.0p-32 + (double)(base + wave * (64 * PPT) + PPT * lane + q)) * c_scale;
    const double v_lo = lane_broadcast(v[0], 0), v_hi = lane_broadcast(v[PPT - 1], 63);
    const int c_lo = px.count_below(v_lo, lane);
    const int c_hi = px.count_below(v_hi, lane);
    e1 = min(c_lo, B - 1); e2 = min(c_hi, B - 1);
    if (e2 > e1 + 1) { slow = true; e2 = e1; }
    const double Pm1 = e1 > 0 ? px.q(e1 - 1) : 0.0;
    const double is1 = px.is(e1);
    slen1 = min(chunk, N - e1 * chunk); ncr1 = (slen1 + 15) >> 4;
    slen2 = min(chunk, N - e2 * chunk); ncr2 = (slen2 + 15) >> 4;
    auto coarse_row = [&](int e, int ncr) -> double { return v2_ldx<COH>(coarse + (size_t)e * crow, (uint32_t)(lane < ncr ? lane : ncr - 1)); };
    cv1 = coarse_row(e1, ncr1);
    if (e2 != e1) {
      const double P1 = px.q(e1), is2 = px.is(e2);
      cv2 = coarse_row(e2, ncr2);
#pragma unroll
      for (int q = 0; q < PPT; q++) {
        const bool second = P1 < v[q];  // the rule of the generic search: the source block is the number of prefixes below v
        in2 |= second ? 1u << q : 0u;
        vp[q] = second ? (v[q] - P1) * is2 : (v[q] - Pm1) * is1;
      }
    } else {
      cv2 = cv1;
#pragma unroll
      for (int q = 0; q < PPT; q++) vp[q] = (v[q] - Pm1) * is1;
    }
  }
  if constexpr (kEarlyZ >= 1) {
    __builtin_amdgcn_sched_barrier(0);
    if (kEarlyZ != 3 || wave == 0) {
#pragma unroll
      for (int r = 0; r < PPT; r++) early_block(r, 0, blk[r]);
    }
    if constexpr (kEarlyZ >= 2) early_second();
    __builtin_amdgcn_sched_barrier(0);
  }
  V2_MARK("phaseB"); V2_STAMP(4);
  // ---- phase B: the rows of the source blocks the wave's targets fall into; their entries (<= 128 per particle of a lane)
  // in flight -- only as many loads as the window has entries
  int w0, len1, len2;
  {
    double wr[2 * PPT];
    const double vp_lo = lane_broadcast(vp[0], 0), vp_hi = lane_broadcast(vp[PPT - 1], 63);
    const int k_lo = min(__builtin_popcountll(__builtin_amdgcn_ballot_w64(lane < ncr1 && cv1 < vp_lo)), ncr1 - 1);
    const int k_hi = min(__builtin_popcountll(__builtin_amdgcn_ballot_w64(lane < ncr2 && cv2 < vp_hi)), ncr2 - 1);
    w0 = 16 * k_lo;
    if (e2 != e1) { len1 = slen1 - w0; len2 = min(16 * (k_hi + 1), slen2); }
    else { len1 = max(1, min(16 * (k_hi + 1), slen1) - w0); len2 = 0; }
    int total = len1 + len2;
    if (total > WIN) { slow = true; total = 1; len1 = 1; len2 = 0; }
    const uint32_t off1 = (uint32_t)(e1 * chunk + w0), off2 = (uint32_t)(e2 * chunk - len1);
    if constexpr (WX && EXACT && PPT >= 2) {
      // Weights exchange with complete blocks: the lane stages E = 2 PPT CONSECUTIVE entries (16-byte loads; the window and
      // both of its segments are whole rows of 16 entries, so a lane's run is all inside one segment and all valid or all
      // beyond the window).  Their scan is serial inside the lane, a short DPP scan over the 16 / E lanes of a row, plus the
      // coarse entry of the row before: ~30 VALU instructions per wave-step instead of ~100 for the row-per-16-lanes form.
      constexpr int E = 2 * PPT, G = 16 / E;
      const int i0 = E * lane, g = lane & (G - 1);
      const bool valid = i0 < total;
      const bool second = e2 != e1 && i0 >= len1;
      double x[E];
#pragma unroll
      for (int j = 0; j < E; j++) x[j] = 0.0;
      if (valid) {
        const uint32_t src0 = (second ? off2 : off1) + (uint32_t)i0;
#pragma unroll
        for (int k = 0; k < E; k += 2) {
          if constexpr (COH) { x[k] = ld32_sc1(lcdf, src0 + k); x[k + 1] = ld32_sc1(lcdf, src0 + k + 1); }
          else { const v2_d2 d = ((BSSM_V2_NT) & 32) ? v2_ld32_d2_nt(lcdf, src0 + k) : v2_ld32_d2(lcdf, src0 + k); x[k] = d.x; x[k + 1] = d.y; }
        }
      }
#pragma unroll
      for (int j = 1; j < E; j++) x[j] += x[j - 1];
      double T = x[E - 1];
      if constexpr (G >= 2) { const double u = dpp_shr_zero<DPP_ROW_SHR1>(T); T += g >= 1 ? u : 0.0; }
      if constexpr (G >= 4) { const double u = dpp_shr_zero<DPP_ROW_SHR2>(T); T += g >= 2 ? u : 0.0; }
      if constexpr (G >= 8) { const double u = dpp_shr_zero<DPP_ROW_SHR4>(T); T += g >= 4 ? u : 0.0; }
      const double before = dpp_shr_zero<DPP_ROW_SHR1>(T);  // the lanes of my row in front of me
      const int row = second ? (i0 - len1) >> 4 : (w0 + i0) >> 4;
      double rbase = v2_lane_gather(cv1, max(row - 1, 0));
      if (e2 != e1) { const double b2 = v2_lane_gather(cv2, max(row - 1, 0)); rbase = second ? b2 : rbase; }
      double offs = (row > 0 ? rbase : 0.0) + (g >= 1 ? before : 0.0);
      offs = valid ? offs : __builtin_huge_val();
      V2_MARK("phaseC"); V2_STAMP(5);
      if constexpr (PPT > 1) lds_wave_sync();  // the transposed uniforms have been read
#pragma unroll
      for (int k = 0; k < E; k += 2) *(v2_d2*)(sw_base + i0 + k) = v2_d2{x[k] + offs, x[k + 1] + offs};
    } else {
#pragma unroll
    for (int q = 0; q < 2 * PPT; q++) {
      wr[q] = __builtin_huge_val();
      if (q * 64 < total) {  // wave-uniform
        const int i = min(q * 64 + lane, total - 1);
        wr[q] = v2_ldx<COH>(lcdf, (i < len1 ? off1 : off2) + (uint32_t)i);
        if constexpr (WX) {  // the exchanged values are weights: their scan inside the row + the coarse entry of the row before
          // (w0 and len1 are multiples of 16 wherever a second segment exists: rows of entries sit in rows of lanes)
          const int i0 = q * 64 + lane;
          double c;
          if (e2 != e1) {
            const bool second = i0 >= len1;
            const int row = second ? (i0 - len1) >> 4 : (w0 + i0) >> 4;
            double x = wr[q];
            x += dpp_shr_zero<DPP_ROW_SHR1>(x);
            x += dpp_shr_zero<DPP_ROW_SHR2>(x);
            x += dpp_shr_zero<DPP_ROW_SHR4>(x);
            x += dpp_shr_zero<DPP_ROW_SHR8>(x);
            const double b1 = v2_lane_gather(cv1, max(row - 1, 0)), b2 = v2_lane_gather(cv2, max(row - 1, 0));
            c = row > 0 ? (second ? b2 : b1) + x : x;
          } else {
            c = v2_entry_scan(wr[q], cv1, (w0 + i0) >> 4);
          }
          wr[q] = i0 < total ? c : __builtin_huge_val();
        }
      }
    }
    V2_MARK("phaseC"); V2_STAMP(5);
    // ---- phase C: window to LDS (+inf behind its entries), bisection, ancestors back to the row layout
    if constexpr (PPT > 1) lds_wave_sync();  // the transposed uniforms have been read
#pragma unroll
    for (int q = 0; q < 2 * PPT; q++) sw_base[q * 64 + lane] = wr[q];
    }
  }
  lds_wave_sync();
  int anc[PPT];
  if (e2 != e1) v2_window_ancestors<PPT, true>(sw_base, vp, in2, e1, e2, w0, len1, len2, slen1, slen2, anc);
  else v2_window_ancestors<PPT, false>(sw_base, vp, in2, e1, e2, w0, len1, len2, slen1, slen2, anc);
  if constexpr (PPT > 1) {  // anc[q] is the ancestor of slot PPT lane + q: back through LDS to rows of 64 consecutive slots
    lds_wave_sync();
#pragma unroll
    for (int q = 0; q < PPT; q++) sw_u32[PPT * lane + q] = (uint32_t)anc[q];
    lds_wave_sync();
#pragma unroll
    for (int r = 0; r < PPT; r++) anc[r] = (int)sw_u32[r * 64 + lane];
  }
  V2_MARK("slow"); V2_STAMP(6);
  if (slow) {  // wave-uniform, rare: more than one block boundary inside the wave's slots, a ragged wave, or a window beyond WIN
               // entries: the cooperative search row by row (64 consecutive slots each)
    lds_wave_sync();
#pragma unroll 1
    for (int r = 0; r < PPT; r++) {
      uint32_t ur = blk[0].w;
#pragma unroll
      for (int q = 1; q < PPT; q++) ur = (r == q) ? blk[q].w : ur;  // r is wave-uniform: scalar compare + select
      const int li = v2_li<PPT>(r, tid);
      if constexpr (kSharedU) ur = rng_resample_word<KP>(key, (uint32_t)t, (uint32_t)(base + li), blk[0]);
      const double vr = ((double)ur * 0x1.0p-32 + (double)(base + li)) * c_scale;  // the target, as in phase A
      const int ar = v2_search_generic<PPT, WX, COH>(vr, li < len, px, B, N, coarse, lcdf, sw_base, lane);
#pragma unroll
      for (int q = 0; q < PPT; q++) anc[q] = (r == q) ? ar : anc[q];
    }
  }
  if constexpr ((BSSM_V2_PRIO) == 2) __builtin_amdgcn_s_setprio(2);
  V2_MARK("phaseD"); V2_STAMP(7);
  // ---- phase D: ancestor indices out, ancestor states in flight -- at most ROLL rows of them at a time: row r + ROLL is
  // requested as soon as row r has been propagated (BSSM_V2_ROLL = PPT: all rows at once, as before)
#ifndef BSSM_V2_ROLL
#define BSSM_V2_ROLL PPT
#endif
#ifndef BSSM_V2_PIN_BLK
#define BSSM_V2_PIN_BLK 0
#endif
  // kFuse (rows of several series at four particles per lane): a particle's log-weight is evaluated right behind its
  // transition, its state never outlives the iteration, and the ancestor rows are requested two at a time
  constexpr bool kFuse = (MS && PPT >= 4) || kFn;  // (functor weights need the ancestor next to the new state: pf_functors.h)
#ifndef BSSM_V2_FUSE_ROLL
#define BSSM_V2_FUSE_ROLL 2
#endif
  constexpr int ROLL = kFuse ? (BSSM_V2_FUSE_ROLL) : ((BSSM_V2_ROLL) < PPT ? (BSSM_V2_ROLL) : PPT);
  double a[ROLL][MP];
#pragma unroll
  for (int r = 0; r < PPT; r++) {
    const int li = v2_li<PPT>(r, tid);
    if (li >= len) anc[r] = 0;
    if (idx_out && li < len) { if constexpr ((BSSM_V2_NT) & 1) st32_nt(idx_out, li, (uint32_t)anc[r]); else st32(idx_out, li, (uint32_t)anc[r]); }
  }
  // (measured per shape, profiles/r02c_ab_nontemporal.txt: +2.3 % for the full-output m = 4 batch whose two launches'
  // state slabs, 67 MB, exceed the L2s; -1 ... -3 % for log-likelihood-only runs (the state ping-pong stays cached), for
  // m = 1 and for the rows of several series -- bit 6 forces it on everywhere)
  constexpr bool kNtGather = ((BSSM_V2_NT) & 8) && ((WX && !MS && MP == 4) || ((BSSM_V2_NT) & 64));
  auto gather = [&](int r) {
#pragma unroll
    for (int c = 0; c < MP; c++) {  // (padding components re-read the last real one: no branch per load)
      const double av = COH ? ld32_sc1(st_old + (size_t)min(c, m - 1) * N, (uint32_t)anc[r])
                            : (kNtGather ? ld32_nt(st_old + (size_t)min(c, m - 1) * N, anc[r]) : ld32(st_old + (size_t)min(c, m - 1) * N, anc[r]));
      a[r % ROLL][c] = (c < m) ? av : 0.0;
    }
  };
#pragma unroll
  for (int r = 0; r < ROLL; r++) gather(r);
#if BSSM_V2_PIN_BLK
  // the Philox words of the transition normals stay raw integers until here: the compiler otherwise starts the
  // Box-Muller arithmetic before the search and carries (spills) its double intermediates across it
#pragma unroll
  for (int r = 0; r < PPT; r++) __asm__ volatile("" : "+v"(blk[r].x), "+v"(blk[r].y), "+v"(blk[r].z));
#endif
  if constexpr ((BSSM_V2_PRIO) != 0) __builtin_amdgcn_s_setprio(0);
  V2_MARK("phaseE"); V2_STAMP(8);
  // ---- phase E: transition normals, alpha_{t+1} = C_t + T_t a + R_t eta (:663-670), signal at t+1
  const bool have_obs = tn < n;
  const cdbl* row = as_uniform_table(b.obs + (size_t)f * b.obs_fstride + (size_t)(have_obs ? tn : 0) * (ROW_HDR + (MS ? b.p : 1) * (SER_HDR + MP)));
  double s[PPT], w[PPT], xs[(MS && !kFuse) ? PPT : 1][MP];
  const bool weigh = have_obs && row[RH_FLAG] != 0.0;  // otherwise nothing is observed at t+1: the weights are ones
  const double* __restrict__ fpar = kFn ? b.fpar + (size_t)f * b.fpar_fstride : nullptr;
  const double* __restrict__ fn_ob = kFn ? b.obs + (size_t)f * b.obs_fstride + (size_t)(have_obs ? tn : 0) * (ROW_HDR + SER_HDR + MP) : nullptr;
  const double* __restrict__ fn_amats = (kFn && b.amats) ? b.amats + (size_t)f * b.amats_fstride : nullptr;
#pragma unroll
  for (int r = 0; r < PPT; r++) {
    const int li = v2_li<PPT>(r, tid), j = base + li;
    double z[KP], x[MP];
    if constexpr (kEarlyZ >= 1) {
#pragma unroll
      for (int c = 0; c < 4 && c < KP; c++) z[c] = zq[r][c];
      if constexpr (KP > 4) {
        if constexpr (kEarlyZ >= 2) {
#pragma unroll
          for (int c = 4; c < KP; c++) z[c] = zq[r][c];
        } else {
          const u32x4 w2 = philox4x32_10((uint32_t)j, (uint32_t)t, 1u, RNG_TRANSITION, key.seed, key.stream);
          box_muller_packed(w2.x, w2.y, z[4], z[5 < KP ? 5 : 4]);
          if constexpr (KP > 6) box_muller_packed(w2.z, w2.w, z[6], z[7 < KP ? 7 : 6]);
        }
      }
    } else if constexpr (!kFn) {
      rng_normals_blk<KP>(blk[r], key, RNG_TRANSITION, (uint32_t)t, (uint32_t)j, z);
    }
    if constexpr (kFn) {
      if (!psi_mode) functor_propagate<MP, KP, FID>(fpar, key, blk[r], (uint32_t)t, (uint32_t)j, a[r % ROLL], x);
      else if (psi_mode == 2) functor_ekpf_propagate<MP, FID>(fpar, fn_ob, tn >= n, key, blk[r], (uint32_t)t, (uint32_t)j, a[r % ROLL], x);
      else {
        rng_normals_blk<KP, false>(blk[r], key, RNG_TRANSITION, (uint32_t)t, (uint32_t)j, z);
        int zero = 0;
        __asm__ volatile("" : "+v"(zero));
        propagate<MP, KP>((const double*)(s_mat + zero), a[r % ROLL], z, x);
      }
    } else if constexpr (kMfma) {
      propagate_mfma8_h<KP>((const double*)s_mat, a[r % ROLL], z, x, sw_base, lane);
    } else if constexpr (kMfma16) {
      propagate_mfma16_q<KP, 64>((const double*)s_mat, a[r % ROLL], z, x, sw_base, lane);
    } else {
      int zero = 0;  // opaque: the matrix is re-read per particle instead of being kept live in 72 VGPRs
      __asm__ volatile("" : "+v"(zero));
      propagate<MP, KP>((const double*)(s_mat + zero), a[r % ROLL], z, x);
    }
    if constexpr (kFn) {  // (before the next row's gather overwrites the ancestor)
      w[r] = have_obs ? weight_dispatch<MP, FN, PM>(b, fpar, fn_ob, psi_mode == 2 ? fn_amats : fn_amats + (size_t)t * fn_amat_row<MP>(), tn, x, a[r % ROLL]) : 0.0;
      __asm__ volatile("" : "+v"(w[r]));
    }
    if (r + ROLL < PPT) {
      __builtin_amdgcn_sched_barrier(0);
      gather(r + ROLL);
    }
    if (li < len) {
#pragma unroll
      for (int c = 0; c < MP; c++) if (c < m) { if constexpr ((BSSM_V2_NT) & 4) st32_nt(st_new + (size_t)c * N, li, x[c]); else st32(st_new + (size_t)c * N, li, x[c]); }
    }
    if constexpr (kFn) {
    } else if constexpr (kFuse) {
      w[r] = weigh ? v2_log_weight_multi_one<MP>(b.p, row, b.psi != 0, x) : 0.0;
      __asm__ volatile("" : "+v"(w[r]));
    } else if constexpr (MS) {
#pragma unroll
      for (int c = 0; c < MP; c++) xs[r][c] = x[c];
    } else {
      s[r] = v2_signal<MP>(row + ROW_HDR, x);
      __asm__ volatile("" : "+v"(s[r]));  // now, not in the density phase: the compiler otherwise keeps (spills) x until then
    }
    __builtin_amdgcn_sched_barrier(0);  // particles one after the other: bounds the live registers of the transcendental code
  }
  V2_MARK("density"); V2_STAMP(9);
  if (BSSM_V2_RMW_LATE && wt) rmw_load(wt_old);
  if constexpr (!kFuse) {
    if (weigh) {
      if constexpr (MS) v2_log_weights_multi<MP, PPT>(b.p, row, b.psi != 0, xs, w);
      else v2_log_weights<PPT>(row, b.psi != 0, s, w);
    } else {  // nothing observed at t+1: weights are ones (src/model_ssm_ung.cpp:644-647)
#pragma unroll
      for (int r = 0; r < PPT; r++) w[r] = 0.0;
    }
  }
  if (BSSM_V2_RMW_LATE && wt) rmw_store(wt_old);
  if constexpr ((BSSM_V2_PRIO) == 3 || (BSSM_V2_PRIO) == 4) __builtin_amdgcn_s_setprio(3);  // the block statistics: waves about to finish first
  V2_MARK("stats"); V2_STAMP(10);
#pragma unroll
  for (int r = 0; r < PPT; r++) if (v2_li<PPT>(r, tid) >= len) w[r] = -INFINITY;
  v2_block_stats<PPT, WX>(b, ws, f, pb, base, len, tn, w, s_x);
  V2_STAMP(11);
  return true;
}

// one launch per time step: the kernel boundary is the grid-wide synchronisation
template <int MP, int KP, int PPT, bool MS = false, bool WX = false, bool EXACT = false>
__global__ void __launch_bounds__(GRID_NT, v2_occupancy(MP, PPT, MS)) pf_grid_step_v2(const FilterBatch b, const GridWorkspace ws, const int t) {
  int pb, f;
  v2_block_of(ws, pb, f);
  (void)v2_step_body<MP, KP, PPT, MS, WX, EXACT, false>(b, ws, t, pb, f, true);
}
// the same step for a functor model (ssm_nlg / ssm_sde; FN = its id or FN_ANY, PM = the sampler); tail guards always on
constexpr int v2_fn_occupancy(int mp, int ppt) { return mp * ppt <= 2 ? 4 : (mp * ppt <= 4 ? 3 : 2); }
template <int MP, int PPT, bool WX, int FN, int PM>
__global__ void __launch_bounds__(GRID_NT, v2_fn_occupancy(MP, PPT)) pf_grid_step_v2_fn(const FilterBatch b, const GridWorkspace ws, const int t) {
  int pb, f;
  v2_block_of(ws, pb, f);
  (void)v2_step_body<MP, MP, PPT, false, WX, false, false, FN, PM>(b, ws, t, pb, f, true);
}
#ifndef __HIPCC_RTC__   // (host-side kernel selection: not part of the device-only view hiprtc compiles)
template <int MP, int PPT, int FNV>
auto v2_fn_kernels(int psi, bool wx, void (**init)(const FilterBatch, const GridWorkspace)) -> void (*)(const FilterBatch, const GridWorkspace, const int) {
  // (BSSM_FN_PSI_MASK, pf_filter_impl.h: the samplers this translation unit instantiates -- a user model's library builds them in separate units)
  if constexpr (((BSSM_FN_PSI_MASK) & 1) != 0) { if (psi == 0) { *init = pf_grid_init_v2<MP, PPT, false, FNV, 0>; return wx ? pf_grid_step_v2_fn<MP, PPT, true, FNV, 0> : pf_grid_step_v2_fn<MP, PPT, false, FNV, 0>; } }
  if constexpr (((BSSM_FN_PSI_MASK) & 2) != 0) { if (psi == 1) { *init = pf_grid_init_v2<MP, PPT, false, FNV, 1>; return wx ? pf_grid_step_v2_fn<MP, PPT, true, FNV, 1> : pf_grid_step_v2_fn<MP, PPT, false, FNV, 1>; } }
  if constexpr (((BSSM_FN_PSI_MASK) & 4) != 0) { if (psi == 2) { *init = pf_grid_init_v2<MP, PPT, false, FNV, 2>; return wx ? pf_grid_step_v2_fn<MP, PPT, true, FNV, 2> : pf_grid_step_v2_fn<MP, PPT, false, FNV, 2>; } }
  return nullptr;
}


#endif

#ifdef BSSM_XCD_PERSIST
}  // namespace bssm
#include "experiments/xcd_persist.h"   // measured and NOT kept (profiles/r04_xcd_persistent.txt): only `make persist` builds see it
namespace bssm {
#endif

#ifndef __HIPCC_RTC__
template <int MP, int KP, int PPT, bool MS>
inline auto v2_step_kernel(bool wx, bool exact) -> void (*)(const FilterBatch, const GridWorkspace, const int) {
  if (wx) return exact ? pf_grid_step_v2<MP, KP, PPT, MS, true, true> : pf_grid_step_v2<MP, KP, PPT, MS, true, false>;
  return exact ? pf_grid_step_v2<MP, KP, PPT, MS, false, true> : pf_grid_step_v2<MP, KP, PPT, MS, false, false>;
}

#endif

}  // namespace bssm
