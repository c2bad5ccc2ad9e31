// pf_row_step.h -- a filter of at most 16 384 particles spread over its ROWS: one wave per row of 64 particles, one launch
// per time step, results BIT-IDENTICAL to pf_wg_kernel's (both evaluate pf_canon.h).
//
// Why it exists (round-4 review, item 1; src/approx_mcmc.cpp:364-417): a particle system that takes more than half a CU's LDS
// runs one workgroup-resident filter per CU, so a post-correction shard goes through the chip in rounds of 256 filters that all
// take the same time, and a sparsely filled LAST round (one rank's 293-draw share of config C5 at eight ranks = 256 + 37) costs a
// full round on 14 % of the CUs.  Those tail draws go through this kernel instead: 37 filters x 64 rows = 2 368 waves per
// step over the whole chip.  Because a draw's bits do not depend on which family ran it, the split is invisible in the output
// -- the result of a post-correction call is the same for every device list.
//
// Exchange between launches (GridWorkspace with chunk = 64, B = rows): `lcdf[2][F][N]` the row-local inclusive scans of
// exp(w - M_row), `partials[2][F][B]` = (M_row, S_row); the kernel boundary is the grid-wide synchronisation.  A step has two
// dependent global round trips behind the partials (the rows a wave draws from -- contiguous in `lcdf`, <= 4 per round -- and the
// ancestor states), one workgroup barrier (math tables + the combine of wave 0), no coarse table; the weights output of time t is
// written row-scaled by the launch that computes it and rescaled by the next one.
#pragma once

// analysis builds (-DBSSM_ROW_STAMPS, `make rowstamps`): shader-clock stamps of one wave (workgroup 3 of filter 0, wave 1) at the
// phase boundaries, each behind a wait for the wave's outstanding memory operations so that latency is charged to the phase
// that issued the loads (scripts/row_phase_probe.py through bssm_plan_debug_timeline)
#ifdef BSSM_ROW_STAMPS
#define ROW_STAMP(k)                                                                                 \
  do {                                                                                               \
    __asm__ volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                   \
    if (ws.dbg && blockIdx.x == 3 && blockIdx.y == 0 && threadIdx.x == 64 && t >= 1)                   \
      ws.dbg[(size_t)t * 16 + (k)] = (long long)__builtin_readcyclecounter();                        \
    __asm__ volatile("" ::: "memory");                                                               \
  } while (0)
#else
#define ROW_STAMP(k)
#endif

namespace bssm {

constexpr int ROW_NT = 256;    // four rows per workgroup
constexpr int ROW_WIN = 256;   // entries of the exchange array a wave stages per search round (four source rows)

// the filter variants this family carries: functor models of every sampler and the linear-state models without a matrix-core
// transition (pf_wg_kernel runs eight- and sixteen-dimensional states on the MFMA units, whose summation order differs)
template <int MP, int FN> constexpr bool row_step_shape() { return FN != FN_NONE || MP <= 4; }
constexpr int row_occupancy(int mp, bool fn) { return fn ? (mp <= 2 ? 4 : 2) : (mp <= 2 ? 5 : 4); }

// row statistics of the new weights and their hand-over to the next launch
__device__ __forceinline__ void row_publish(const FilterBatch& b, const GridWorkspace& ws, const int f, const int g, const int j,
                                            const bool act, const int tn, const double w, const int lane) {
  const CanonRow rs = canon_row_stats(act ? w : -INFINITY);
  double* __restrict__ lcdf_new = ws.lcdf + ((size_t)(tn & 1) * b.F + f) * b.N;
  if (act) lcdf_new[j] = rs.sc;
  // weights output: exp(w - M_row) now; launch tn multiplies by exp(M_row - M) once the maximum of the time point is known (the
  // last time point has no launch of its own: its weights are ones and leave here as they are)
  if (b.weights && act) b.weights[((size_t)f * (b.n + 1) + tn) * b.N + j] = rs.e;
  double* __restrict__ part = ws.partials + (((size_t)(tn & 1) * b.F + f) * ws.B + g) * 2;
  if (lane == 0) part[0] = rs.M;
  if (lane == 63) part[1] = rs.sc;
}

// wave 0 of a workgroup: the (M_row, S_row) partials of the time-t weights of one filter, combined and handed to the other waves
__device__ __forceinline__ void row_combine_publish(const double* __restrict__ part, const int B, double* s_comb, const int lane) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  double pmx[ROW_NP], psm[ROW_NP];
#pragma unroll
  for (int k = 0; k < ROW_NP; k++) {
    pmx[k] = -INFINITY; psm[k] = 0.0;
    if (lane + 64 * k < B) { const d2 pp = *(const d2*)(part + 2 * (lane + 64 * k)); pmx[k] = pp.x; psm[k] = pp.y; }
  }
  canon_publish(canon_combine<ROW_NP>(pmx, psm, B), s_comb, lane);
}

// t = -1: initial particles and their weights (src/model_ssm_ung.cpp:626, :543; functor models: pf_functors.h)
template <int MP, int KP, int FN = FN_NONE, int PM = -1>
__global__ void __launch_bounds__(ROW_NT) pf_row_init(const FilterBatch b, const GridWorkspace ws) {
  constexpr int FID = FN > 0 ? FN : 0;
  constexpr bool kPsiFn = FN != FN_NONE && PM == 1, kPack = FN == FN_NONE;
  const int psi_mode = PM >= 0 ? PM : b.psi;
  math_tab_init();
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int f = ws.f0 + blockIdx.y, g = blockIdx.x * 4 + wave, N = b.N, n = b.n, m = b.m;
  if (g >= ws.B) return;
  const int j = g * CANON_ROW + lane;
  const bool act = j < N;
  const RngKey key{b.seed, b.stream0 + (uint32_t)f};
  const double* __restrict__ obs = b.obs + (size_t)f * b.obs_fstride;
  const double* __restrict__ ini = b.init + (size_t)f * b.init_fstride;
  const double* __restrict__ fpar = FN ? b.fpar + (size_t)f * b.fpar_fstride : nullptr;
  const double* __restrict__ amats = (FN && b.amats) ? b.amats + (size_t)f * b.amats_fstride : nullptr;
  double x[MP];
  if (FN && !psi_mode) {
    if constexpr (FN != FN_NONE) functor_initial<MP, FID>(fpar, ini, key, (uint32_t)j, x);
  } else {
    double z[MP];
    rng_normals<MP, kPack>(key, RNG_INIT, 0, (uint32_t)j, z);
    initial_state<MP>(ini, z, x);
  }
  double* __restrict__ st_new = b.alpha_hist ? b.alpha_hist + (size_t)f * (n + 1) * m * N : ws.state[0] + (size_t)f * m * N;
  if (act) {
#pragma unroll
    for (int c = 0; c < MP; c++) if (c < m) st_new[(size_t)c * N + j] = x[c];
  }
  double w;
  if constexpr (kPsiFn) w = functor_psi_weight<MP, FID>(as_uniform_table(fpar), as_uniform_table(obs), as_uniform_table(amats), 0, x, x);
  else w = weight_dispatch<MP, FN, PM>(b, fpar, obs, amats, 0, x, x);
  row_publish(b, ws, f, g, j, act, 0, w, lane);
}

// t in [0, n): resample on the time-t weights, propagate to t + 1, weigh there (ones when t + 1 == n or nothing is observed)
template <int MP, int KP, int FN = FN_NONE, int PM = -1>
__global__ void __launch_bounds__(ROW_NT, row_occupancy(MP, FN != FN_NONE)) pf_row_step(const FilterBatch b, const GridWorkspace ws, const int t) {
  constexpr int FID = FN > 0 ? FN : 0;
  constexpr bool kPsiFn = FN != FN_NONE && PM == 1, kPack = FN == FN_NONE;
  constexpr int ucnt = step_draw_count<KP, FN>();
  const int psi_mode = PM >= 0 ? PM : b.psi;
  // wave priority: a step of the row family is a short chain of dependent latencies; when it runs BESIDE a full round of the workgroup
  // kernel (the tail of a post-correction chunk, api_postcorrect.cpp) its waves share SIMDs with waves that are bound by VALU issue, and
  // at equal priority a step took ~25 us instead of ~7 -- the tail then outlasted the round it was meant to hide under
  __builtin_amdgcn_s_setprio(3);
  __shared__ double s_win[4 * ROW_WIN];
  __shared__ double s_comb[3 * ROW_NP * 64 + 2];  // the combine of wave 0 (pf_canon.h: canon_publish)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int f = ws.f0 + blockIdx.y, g = blockIdx.x * 4 + wave, B = ws.B, N = b.N, n = b.n, m = b.m, tn = t + 1;
  ROW_STAMP(0);
  const MathTabRegs mtab = math_tab_load_begin256();
  const RngKey key{b.seed, b.stream0 + (uint32_t)f};
  const int j = g * CANON_ROW + lane;
  const bool has_row = g < B, act = has_row && j < N;
  const double* __restrict__ lcdf = ws.lcdf + ((size_t)(t & 1) * b.F + f) * N;
  const double* __restrict__ st_old = b.alpha_hist ? b.alpha_hist + ((size_t)f * (n + 1) + t) * m * N : ws.state[t & 1] + (size_t)f * m * N;
  // ---- the (M_row, S_row) partials of the time-t weights: combined by wave 0 while the others draw their step blocks
  if (wave == 0) row_combine_publish(ws.partials + ((size_t)(t & 1) * b.F + f) * B * 2, B, s_comb, lane);
  const u32x4 blk = rng_step_block(key, (uint32_t)t, (uint32_t)j);
  const uint32_t uword = rng_resample_word<ucnt>(key, (uint32_t)t, (uint32_t)j, blk);
  ROW_STAMP(1);  // tables + partials (wave 0: + combine) + step block
  math_tab_load_end256(mtab);
  lds_barrier();  // (1) the only workgroup barrier of the step
  ROW_STAMP(2);  // barrier wait
  const CanonPrefix<ROW_NP> px = canon_prefix_load<ROW_NP>(s_comb, lane, B);
  const double M = canon_published_M<ROW_NP>(s_comb), S = canon_published_S<ROW_NP>(s_comb);
  if (b.weights && act) {  // the stored weights of time t are exp(w - M_row): to the scale of the whole time point (also on a degenerate step, :676-682)
    double* __restrict__ wt = b.weights + ((size_t)f * (n + 1) + t) * N;
    wt[j] = wt[j] * canon_published_scale<ROW_NP>(s_comb, g);
  }
  if (!(S > 0.0)) {
    // degenerate weights (src/model_ssm_ung.cpp:678-682): stop and poison this row's slot of the next parity, so every later
    // launch sees a non-finite S too (no flag load on healthy steps); later output slots stay zero like the reference's
    if (has_row && lane == 0) {
      double* __restrict__ pn = ws.partials + (((size_t)(tn & 1) * b.F + f) * B + g) * 2;
      pn[0] = __builtin_nan("");
      pn[1] = __builtin_nan("");
      if (g == 0) atomicCAS(&b.dead_at[f], 0, t + 1);
    }
    return;
  }
  if (!has_row) return;  // (a workgroup's spare waves: nothing below synchronises across waves)
  const double c_scale = canon_stratum(S, N);
  if (g == B - 1 && lane == 63) ws.llinc[(size_t)f * (n + 1) + t] = canon_llinc(M, c_scale);  // :683, single writer

  // ---- stratified resampling (pf_canon.h): source row per slot, then the bisection inside it.  The rows [c_lo, c_hi] a wave draws
  // from are contiguous in the exchange array: staged four at a time in the wave's window
  const CanonSource src = canon_source_row(canon_target(uword, j, c_scale), act, px, lane);
  ROW_STAMP(3);  // source rows
  double* sw = s_win + wave * ROW_WIN;
  int anc = 0;
  constexpr bool kLinearStep = FN == FN_NONE || PM == 1;  // the transition is C_t + T_t a + R_t eta (linear-state models, psi-APF proposal)
  double z[KP];
  for (int r0 = src.c_lo; r0 <= src.c_hi; r0 += 4) {  // wave-uniform bounds
    const int nr = min(4, src.c_hi - r0 + 1);
    double xr[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int k = (r0 + q) * CANON_ROW + lane;
      xr[q] = (q < nr && k < N) ? lcdf[k] : 0.0;
    }
    if constexpr (kLinearStep) {  // the transition normals depend on nothing but (seed, t, slot): drawn under the round trip of the rows
      if (r0 == src.c_lo) {
        rng_normals_blk<KP, kPack>(blk, key, RNG_TRANSITION, (uint32_t)t, (uint32_t)j, z);
#pragma unroll
        for (int c = 0; c < KP; c++) __asm__ volatile("" : "+v"(z[c]));
      }
    }
    if (r0 == src.c_lo) ROW_STAMP(4);  // rows loaded, transition normals drawn under them
    lds_wave_sync();  // the previous round's reads are done
#pragma unroll
    for (int q = 0; q < 4; q++) sw[q * CANON_ROW + lane] = xr[q];
    lds_wave_sync();
    if (act && src.e >= r0 && src.e < r0 + 4) {
      const int slen = min(CANON_ROW, N - src.e * CANON_ROW);
      anc = src.e * CANON_ROW + min(canon_count_below(sw + (src.e - r0) * CANON_ROW, slen, src.vp), slen - 1);
    }
  }
  ROW_STAMP(5);  // staged, bisected
  if (b.indices && act) __builtin_nontemporal_store((uint32_t)anc, &b.indices[((size_t)f * n + t) * N + j]);

  // ---- ancestor state, transition to t + 1 (src/model_ssm_ung.cpp:657-670; functor models: pf_functors.h), weight there
  double* __restrict__ st_new = b.alpha_hist ? b.alpha_hist + ((size_t)f * (n + 1) + tn) * m * N : ws.state[tn & 1] + (size_t)f * m * N;
  double a[MP], x[MP];
#pragma unroll
  for (int c = 0; c < MP; c++) a[c] = (c < m) ? st_old[(size_t)c * N + anc] : 0.0;
  ROW_STAMP(6);  // ancestor gather
  const double* __restrict__ mats = b.mats + (size_t)f * b.mats_fstride;
  const double* __restrict__ obs = b.obs + (size_t)f * b.obs_fstride;
  const double* __restrict__ fpar = FN ? b.fpar + (size_t)f * b.fpar_fstride : nullptr;
  const double* __restrict__ amats = (FN && b.amats) ? b.amats + (size_t)f * b.amats_fstride : nullptr;
  const int obs_stride = ROW_HDR + b.p * (SER_HDR + MP);
  const double* __restrict__ ob = obs + (size_t)(tn < n ? tn : 0) * obs_stride;
  if (FN && !psi_mode) {
    if constexpr (FN != FN_NONE) functor_propagate<MP, KP, FID>(fpar, key, blk, (uint32_t)t, (uint32_t)j, a, x);
  } else if (FN && psi_mode == 2) {
    if constexpr (FN != FN_NONE) functor_ekpf_propagate<MP, FID>(fpar, ob, tn >= n, key, blk, (uint32_t)t, (uint32_t)j, a, x);
  } else {
    if constexpr (!kLinearStep) rng_normals_blk<KP, kPack>(blk, key, RNG_TRANSITION, (uint32_t)t, (uint32_t)j, z);  // (PM = -1: the sampler is a run-time value)
    propagate<MP, KP>(as_uniform_table(mats + (size_t)t * b.mats_tstride), a, z, x);
  }
  ROW_STAMP(7);  // transition
  if (act) {
#pragma unroll
    for (int c = 0; c < MP; c++) if (c < m) st_new[(size_t)c * N + j] = x[c];
  }
  double w;
  if constexpr (kPsiFn) w = tn < n ? functor_psi_weight<MP, FID>(as_uniform_table(fpar), as_uniform_table(ob), as_uniform_table(amats + (size_t)t * fn_amat_row<MP>()), tn, x, a) : 0.0;
  else w = tn < n ? weight_dispatch<MP, FN, PM>(b, fpar, ob, (FN && psi_mode == 2) ? amats : amats + (size_t)t * fn_amat_row<MP>(), tn, x, a) : 0.0;
  ROW_STAMP(8);  // state store + weight
  row_publish(b, ws, f, g, j, act, tn, w, lane);
  ROW_STAMP(9);  // row statistics + hand-over
  ROW_STAMP(10); ROW_STAMP(11);
}

#ifndef __HIPCC_RTC__   // host side
// workspace of the row family: GridWorkspace with chunk = 64 (no coarse table, no weights scale factors, no control block)
inline size_t row_workspace_bytes_impl(const FilterBatch& b) {
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  const size_t B = (size_t)(b.N + CANON_ROW - 1) / CANON_ROW;
  return 2 * al((size_t)b.F * b.m * b.N * sizeof(double)) + al((size_t)2 * b.F * b.N * sizeof(double)) + al((size_t)2 * b.F * B * 2 * sizeof(double)) +
         al((size_t)2 * b.F * (b.n + 1) * sizeof(double));
}

template <int MP, int KP, int FNV>
auto row_fn_kernels(int psi, void (**init)(const FilterBatch, const GridWorkspace)) -> void (*)(const FilterBatch, const GridWorkspace, const int) {
  if constexpr (((BSSM_FN_PSI_MASK) & 1) != 0) { if (psi == 0) { *init = pf_row_init<MP, KP, FNV, 0>; return pf_row_step<MP, KP, FNV, 0>; } }
  if constexpr (((BSSM_FN_PSI_MASK) & 2) != 0) { if (psi == 1) { *init = pf_row_init<MP, KP, FNV, 1>; return pf_row_step<MP, KP, FNV, 1>; } }
  if constexpr (((BSSM_FN_PSI_MASK) & 4) != 0) { if (psi == 2) { *init = pf_row_init<MP, KP, FNV, 2>; return pf_row_step<MP, KP, FNV, 2>; } }
  return nullptr;
}

// the time loop of the row family: one launch per step
inline hipError_t launch_row_loop(const FilterBatch& b, const GridWorkspace& ws, hipStream_t stream, StepTiming* timing,
                                  void (*kern)(const FilterBatch, const GridWorkspace, const int), void (*kern_init)(const FilterBatch, const GridWorkspace),
                                  const int max_rows = ROW_MAX_ROWS) {
  if (b.N > CANON_ROW * max_rows || ws.chunk != CANON_ROW || ws.B != (b.N + CANON_ROW - 1) / CANON_ROW) return hipErrorInvalidValue;
  hipError_t e;
  if ((e = hipMemsetAsync(b.dead_at, 0, sizeof(int) * b.F, stream)) != hipSuccess) return e;
  if ((e = hipMemsetAsync(ws.llinc, 0, sizeof(double) * b.F * (b.n + 1), stream)) != hipSuccess) return e;
  const int wgs = (ws.B + 3) / 4;
  // A step is a chain of dependent latencies, not work (a 19-filter launch of the C5 shape: 6.7 us, during which the chip is busy a
  // fraction of the time), so a batch is stepped as interleaved launch sequences over disjoint groups of its filters; more than
  // two sequences measured no better (profiles/r05_row_family_probe.txt), BSSM_AMD_GRID_STREAMS = 1 ... 4 overrides
  constexpr int kMaxParts = 4;
  int parts = b.F >= 8 ? 2 : 1;
  if (opt().grid_streams > 0) parts = opt().grid_streams;
  parts = std::max(1, std::min({parts, kMaxParts, b.F}));
  if (!ws.side) parts = 1;
  hipStream_t lanes[kMaxParts] = {stream, stream, stream, stream};
  GridWorkspace wsp[kMaxParts] = {ws, ws, ws, ws};
  int fcount[kMaxParts] = {b.F, 0, 0, 0};
  hipEvent_t fork = nullptr, join = nullptr;
  if (parts >= 2) {
    static thread_local TlsStream extra_side[kMaxParts - 2];
    lanes[1] = ws.side;
    for (int q = 2; q < parts; q++) { lanes[q] = extra_side[q - 2].get(); if (!lanes[q]) { parts = q; break; } }
    int f_at = 0;
    for (int q = 0; q < parts; q++) { fcount[q] = (b.F * (q + 1)) / parts - (b.F * q) / parts; wsp[q].f0 = ws.f0 + f_at; f_at += fcount[q]; }
    if ((e = hipEventCreateWithFlags(&fork, hipEventDisableTiming)) != hipSuccess) return e;
    (void)hipEventRecord(fork, stream);
    for (int q = 1; q < parts; q++) (void)hipStreamWaitEvent(lanes[q], fork, 0);
  }
  if (timing) timing->concurrent = parts;
  for (int t = -1; t < b.n; t++) {
    for (int q = 0; q < parts; q++) {
      if (t < 0) hipLaunchKernelGGL(kern_init, dim3(wgs, fcount[q]), dim3(ROW_NT), 0, lanes[q], b, wsp[q]);
      else hipLaunchKernelGGL(kern, dim3(wgs, fcount[q]), dim3(ROW_NT), 0, lanes[q], b, wsp[q], t);
    }
  }
  if (parts >= 2) {
    for (int q = 1; q < parts; q++) {
      if ((e = hipEventCreateWithFlags(&join, hipEventDisableTiming)) != hipSuccess) return e;
      (void)hipEventRecord(join, lanes[q]);
      (void)hipStreamWaitEvent(stream, join, 0);
      (void)hipEventDestroy(join);
    }
    (void)hipEventDestroy(fork);
  }
  if ((e = hipGetLastError()) != hipSuccess) return e;
  hipLaunchKernelGGL(loglik_finalize_kernel, dim3(b.F), dim3(64), 0, stream, b.loglik, ws.llinc, b.dead_at, b.n, b.F);
  return hipGetLastError();
}

template <int MP, int KP>
hipError_t launch_row_mk(const FilterBatch& b, const GridWorkspace& ws, hipStream_t stream, StepTiming* timing) {
  constexpr int kFn = fn_kernel_id<MP>();
  if (b.fpar && kFn > 0 && b.fn_id != kFn) return hipErrorInvalidValue;
  void (*kern)(const FilterBatch, const GridWorkspace, const int) = nullptr;
  void (*kern_init)(const FilterBatch, const GridWorkspace) = nullptr;
  if (b.fpar) {
    if constexpr (MP == KP && MP <= FN_MP_MAX) kern = row_fn_kernels<MP, KP, kFn>(b.psi, &kern_init);
  } else {
#ifndef BSSM_USER_MODEL
    if constexpr (row_step_shape<MP, FN_NONE>()) { kern = pf_row_step<MP, KP>; kern_init = pf_row_init<MP, KP>; }
#endif
  }
  if (!kern || !kern_init) return hipErrorInvalidValue;
#ifndef BSSM_USER_MODEL
  { char nm[96]; std::snprintf(nm, sizeof nm, "pf_row_step<%d,%d,%s>", MP, KP, b.fpar ? (kFn > 0 ? "model" : "any") : "false"); note_filter_kernel(nm); }
#endif
  return launch_row_loop(b, ws, stream, timing, kern, kern_init);
}

#endif  // !__HIPCC_RTC__

}  // namespace bssm
