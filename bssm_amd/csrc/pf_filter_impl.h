// pf_filter_impl.h -- hand-written CDNA4 (gfx950) kernels for the bssm particle-filter hot path.
//
// Execution shapes for the same recursion (src/model_ssm_ung.cpp:611-712 bootstrap,
// :536-609 psi-APF, src/model_ssm_ulg.cpp:557-643 gaussian):
//
//  * pf_wg_kernel      one WORKGROUP per filter, particle system resident in LDS, the whole
//                      time loop inside one launch (2 workgroup barriers per time step, no
//                      grid-wide synchronisation).  This is the shape of IS post-correction:
//                      thousands of independent filters (src/approx_mcmc.cpp:370-417).
//  * pf_grid_step_v2   (pf_grid_v2.h) one filter spread over many workgroups (N up to 131 072), one
//                      launch per time step; the kernel boundary is the only grid-wide synchronisation.
//                      Blocks exchange (max, sum) partials and block-local weights / scans so
//                      that the global CDF never has to be materialised.
//  * pf_row_step       (pf_row_step.h) one wave per row of 64 particles, one launch per step, the
//                      arithmetic of the workgroup kernel (pf_canon.h): N <= 16 384.
//  * pf_big_*          (pf_big.h) states of 17 ... 64 dimensions on the matrix cores.
//
// Both draw every random number from the counter-based stream in pf_device.h and use
// fixed-shape reductions/scans only (no floating-point atomics): results are bit-reproducible
// run to run and independent of the number of GPUs a batch is sharded over.
#pragma once
#ifndef __HIPCC_RTC__
#include <type_traits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#endif

#include "pf_device.h"
#include "pf_canon.h"
#include "pf_functors.h"
#include "pf_kernels.h"
#ifndef __HIPCC_RTC__
#include "options.h"
#include "tls_stream.h"
#endif

namespace bssm {

template <class A, class B> struct bssm_same_type { static constexpr bool value = false; };
template <class A> struct bssm_same_type<A, A> { static constexpr bool value = true; };

// loglik[f] = sum_t llinc[f][t] in time order (the reference's accumulation order), -inf after a degenerate step.
// (static: every translation unit that instantiates the step kernels carries its own copy, incl. run-time compiled
// user models)
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding global load and
// store of the wave (s_waitcnt vmcnt(0)) before it arrives; the filter kernels exchange data between lanes through
// LDS alone (history / state stores are never read back inside a launch), so their barriers need not wait for the
// HBM round trip of stores that are still in flight.
__device__ __forceinline__ void lds_barrier() {
  __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

static __global__ void loglik_finalize_kernel(double* __restrict__ loglik, const double* __restrict__ llinc,
                                              const int* __restrict__ dead_at, int n, int F) {
  // one wave per filter: coalesced loads of 64 increments, then the increments are added in time order by reading the
  // lanes one after the other (the same summation order as a sequential loop, without its serialised load latency)
  const int f = blockIdx.x, lane = threadIdx.x;
  if (f >= F) return;
  const double* src = llinc + (size_t)f * (n + 1);
  double acc = 0.0;
  for (int t0 = 0; t0 < n; t0 += 64) {
    const double v = (t0 + lane < n) ? src[t0 + lane] : 0.0;
    const int cnt = min(64, n - t0);
    const int lo = __double2loint(v), hi = __double2hiint(v);
    if (cnt == 64) {
#pragma unroll
      for (int j = 0; j < 64; j++)
        acc += __hiloint2double(__builtin_amdgcn_readlane(hi, j), __builtin_amdgcn_readlane(lo, j));
    } else {
      for (int j = 0; j < cnt; j++)
        acc += __hiloint2double(__builtin_amdgcn_readlane(hi, j), __builtin_amdgcn_readlane(lo, j));
    }
  }
  if (lane == 0) loglik[f] = dead_at[f] ? -INFINITY : acc;
}


// ------------------------------------------------------------------------------------------
// shared per-particle pieces
// ------------------------------------------------------------------------------------------
template <int MP, class PT>
__device__ __forceinline__ double signal_of(PT ob, const double (&x)[MP]) {
  // D(t) + Z_t' alpha + xbeta(t)   (src/model_ssm_ung.cpp:462-463); D + xbeta pre-added on the host
  double za = 0.0;
#pragma unroll
  for (int c = 0; c < MP; c++) za += ob[SER_HDR + c] * x[c];
  return ob[OB_D] + za;
}

// log-weight of one particle at one time point: sum over the observed series of the row
// (univariate: src/model_ssm_ung.cpp:447-493 / :385-439; multivariate: src/model_ssm_mng.cpp:407-442 / :357-393)
template <int MP, class PT>
__device__ __forceinline__ double log_weight_of(const FilterBatch& b, PT row, const double (&x)[MP]) {
  if (row[RH_FLAG] == 0.0) return 0.0;  // nothing observed at t: weights are ones (src/model_ssm_ung.cpp:644-647)
  double w = 0.0;
  for (int j = 0; j < b.p; j++) {
    const PT ob = row + ROW_HDR + j * (SER_HDR + MP);
    if (ob[OB_FLAG] == 0.0) continue;
    const double s = signal_of<MP, PT>(ob, x);
    const int dist = (int)ob[OB_DIST];
    w += b.psi ? log_psi_weight(dist, ob, ob[OB_PHI], s) : log_obs_density(dist, ob, ob[OB_PHI], s);
  }
  return b.psi ? w - row[RH_SCALE] : w;
}

// alpha_{t+1} = C_t + T_t a + R_t eta   (src/model_ssm_ung.cpp:663-670); for the psi-APF the host
// folds alphahat/Ct/Vt into the same three operands (src/model_ssm_ung.cpp:585-586).
// BSSM_PROPAGATE_COLS: for eight- and sixteen-dimensional states read through scalar loads the matrices go through the scalar
// file ONE COLUMN AT A TIME (16 / 32 SGPRs), each column's loads re-issued behind an opaque pointer and pinned in place.  In the
// row-by-row form the compiler requests all 136 / 528 doubles up front -- 272+ SGPRs against a file of 106: the rest lived in VGPR
// lanes (600-2000 spill slots in the m = 8 workgroup kernels) and every use came back through v_readlane.  Every x_j is still
// accumulated over the columns in the same order with the same fused multiply-adds: same bits.
#ifndef BSSM_PROPAGATE_COLS
#define BSSM_PROPAGATE_COLS 1
#endif
template <int MP, int KP, class PT>
__device__ __forceinline__ void propagate(PT mt, const double (&a)[MP], const double (&z)[KP],
                                          double (&x)[MP]) {
  const PT Tm = mt;
  const PT Rm = mt + MP * MP;
  const PT Cv = mt + MP * MP + MP * KP;
  if constexpr ((BSSM_PROPAGATE_COLS) != 0 && MP >= 8 && bssm_same_type<PT, const cdbl*>::value) {
    double s[MP], q[MP];
#pragma unroll
    for (int j = 0; j < MP; j++) { s[j] = 0.0; q[j] = 0.0; }
#pragma unroll
    for (int c = 0; c < MP; c++) {
      const cdbl* col = reload_table(Tm + MP * c);
#pragma unroll
      for (int j = 0; j < MP; j++) s[j] += col[j] * a[c];
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int c = 0; c < KP; c++) {
      const cdbl* col = reload_table(Rm + MP * c);
#pragma unroll
      for (int j = 0; j < MP; j++) q[j] += col[j] * z[c];
      __builtin_amdgcn_sched_barrier(0);
    }
    const cdbl* cv = reload_table(Cv);
#pragma unroll
    for (int j = 0; j < MP; j++) x[j] = cv[j] + s[j] + q[j];
    return;
  }
#pragma unroll
  for (int j = 0; j < MP; j++) {
    double s = 0.0, q = 0.0;
#pragma unroll
    for (int c = 0; c < MP; c++) s += Tm[j + MP * c] * a[c];
#pragma unroll
    for (int c = 0; c < KP; c++) q += Rm[j + MP * c] * z[c];
    x[j] = Cv[j] + s + q;
  }
}

template <int MP, class PT>
__device__ __forceinline__ void initial_state(PT ini, const double (&z)[MP], double (&x)[MP]) {
  // a1 + L u   (src/model_ssm_ung.cpp:626; psi: alphahat_0 + V_0 u, :543)
#pragma unroll
  for (int j = 0; j < MP; j++) {
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < MP; c++) s += ini[MP + j + MP * c] * z[c];
    x[j] = ini[j] + s;
  }
}

// ------------------------------------------------------------------------------------------
// Workgroup-resident filter
// ------------------------------------------------------------------------------------------
// Transition of 64 particles (one wave) for an 8-dimensional state on the matrix cores:
//   X' (8 x 64) = C + [T R] (8 x 16) . [A ; Z] (16 x 64)     with v_mfma_f64_16x16x4_f64
// (ssm_mng's m = 8, k <= 8 case -- the one place on this path where the transition is a real dense contraction;
// src/model_ssm_mng.cpp:611-618).  Operand staging goes through a per-wave LDS tile: lanes hold "my particle, all
// components", the MFMA wants B[k = lane>>4 (+4kk)][j = lane&15] and returns D[row = (lane>>4) + 4 reg][col = lane&15]
// (f64 fragment layout, cdna_hip_programming.md section 3).  Rows 8..15 of [T R] are zero padding.
template <int KP, class PT>
__device__ __forceinline__ void propagate_mfma8(PT mt, const double (&a)[8], const double (&z)[KP],
                                                double (&x)[8], double* tile /* 16 x 64 doubles, this wave's */, int lane) {
  typedef double d4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int c = 0; c < 8; c++) tile[c * 64 + lane] = a[c];
#pragma unroll
  for (int c = 0; c < 8; c++) tile[(8 + c) * 64 + lane] = c < KP ? z[c < KP ? c : 0] : 0.0;
  const int i = lane & 15, kq = lane >> 4;
  double afr[4];
#pragma unroll
  for (int kk = 0; kk < 4; kk++) {
    const int k = 4 * kk + kq;  // column of [T R]
    afr[kk] = (i < 8) ? (k < 8 ? mt[i + 8 * k] : ((k - 8) < KP ? mt[64 + i + 8 * (k - 8)] : 0.0)) : 0.0;
  }
  const PT Cv = mt + 64 + 8 * KP;
  const double c0 = Cv[kq], c1 = Cv[kq + 4];
  __builtin_amdgcn_wave_barrier();
  d4 acc[4];
#pragma unroll
  for (int blk = 0; blk < 4; blk++) {
    acc[blk] = d4{c0, c1, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 4; kk++)
      acc[blk] = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[kk], tile[(4 * kk + kq) * 64 + 16 * blk + i], acc[blk], 0, 0, 0);
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int blk = 0; blk < 4; blk++) {  // rows kq and kq + 4 of the result are state components
    tile[kq * 64 + 16 * blk + i] = acc[blk][0];
    tile[(kq + 4) * 64 + 16 * blk + i] = acc[blk][1];
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int c = 0; c < 8; c++) x[c] = tile[c * 64 + lane];
}

// The same contraction through an 8 x 64 tile (4 KB per wave instead of 8): the ancestors' half of [A ; Z] first, then
// the noise half into the same rows -- the round-2 step kernel keeps four workgroups per CU with it.
__device__ __forceinline__ void lds_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int KP, class PT>
__device__ __forceinline__ void propagate_mfma8_h(PT mt, const double (&a)[8], const double (&z)[KP],
                                                  double (&x)[8], double* tile /* 8 x 64 doubles, this wave's */, int lane) {
  typedef double d4 __attribute__((ext_vector_type(4)));
  const int i = lane & 15, kq = lane >> 4;
  double afr[4];
#pragma unroll
  for (int kk = 0; kk < 4; kk++) {
    const int k = 4 * kk + kq;  // column of [T R]
    afr[kk] = (i < 8) ? (k < 8 ? mt[i + 8 * k] : ((k - 8) < KP ? mt[64 + i + 8 * (k - 8)] : 0.0)) : 0.0;
  }
  const PT Cv = mt + 64 + 8 * KP;
  const double c0 = Cv[kq], c1 = Cv[kq + 4];
  lds_wave_sync();  // earlier readers of the tile (the search window) are done
#pragma unroll
  for (int c = 0; c < 8; c++) tile[c * 64 + lane] = a[c];
  lds_wave_sync();
  d4 acc[4];
#pragma unroll
  for (int blk = 0; blk < 4; blk++) {
    acc[blk] = d4{c0, c1, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 2; kk++)
      acc[blk] = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[kk], tile[(4 * kk + kq) * 64 + 16 * blk + i], acc[blk], 0, 0, 0);
  }
  lds_wave_sync();
#pragma unroll
  for (int c = 0; c < 8; c++) tile[c * 64 + lane] = c < KP ? z[c < KP ? c : 0] : 0.0;
  lds_wave_sync();
#pragma unroll
  for (int blk = 0; blk < 4; blk++) {
#pragma unroll
    for (int kk = 0; kk < (KP > 4 ? 2 : 1); kk++)
      acc[blk] = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[2 + kk], tile[(4 * kk + kq) * 64 + 16 * blk + i], acc[blk], 0, 0, 0);
  }
  lds_wave_sync();
#pragma unroll
  for (int blk = 0; blk < 4; blk++) {  // rows kq and kq + 4 of the result are state components
    tile[kq * 64 + 16 * blk + i] = acc[blk][0];
    tile[(kq + 4) * 64 + 16 * blk + i] = acc[blk][1];
  }
  lds_wave_sync();
#pragma unroll
  for (int c = 0; c < 8; c++) x[c] = tile[c * 64 + lane];
}

// The same contraction through a 4 x 64 tile (2 KB per wave): one K = 4 chunk of [A ; Z] at a time, the result in two halves.
// Same multiply-accumulate order as propagate_mfma8_h (same bits).  For the workgroup-resident filter at four particles per lane,
// whose only free LDS during the transition is its own CDF array (dead between barrier (4) and the next step's barrier (1)).
template <int KP, class PT>
__device__ __forceinline__ void propagate_mfma8_q(PT mt, const double (&a)[8], const double (&z)[KP],
                                                  double (&x)[8], double* tile /* 4 x 64 doubles, this wave's */, int lane) {
  typedef double d4 __attribute__((ext_vector_type(4)));
  const int i = lane & 15, kq = lane >> 4;
  double afr[4];
#pragma unroll
  for (int kk = 0; kk < 4; kk++) {
    const int k = 4 * kk + kq;  // column of [T R]
    afr[kk] = (i < 8) ? (k < 8 ? mt[i + 8 * k] : ((k - 8) < KP ? mt[64 + i + 8 * (k - 8)] : 0.0)) : 0.0;
  }
  const PT Cv = mt + 64 + 8 * KP;
  const double c0 = Cv[kq], c1 = Cv[kq + 4];
  d4 acc[4];
#pragma unroll
  for (int blk = 0; blk < 4; blk++) acc[blk] = d4{c0, c1, 0.0, 0.0};
#pragma unroll
  for (int st = 0; st < (KP > 4 ? 4 : 3); st++) {
    lds_wave_sync();  // earlier readers of the tile are done
#pragma unroll
    for (int c = 0; c < 4; c++) {
      double v;
      if (st < 2) v = a[4 * st + c];
      else { const int zc = 4 * (st - 2) + c; v = zc < KP ? z[zc < KP ? zc : 0] : 0.0; }
      tile[c * 64 + lane] = v;
    }
    lds_wave_sync();
#pragma unroll
    for (int blk = 0; blk < 4; blk++)
      acc[blk] = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[st], tile[kq * 64 + 16 * blk + i], acc[blk], 0, 0, 0);
  }
#pragma unroll
  for (int h = 0; h < 2; h++) {  // rows kq (h = 0) and kq + 4 (h = 1) of the result are state components
    lds_wave_sync();
#pragma unroll
    for (int blk = 0; blk < 4; blk++) tile[kq * 64 + 16 * blk + i] = acc[blk][h];
    lds_wave_sync();
#pragma unroll
    for (int c = 0; c < 4; c++) x[4 * h + c] = tile[c * 64 + lane];
  }
}

// Sixteen-dimensional states (m = 9 ... 16: seasonal components): X' (16 x 64) = C + [T R] (16 x (16 + KP)) . [A ; Z], every row of
// the 16 x 16 result tile a state component.  One K = 4 chunk of [A ; Z] at a time through a 4 x 64 tile (2 KB per wave), the result
// in four quarters.  The scalar-load form needs 336 ... 528 doubles in a scalar file of 106 registers: 4 300 VALU instructions per
// particle-step, most of them v_readlane (profiles/r04_wg_occupancy_ab.txt).
// W = 64: the tile holds a chunk of all 64 particles of the wave; W = 32: of one half-wave at a time (1 KB per wave -- what the CDF
// array of a two-particles-per-lane filter leaves per wave); same multiply-accumulate order per particle either way.
template <int KP, int W, class PT>
__device__ __forceinline__ void propagate_mfma16_q(PT mt, const double (&a)[16], const double (&z)[KP],
                                                   double (&x)[16], double* tile /* 4 x W doubles, this wave's */, int lane) {
  typedef double d4 __attribute__((ext_vector_type(4)));
  constexpr int KZ = (KP + 3) / 4, NK = 4 + KZ;  // K chunks: four of the state, KZ of the noise
  constexpr int NH = 64 / W, BPH = W / 16;        // passes per chunk, 16-particle blocks per pass
  const int i = lane & 15, kq = lane >> 4, lw = lane & (W - 1);
  const bool mine[2] = {NH == 1 || lane < 32, NH == 2 && lane >= 32};
  double afr[NK];
#pragma unroll
  for (int kk = 0; kk < NK; kk++) {
    const int k = 4 * kk + kq;  // column of [T R]
    afr[kk] = k < 16 ? mt[i + 16 * k] : ((k - 16) < KP ? mt[256 + i + 16 * (k - 16)] : 0.0);
  }
  const PT Cv = mt + 256 + 16 * KP;
  d4 acc[4];
  {
    const double c0 = Cv[kq], c1 = Cv[kq + 4], c2 = Cv[kq + 8], c3 = Cv[kq + 12];
#pragma unroll
    for (int blk = 0; blk < 4; blk++) acc[blk] = d4{c0, c1, c2, c3};
  }
#pragma unroll
  for (int st = 0; st < NK; st++) {
#pragma unroll
    for (int h = 0; h < NH; h++) {
      lds_wave_sync();  // earlier readers of the tile are done
      if (mine[h]) {
#pragma unroll
        for (int c = 0; c < 4; c++) {
          double v;
          if (st < 4) v = a[4 * st + c];
          else { const int zc = 4 * (st - 4) + c; v = zc < KP ? z[zc < KP ? zc : 0] : 0.0; }
          tile[c * W + lw] = v;
        }
      }
      lds_wave_sync();
#pragma unroll
      for (int bb = 0; bb < BPH; bb++)
        acc[BPH * h + bb] = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[st], tile[kq * W + 16 * bb + i], acc[BPH * h + bb], 0, 0, 0);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; q++) {  // rows kq + 4 q of the result
#pragma unroll
    for (int h = 0; h < NH; h++) {
      lds_wave_sync();
#pragma unroll
      for (int bb = 0; bb < BPH; bb++) tile[kq * W + 16 * bb + i] = acc[BPH * h + bb][q];
      lds_wave_sync();
      if (mine[h]) {
#pragma unroll
        for (int c = 0; c < 4; c++) x[4 * q + c] = tile[c * W + lw];
      }
    }
  }
}

// one particle's log-weight for whichever model family the kernel was instantiated for
// PM: the sampler of a functor kernel fixed at compile time (0 bootstrap, 1 psi-APF, 2 EKF particle filter; -1 = read
// b.psi): a kernel that carries all three keeps the union of their registers live (the nlg_growth psi-APF kernel at 1024
// threads spilt 52 VGPRs and 161 SGPRs that way)
template <int MP, int FN, int PM = -1>
__device__ __forceinline__ double weight_dispatch(const FilterBatch& b, const double* __restrict__ fpar,
                                                  const double* __restrict__ row, const double* __restrict__ amat, int t,
                                                  const double (&x)[MP], const double (&aprev)[MP]) {
  if constexpr (FN != FN_NONE) {
    constexpr int FID = FN > 0 ? FN : 0;
    const int psi = PM >= 0 ? PM : b.psi;
    {  // EKF particle filter (3 ... 8 states: user models, general small factorisations)
      if (psi == 2) return functor_ekpf_weight<MP, FID>(fpar, row, amat, t, x, aprev);  // amat = the filter's [a1 | P1 | att1 | L1]
    }
    {  // psi-APF: user-defined models with up to 8 states
      if (psi == 1) return functor_psi_weight<MP, FID>(fpar, row, amat, t, x, aprev);
    }
    return functor_log_weight<MP, FID>(fpar, row, x, t);
  } else {
    return log_weight_of<MP>(b, as_uniform_table(row), x);  // wave-uniform row: scalar loads (pf_device.h)
  }
}

// functor-model (FN) kernel shapes: the shipped models have 1 or 2 states; a run-time compiled user model may have up to 8.
// fn_kernel_id<MP>(): the model id the functor kernels of this translation unit are compiled for (FN_ANY: dispatch per
// call) -- a user-model library holds ONE model, and the only shipped two-state model is nlg_growth.
#ifdef BSSM_USER_MODEL
constexpr int FN_MP_MAX = BSSM_USER_M > 8 ? 16 : 8;   // (every sampler up to 8 states since round 4; the bootstrap filter up to 16 since round 5)
#if defined(BSSM_USER_NLG) && !defined(BSSM_USER_SDE)
template <int MP> constexpr int fn_kernel_id() { return FN_USER_NLG; }
#elif defined(BSSM_USER_SDE) && !defined(BSSM_USER_NLG)
template <int MP> constexpr int fn_kernel_id() { return FN_USER_SDE; }
#else
template <int MP> constexpr int fn_kernel_id() { return FN_ANY; }
#endif
#else
constexpr int FN_MP_MAX = 2;
template <int MP> constexpr int fn_kernel_id() { return MP == 2 ? FN_NLG_GROWTH : FN_ANY; }
#endif
// analysis builds (-DBSSM_WG_STAMPS): shader-clock deltas per phase of the time loop, accumulated by the first lane of
// the first and the last wave of workgroup 0 and printed at the end (scripts/wg_phase_probe.py)
#ifdef BSSM_WG_MARKS
#define WG_STAMP(k) __asm__ volatile("; WGMARK " #k ::: "memory")
#elif defined(BSSM_WG_STAMPS)
#define WG_STAMP(k) do { if (stamp_on) { const long long now_ = (long long)__builtin_readcyclecounter(); st_acc[k] += now_ - st_last; st_last = now_; } } while (0)
#else
#define WG_STAMP(k)
#endif
constexpr int WG_NT_MAX = 512;  // 8 waves: 256 VGPRs per lane, no spills in the fp64 transcendental code
// Number of normals of the step draw of a filter variant as far as the stream format goes (pf_device.h): the linear-state
// kernels draw KP normals -- PACKED above two, their resampling uniforms then come from shared RNG_RESAMPLE blocks; the functor
// kernels (ssm_nlg, ssm_sde, EKF particle filter) keep WIDE blocks and the step block's spare word whatever their count.
template <int KP, int FN>
__device__ __forceinline__ constexpr int step_draw_count() { return FN == FN_NONE ? KP : 0; }

// Particle i = r * NT + tid (r < PPT): consecutive lanes own consecutive particles, so LDS
// accesses to the own slot are conflict-free and history stores are coalesced.
// BSSM_WG_NT: the kept history of the workgroup-resident filters (particle system and ancestor table, written once per step,
// read once by the back-trace / summary kernels after the filter) goes out with non-temporal stores
#ifndef BSSM_WG_NT
#define BSSM_WG_NT 1
#endif
// Waves per SIMD the 512-thread kernels are budgeted for.  Two particles per lane of a linear-state filter with at most four
// states came out at 132 VGPRs -- four over the 128 that let TWO workgroups share a CU (N <= 1024 needs 40 KB of LDS each).
// Eight-dimensional states, one particle per lane (N <= 512, 36 KB of LDS): 170 -> 128 VGPRs, four workgroups of 256 threads per
// CU instead of two: +28 % at N = 256; at two particles per lane the same budget spills and loses 7 % (BSSM_WG_OCC8 = 1).
// profiles/r04_wg_occupancy_ab.txt
#ifndef BSSM_WG_OCC8
#define BSSM_WG_OCC8 1
#endif
template <int MP, int PPT, int FN, int NTB>
constexpr int wg_waves_per_simd() {
  if (FN == FN_NONE && MP == 8 && NTB == WG_NT_MAX && PPT <= (BSSM_WG_OCC8)) return 4;
  return (FN == FN_NONE && MP <= 4 && NTB == WG_NT_MAX && PPT == 2) ? 4 : (NTB + 255) / 256;
}
// EX: the particle count is exactly NTB * PPT (4096 = 1024 x 4, the shape IS post-correction of config C5 runs): N is a
// compile-time constant, every `i < N` guard and the pivot / window arithmetic derived from N fold away
template <int MP, int KP, int PPT, int FN = FN_NONE, int NTB = WG_NT_MAX, int PM = -1, bool EX = false>
__device__ __forceinline__ void pf_wg_body(const FilterBatch& b) {
  constexpr int FID = FN > 0 ? FN : 0;
  const int psi_mode = PM >= 0 ? PM : b.psi;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  math_tab_init();
  // more than one particle per lane: the workgroup has exactly NTB threads (launch_wg_mk), so the thread count, the wave
  // count and every index derived from them are compile-time constants
  constexpr bool kFixedNT = PPT > 1;
  const int NT = kFixedNT ? NTB : (int)blockDim.x, tid = threadIdx.x, lane = tid & 63, nw = NT >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // psi-APF of a functor model: parameter block, observation row and approximating rows are wave-uniform -- scalar loads
  constexpr bool kPsiFn = FN != FN_NONE && PM == 1;
  const int Ncap = NT * PPT, N = EX ? NTB * PPT : b.N, n = b.n, f = blockIdx.x;
  double* __restrict__ s_alpha = smem;            // [MP][Ncap]
  double* __restrict__ s_cdf = smem + MP * Ncap;  // [Ncap]
  double* __restrict__ s_max = s_cdf + Ncap;      // [128] row maxima (row g = particles 64 g ... 64 g + 63, pf_canon.h)
  double* __restrict__ s_tot = s_max + CANON_MAX_ROWS;  // [128] row totals
  // T_t, R_t, C_t of the step for the linear-state transition.  Through scalar loads the 36 doubles of a four-dimensional state are
  // 72 SGPRs on top of the loop invariants of the time loop (130-150 SGPRs lived in VGPR lanes and came back through v_readlane).
  // Two remedies measured late in round 4 (same-box A/B, 512 ... 2048 particles, profiles/r04_wg_occupancy_ab.txt):
  //   BSSM_WG_RELOAD_MAT (default): the scalar loads are re-issued per particle (reload_table), so the matrix occupies the scalar
  //     file only inside one transition: +1.5 ... +4 %;
  //   BSSM_WG_LDS_MAT: staged in LDS by the first threads under the search, published by barrier (4), read back as broadcast
  //     ds_reads (what the step kernels do, pf_grid_v2.h `s_mat`): +5 ... +7 % at two / four particles per lane of a 512-thread
  //     workgroup, -18 % in the 1024-thread form, -35 % at one particle per lane, -20 % for eight-dimensional states (136 reads
  //     per particle through the one LDS pipe of the CU) -- not used.
#ifndef BSSM_WG_LDS_MAT
#define BSSM_WG_LDS_MAT 0
#endif
#ifndef BSSM_WG_LDS_MAT_FROM
#define BSSM_WG_LDS_MAT_FROM 99  // padded state dimension from which the LDS form is used: never -- measured slower overall (above)
#endif
#ifndef BSSM_WG_RELOAD_MAT
#define BSSM_WG_RELOAD_MAT 1
#endif
  constexpr int NMAT = MP * MP + MP * KP + MP;
  constexpr bool kLdsMat = FN == FN_NONE && ((BSSM_WG_LDS_MAT) != 0 || MP >= (BSSM_WG_LDS_MAT_FROM));
  double* __restrict__ s_mat = s_tot + CANON_MAX_ROWS;  // [NMAT]
  // Eight-dimensional states, linear transition, one or two particles per lane: the transition on the matrix cores through a
  // 4 KB operand tile per wave (propagate_mfma8_h, the step kernels' routine) -- the scalar-load form spent 41 % of its VALU
  // instructions on v_readlane of spilt dwordx16 tuples (profiles/r04_wg_occupancy_ab.txt).  The tile sits behind s_mat
  // (launch_wg_mk sizes the dynamic LDS for it); above 1024 slots a filter's LDS has no room for it.
#ifndef BSSM_WG_MFMA8
#define BSSM_WG_MFMA8 1
#endif
  constexpr bool kWgMfma = FN == FN_NONE && MP == 8 && PPT <= 2 && NTB <= WG_NT_MAX && (BSSM_WG_MFMA8) != 0;
  double* __restrict__ s_tile = s_mat + ((NMAT + 1) & ~1) + (kWgMfma ? wave * 512 : 0);
  // four particles per lane (1025 ... 2048 slots, 147 KB of state): 2 KB tiles inside the filter's CDF array, which nobody reads
  // between barrier (4) and the next step's barrier (1) (eight waves x 256 doubles = the array's 2048 entries)
  constexpr bool kWgMfmaQ = FN == FN_NONE && MP == 8 && PPT == 4 && NTB == WG_NT_MAX && (BSSM_WG_MFMA8) != 0;
  // sixteen-dimensional states, one particle per lane (N <= 512): 2 KB tiles behind s_mat (launch_wg_mk adds them)
  constexpr bool kWgMfma16 = FN == FN_NONE && MP == 16 && PPT == 1 && NTB == WG_NT_MAX && (BSSM_WG_MFMA8) != 0;
  double* __restrict__ s_tile16 = s_mat + ((NMAT + 1) & ~1) + (kWgMfma16 ? wave * 256 : 0);
  // two particles per lane (513 ... 1024 slots, 139 KB of state): half-wave tiles of 1 KB inside the CDF array (8 waves x 128 doubles)
  constexpr bool kWgMfma16H = FN == FN_NONE && MP == 16 && PPT == 2 && NTB == WG_NT_MAX && (BSSM_WG_MFMA8) != 0;
  // (these two stage their MFMA operands in the CDF array, which the row statistics of the next time point overwrite: barrier (0))
  constexpr bool kTileInCdf = kWgMfmaQ || kWgMfma16H;

  const RngKey key{b.seed, b.stream0 + (uint32_t)f};
  const double* __restrict__ mats = b.mats + (size_t)f * b.mats_fstride;
  const double* __restrict__ obs = b.obs + (size_t)f * b.obs_fstride;
  const double* __restrict__ ini = b.init + (size_t)f * b.init_fstride;
  const double* __restrict__ fpar = FN ? b.fpar + (size_t)f * b.fpar_fstride : nullptr;
  constexpr int ucnt = step_draw_count<KP, FN>();
  constexpr bool kPack = FN == FN_NONE;
  const double* __restrict__ amats = (FN && b.amats) ? b.amats + (size_t)f * b.amats_fstride : nullptr;
  const int obs_stride = ROW_HDR + b.p * (SER_HDR + MP), m = b.m;

  double w[PPT];  // log-weights of my particles at the current time
  double loglik = 0.0;
#ifdef BSSM_WG_STAMPS
  const bool stamp_on = blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == nw - 1);
  long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = (long long)__builtin_readcyclecounter();
#endif

  // ---- initial particles and their weights
#pragma unroll
  for (int r = 0; r < PPT; r++) {
    const int i = r * NT + tid;
    double x[MP];
    if (FN && !psi_mode) {
      if constexpr (FN != FN_NONE) functor_initial<MP, FID>(fpar, ini, key, (uint32_t)i, x);
    } else {
      double z[MP];
      rng_normals<MP, kPack>(key, RNG_INIT, 0, (uint32_t)i, z);
      initial_state<MP>(ini, z, x);
    }
    if (i < N) {
#pragma unroll
      for (int c = 0; c < MP; c++) {
        s_alpha[c * Ncap + i] = x[c];
        if (b.alpha_hist && c < m) b.alpha_hist[(((size_t)f * (n + 1)) * m + c) * N + i] = x[c];
      }
    }
    if constexpr (kPsiFn) w[r] = (i < N) ? functor_psi_weight<MP, FID>(as_uniform_table(fpar), as_uniform_table(obs), as_uniform_table(amats), 0, x, x) : -INFINITY;
    else w[r] = (i < N) ? weight_dispatch<MP, FN, PM>(b, fpar, obs, amats, 0, x, x) : -INFINITY;
    __builtin_amdgcn_sched_barrier(0);  // keep the unrolled particles sequential: bounds live registers
  }

  for (int t = 0; t <= n; t++) {
    WG_STAMP(7);
    // ---- normalise the time-t weights (src/model_ssm_ung.cpp:675-679) in the canonical row form (pf_canon.h): every row of 64
    // consecutive particles -- chunk q = r * nw + wave of this wave -- is scaled by ITS OWN maximum and scanned; the row-local
    // scans are what the search reads, so they are final here and ONE barrier publishes maxima, totals and scans (the rescaled
    // global CDF of rounds 1-4 needed a second pass and a second barrier)
    double e[PPT];
    if constexpr (kTileInCdf) lds_barrier();  // (0) the CDF array doubled as the waves' MFMA operand tiles during the transition
#pragma unroll
    for (int r = 0; r < PPT; r++) {
      const int i = r * NT + tid;
      const CanonRow rs = canon_row_stats(w[r]);  // (slots beyond N carry w = -inf)
      e[r] = rs.e;
      if (i < N) s_cdf[i] = rs.sc;
      if (lane == 63) s_tot[r * nw + wave] = rs.sc;
      if (lane == 0) s_max[r * nw + wave] = rs.M;
    }
    WG_STAMP(2);
    lds_barrier();  // (1)
    WG_STAMP(1);
    // every wave combines the <= 128 row partials redundantly (two per lane, in registers: saves a barrier)
    // (B = rows that hold particles: a function of N alone, whatever the thread geometry)
    const int B = EX ? (NTB * PPT) / CANON_ROW : (N + CANON_ROW - 1) / CANON_ROW;
    constexpr int kNP = PPT * NTB > 64 * CANON_ROW ? 2 : 1;  // the kernel's capacity exceeds 64 rows: a lane holds two row prefixes
    double pmx[kNP], psm[kNP];
#pragma unroll
    for (int k = 0; k < kNP; k++) { pmx[k] = lane + 64 * k < B ? s_max[lane + 64 * k] : -INFINITY; psm[k] = lane + 64 * k < B ? s_tot[lane + 64 * k] : 0.0; }
    const CanonComb<kNP> cc = canon_combine<kNP>(pmx, psm, B);
    const double S = cc.S, M = cc.M;
    if (b.weights) {
#pragma unroll
      for (int r = 0; r < PPT; r++) {
        const int i = r * NT + tid;
        const double sq = canon_row_scale(cc, r * nw + wave);
        if (i < N) b.weights[((size_t)f * (n + 1) + t) * N + i] = e[r] * sq;
      }
    }
    if (!(S > 0.0)) {  // :678-682 (uniform across the block)
      loglik = -INFINITY;
      if (tid == 0) b.dead_at[f] = t + 1;
      break;
    }
    const double c_scale = canon_stratum(S, N);
    if (wave == 0) loglik += canon_llinc(M, c_scale);       // :683 (thread 0 writes the log-likelihood)
    if (t == n) break;
    WG_STAMP(3);

    // ---- stratified resampling (stratified_sample.cpp:15-26) + ancestor gather (:657-661)
    if constexpr (kLdsMat) {  // (the previous step's transition has been read: every wave has passed barrier (1) since)
      const double* __restrict__ msrc = mats + (size_t)t * b.mats_tstride;
      for (int q = tid; q < NMAT; q += NT) s_mat[q] = msrc[q];
    }
    double a[PPT][MP];
    u32x4 blk[PPT];  // step blocks: spare word -> resampling uniform now, words 0-2 -> transition normals below
    // The 20 Philox round keys (seed + r * const, stream + r * const) are loop-invariant: hoisted out of the time loop they
    // sat in SGPRs for its whole body, were spilt to VGPR lanes and came back through 80 v_readlane per wave-step.  Opaque
    // copies of the key make them step-local: 20 scalar adds per step, live only while the step blocks are drawn.
    // (functor kernels only: in the linear-state kernels the same trick costs more scalar registers than it frees -- regime B of
    // bench.py, 512 filters of 2048 particles with four noise terms: 25.8 ms without it, 26.1 with it, and 30.6 with a second
    // opaque key in front of the further transition blocks: profiles/r03_regb_ab.txt)
    RngKey key_t = key;
    if constexpr (FN != FN_NONE) __asm__ volatile("" : "+s"(key_t.seed), "+s"(key_t.stream));
    // (a PACKED step draw at four particles per lane: the four lanes of a quad share their rows' RNG_RESAMPLE blocks)
    constexpr bool kQuadU = ucnt > 2 && PPT == 4;
    uint32_t uq[4] = {0u, 0u, 0u, 0u};
    if constexpr (kQuadU) rng_resample_words_quad4(key_t, (uint32_t)t, (uint32_t)NT, (uint32_t)tid, uq);
    // source row and target in the row's scale for every particle of the lane (pf_canon.h: the 64 slots of a wave-row are
    // consecutive, two ballots bound the source rows), then the bisections inside the rows: those of a lane's particles advance
    // together -- their LDS reads are independent, so PPT of them are in flight per step instead of one
    int lo[PPT], srow[PPT], slen[PPT];
    double vp[PPT];
#pragma unroll
    for (int r = 0; r < PPT; r++) {
      const int i = r * NT + tid;
      blk[r] = rng_step_block(key_t, (uint32_t)t, (uint32_t)i);
      const uint32_t uword = kQuadU ? uq[r & 3] : rng_resample_word<ucnt>(key_t, (uint32_t)t, (uint32_t)i, blk[r]);
      const CanonSource src = canon_source_row(canon_target(uword, i, c_scale), i < N, cc.px, lane);
      srow[r] = src.e * CANON_ROW;
      slen[r] = EX ? CANON_ROW : min(CANON_ROW, N - srow[r]);
      vp[r] = src.vp;
      lo[r] = 0;
    }
#pragma unroll
    for (int step = 32; step >= 1; step >>= 1) {
      double xm[PPT];
#pragma unroll
      for (int r = 0; r < PPT; r++) xm[r] = s_cdf[srow[r] + min(lo[r] + step - 1, slen[r] - 1)];
#pragma unroll
      for (int r = 0; r < PPT; r++) lo[r] += (lo[r] + step - 1 < slen[r] && xm[r] < vp[r]) ? step : 0;
    }
#pragma unroll
    for (int r = 0; r < PPT; r++) {
      const int i = r * NT + tid;
      lo[r] = srow[r] + min(lo[r], slen[r] - 1);
      if (i >= N) lo[r] = 0;
#pragma unroll
      for (int c = 0; c < MP; c++) a[r][c] = s_alpha[c * Ncap + lo[r]];
      if (b.indices && i < N) {
        if constexpr (BSSM_WG_NT) __builtin_nontemporal_store((uint32_t)lo[r], &b.indices[((size_t)f * n + t) * N + i]);
        else b.indices[((size_t)f * n + t) * N + i] = (uint32_t)lo[r];
      }
    }
    WG_STAMP(4);
    lds_barrier();  // (4) all ancestors read before the slots are overwritten
    WG_STAMP(1);

    // ---- transition to t+1 and the weights there
    const cdbl* mt = as_uniform_table(mats + (size_t)t * b.mats_tstride);
    const double* __restrict__ ob = obs + (size_t)(t + 1 < n ? t + 1 : 0) * obs_stride;
    const RngKey key_p = key;
#pragma unroll
    for (int r = 0; r < PPT; r++) {
      const int i = r * NT + tid;
      double x[MP];
      if (FN && !psi_mode) {
        if constexpr (FN != FN_NONE) functor_propagate<MP, KP, FID>(fpar, key_p, blk[r], (uint32_t)t, (uint32_t)i, a[r], x);
      } else if (FN && psi_mode == 2) {
        if constexpr (FN != FN_NONE) functor_ekpf_propagate<MP, FID>(fpar, ob, t + 1 >= n, key_p, blk[r], (uint32_t)t, (uint32_t)i, a[r], x);
      } else {
        double z[KP];
        rng_normals_blk<KP, kPack>(blk[r], key_p, RNG_TRANSITION, (uint32_t)t, (uint32_t)i, z);
        if constexpr (kWgMfma) {
          propagate_mfma8_h<KP>((const double*)(mats + (size_t)t * b.mats_tstride), a[r], z, x, s_tile, lane);
        } else if constexpr (kWgMfma16) {
          propagate_mfma16_q<KP, 64>((const double*)(mats + (size_t)t * b.mats_tstride), a[r], z, x, s_tile16, lane);
        } else if constexpr (kWgMfma16H) {
          propagate_mfma16_q<KP, 32>((const double*)(mats + (size_t)t * b.mats_tstride), a[r], z, x, s_cdf + wave * 128, lane);
        } else if constexpr (kWgMfmaQ) {
          propagate_mfma8_q<KP>((const double*)(mats + (size_t)t * b.mats_tstride), a[r], z, x, s_cdf + wave * 256, lane);
        } else if constexpr (kLdsMat) {
          int zero = 0;  // opaque: the matrix is re-read per particle instead of being kept live in 2 NMAT VGPRs
          __asm__ volatile("" : "+v"(zero));
          propagate<MP, KP>((const double*)(s_mat + zero), a[r], z, x);
        } else if constexpr ((BSSM_WG_RELOAD_MAT) != 0) {
          propagate<MP, KP>(reload_table(mt), a[r], z, x);  // scalar loads re-issued per particle: the matrix lives in SGPRs only here
        } else {
          propagate<MP, KP>(mt, a[r], z, x);
        }
      }
      if (i < N) {
#pragma unroll
        for (int c = 0; c < MP; c++) {
          s_alpha[c * Ncap + i] = x[c];
          if (b.alpha_hist && c < m) {
            if constexpr (BSSM_WG_NT) __builtin_nontemporal_store(x[c], &b.alpha_hist[(((size_t)f * (n + 1) + (t + 1)) * m + c) * N + i]);
            else b.alpha_hist[(((size_t)f * (n + 1) + (t + 1)) * m + c) * N + i] = x[c];
          }
        }
      }
      if constexpr (kPsiFn) w[r] = (i < N) ? (t + 1 < n ? functor_psi_weight<MP, FID>(as_uniform_table(fpar), as_uniform_table(ob), as_uniform_table(amats + (size_t)t * fn_amat_row<MP>()), t + 1, x, a[r]) : 0.0) : -INFINITY;
      else w[r] = (i < N) ? (t + 1 < n ? weight_dispatch<MP, FN, PM>(b, fpar, ob, (FN && psi_mode == 2) ? amats : amats + (size_t)t * fn_amat_row<MP>(), t + 1, x, a[r]) : 0.0) : -INFINITY;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#ifdef BSSM_WG_STAMPS
  if (stamp_on)
    printf("wg stamps wave %d: max %lld | barrier waits %lld | exp+scan %lld | totals+cdf %lld | search+gather %lld | propagate+weigh %lld cycles (n = %d)\n",
           wave, st_acc[0], st_acc[1], st_acc[2], st_acc[3], st_acc[4], st_acc[7], n);
#endif
  if (tid == 0 && b.loglik) b.loglik[f] = loglik;
}
template <int MP, int KP, int PPT, int FN = FN_NONE, int NTB = WG_NT_MAX, int PM = -1, bool EX = false>
__global__ void __launch_bounds__(NTB, (wg_waves_per_simd<MP, PPT, FN, NTB>())) pf_wg_kernel(const FilterBatch b) {
  pf_wg_body<MP, KP, PPT, FN, NTB, PM, EX>(b);
}
// LEAN: the same filter in 96 registers (five waves per SIMD budgeted, four used by the workgroup) -- the LAST round of a post-correction
// chunk runs it when the chunk has a tail: the register file then has room for the tail's row-family workgroups (64 registers per lane),
// and the tail steps BESIDE that round instead of behind it (api_postcorrect.cpp).  Same arithmetic, same bits; more spills, so the
// full rounds keep the 128-register kernel.
template <int MP, int KP, int PPT, int FN = FN_NONE, int NTB = WG_NT_MAX, int PM = -1, bool EX = false>
__global__ void __launch_bounds__(NTB, NTB / 256 + 1) pf_wg_kernel_lean(const FilterBatch b) {
  pf_wg_body<MP, KP, PPT, FN, NTB, PM, EX>(b);
}

#ifndef __HIPCC_RTC__   // host side: kernel selection and launch (the in-process compile of a user model takes the kernels only)
// the functor kernel of one shape for the batch's sampler (0 bootstrap, 1 psi-APF, 2 EKF particle filter: MP <= 2 only)
// the exact-size psi-APF variant: instantiated for the 1024-thread kernels at four particles per lane only
template <int MP, int PPT, int FNV, int NTB>
auto wg_fn_kernel_exact(int psi, bool lean = false) -> void (*)(const FilterBatch) {
  if constexpr (NTB == 1024 && PPT == 4 && MP <= 2) { if (psi == 1) return lean ? pf_wg_kernel_lean<MP, MP, PPT, FNV, NTB, 1, true> : pf_wg_kernel<MP, MP, PPT, FNV, NTB, 1, true>; }
  return nullptr;
}
// BSSM_FN_PSI_MASK: the samplers (bit 0 bootstrap, bit 1 psi-APF, bit 2 EKF particle filter) a translation unit instantiates
// functor kernels for -- a run-time compiled user model is built as three such parts side by side (user_model_tu.hip)
#ifndef BSSM_FN_PSI_MASK
#define BSSM_FN_PSI_MASK 7
#endif
template <int MP, int PPT, int FNV, int NTB>
auto wg_fn_kernel(int psi) -> void (*)(const FilterBatch) {
  if constexpr (((BSSM_FN_PSI_MASK) & 1) != 0) { if (psi == 0) return pf_wg_kernel<MP, MP, PPT, FNV, NTB, 0>; }
  if constexpr (((BSSM_FN_PSI_MASK) & 2) != 0) { if (psi == 1) return pf_wg_kernel<MP, MP, PPT, FNV, NTB, 1>; }
  if constexpr (((BSSM_FN_PSI_MASK) & 4) != 0) { if (psi == 2) return pf_wg_kernel<MP, MP, PPT, FNV, NTB, 2>; }
  return nullptr;
}
#define WG_FN_KERNEL(PPT_, NTB_) wg_fn_kernel<MP, PPT_, kFn, NTB_>(b.psi)
// the 96-register forms (pf_wg_kernel_lean): four particles per lane of the 1024-thread kernels -- the shapes a post-correction chunk with a
// tail runs as its last full round (api_postcorrect.cpp); at eight per lane the 128-register kernel already spills 116 ... 167 registers
template <int MP, int FNV>
auto wg_fn_kernel_lean4(int psi) -> void (*)(const FilterBatch) {
  if constexpr (((BSSM_FN_PSI_MASK) & 1) != 0) { if (psi == 0) return pf_wg_kernel_lean<MP, MP, 4, FNV, 1024, 0>; }
  if constexpr (((BSSM_FN_PSI_MASK) & 2) != 0) { if (psi == 1) return pf_wg_kernel_lean<MP, MP, 4, FNV, 1024, 1>; }
  if constexpr (((BSSM_FN_PSI_MASK) & 4) != 0) { if (psi == 2) return pf_wg_kernel_lean<MP, MP, 4, FNV, 1024, 2>; }
  return nullptr;
}
#ifndef BSSM_WG_WIDE_FROM
#define BSSM_WG_WIDE_FROM 1025   // (2048 until late round 4: N = 1536, m = 1 ran 512 x 4 at 144 VGPRs, one workgroup of two waves per SIMD per CU: +30 % as 1024 x 2)
#endif
template <int MP, int KP>
hipError_t launch_wg_mk(const FilterBatch& b, hipStream_t stream) {
  const int N = b.N;
  constexpr int kFn = fn_kernel_id<MP>();
  if (b.fpar && kFn > 0 && b.fn_id != kFn) return hipErrorInvalidValue;  // not the model these kernels were compiled for
  // 1024-thread workgroups (16 waves, 128 VGPRs per lane) for small states and large N: four waves per SIMD instead
  // of two hide the barrier and LDS latency of the one workgroup a CU can hold at these sizes
  bool wide = false;
  if constexpr (MP <= 2) {
    wide = N >= (BSSM_WG_WIDE_FROM);
    if (opt().wg_wide >= 0) wide = opt().wg_wide != 0 && N > 1024;
  }
  // four-dimensional states of the linear-state models (late round 4): above 1024 particles a filter is given 2048 slots, more than
  // half of a CU's LDS (5 x 2048 doubles + tables > 80 KB), so a CU holds ONE 512-thread workgroup = two waves per SIMD, and the barriers of a
  // step leave the VALU idle (regime B of bench.py: 0.60 of the issue roof).  The table-driven math of rounds 2-3 fits the
  // 128-register budget of a 1024-thread workgroup at two / four particles per lane without a spill
  if constexpr (MP == 4) {
    wide = N > 1024 && !b.fpar;
    if (opt().wg_wide >= 0) wide = opt().wg_wide != 0 && N > 1024 && !b.fpar;
  }
  const int ntmax = wide ? 1024 : WG_NT_MAX;
  int ppt = 1;
  while (ppt < 16 && N > ntmax * ppt) ppt *= 2;
  const int nt = ppt == 1 ? (N + 63) / 64 * 64 : ntmax;
  size_t lds = ((size_t)(MP + 1) * nt * ppt + 2 * CANON_MAX_ROWS + (MP * MP + MP * KP + MP)) * sizeof(double);
  if (MP == 8 && !b.fpar && !wide && ppt <= 2) lds += (1 + (size_t)(nt / 64) * 512) * sizeof(double);  // per-wave MFMA operand tiles (pf_wg_kernel, kWgMfma)
  if (MP == 16 && !b.fpar && !wide && ppt == 1) lds += (1 + (size_t)(nt / 64) * 256) * sizeof(double);  // (kWgMfma16)
  if (N > nt * ppt || lds > 160 * 1024) return hipErrorInvalidValue;
  void (*kern)(const FilterBatch) = nullptr;
  bool lean_used = false;
  (void)lean_used;
  if (wide) {
#ifndef BSSM_USER_MODEL
    if constexpr (MP == 4) {
      switch (ppt) {
        case 2: kern = pf_wg_kernel<MP, KP, 2, FN_NONE, 1024>; break;
        case 4: kern = pf_wg_kernel<MP, KP, 4, FN_NONE, 1024>; break;   // (no 96-register form: measured, four-dimensional states lose more to its spills than the tail gains -- profiles/r05_tail_concurrent.txt)
        default: return hipErrorInvalidValue;
      }
    }
#endif
    if constexpr (MP <= 2) {
      const bool fn = b.fpar != nullptr;
      if (fn && MP != KP) return hipErrorInvalidValue;
      switch (ppt) {
#ifndef BSSM_USER_MODEL
        case 2: kern = fn ? (MP == KP ? WG_FN_KERNEL(2, 1024) : nullptr) : pf_wg_kernel<MP, KP, 2, FN_NONE, 1024>; break;
        case 4: kern = fn ? (MP == KP ? WG_FN_KERNEL(4, 1024) : nullptr) : pf_wg_kernel<MP, KP, 4, FN_NONE, 1024>;
          if (b.lean) {
            if (!fn) { kern = pf_wg_kernel_lean<MP, KP, 4, FN_NONE, 1024>; lean_used = true; }
            else if constexpr (MP == KP) { if (auto kl = wg_fn_kernel_lean4<MP, kFn>(b.psi)) { kern = kl; lean_used = true; } }
          }
          if (fn && MP == KP && N == 4096 && !opt().wg_no_exact) { if (auto kx = wg_fn_kernel_exact<MP, 4, kFn, 1024>(b.psi, b.lean != 0)) { kern = kx; lean_used = b.lean != 0; } }
          break;
        default: kern = fn ? (MP == KP ? WG_FN_KERNEL(8, 1024) : nullptr) : pf_wg_kernel<MP, KP, 8, FN_NONE, 1024>; break;
#else
        case 2: kern = fn ? WG_FN_KERNEL(2, 1024) : nullptr; break;
        case 4: kern = fn ? WG_FN_KERNEL(4, 1024) : nullptr;
          if (fn && b.lean) { if (auto kl = wg_fn_kernel_lean4<MP, kFn>(b.psi)) { kern = kl; lean_used = true; } }
          break;
        default: kern = fn ? WG_FN_KERNEL(8, 1024) : nullptr; break;
#endif
      }
      if (ppt > 8 || !kern) return hipErrorInvalidValue;
    }
  } else {
#ifndef BSSM_USER_MODEL  // a user-model library only needs the functor (FN) variants: halves its compile time
  switch (ppt) {
    case 1:
      kern = pf_wg_kernel<MP, KP, 1>;
      // eight-dimensional states, up to 256 particles: the variant without the 128-register budget (the budget buys a second /
      // fourth workgroup per CU for 512-thread workgroups; small workgroups have the occupancy anyway and lose 5 ... 30 % to its spills)
      if constexpr (MP == 8) { if (nt <= 256) kern = pf_wg_kernel<MP, KP, 1, FN_NONE, 256>; }
      break;
    case 2: kern = pf_wg_kernel<MP, KP, 2>; break;
    case 4: kern = pf_wg_kernel<MP, KP, 4>; break;
    case 8: kern = pf_wg_kernel<MP, KP, 8>; break;
    default:
      if constexpr (MP <= 2) kern = pf_wg_kernel<MP, KP, 16>; else return hipErrorInvalidValue;
  }
#endif
  if (b.fpar) {  // built-in non-linear / SDE functors: instantiated for the shapes the shipped models have
    if constexpr (MP == KP && MP <= FN_MP_MAX) {
      switch (ppt) {
        case 1: kern = WG_FN_KERNEL(1, WG_NT_MAX); break;
        case 2: kern = WG_FN_KERNEL(2, WG_NT_MAX); break;
        case 4: kern = WG_FN_KERNEL(4, WG_NT_MAX); break;
        case 8: kern = WG_FN_KERNEL(8, WG_NT_MAX); break;
        default:
          if constexpr (MP <= 2) kern = WG_FN_KERNEL(16, WG_NT_MAX); else return hipErrorInvalidValue;
      }
    } else {
      return hipErrorInvalidValue;
    }
  }
  }
  if (!kern) return hipErrorInvalidValue;
#ifndef BSSM_USER_MODEL
  { char nm[96]; std::snprintf(nm, sizeof nm, "pf_wg_kernel<%d,%d,%d,%s,%d>", MP, KP, ppt, b.fpar ? (kFn > 0 ? "model" : "any") : "false", wide ? 1024 : WG_NT_MAX); if (lean_used) std::strncat(nm, " lean", sizeof nm - std::strlen(nm) - 1); note_filter_kernel(nm); }
#endif
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  if ((e = hipMemsetAsync(b.dead_at, 0, sizeof(int) * b.F, stream)) != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(b.F), dim3(nt), lds, stream, b);
  return hipGetLastError();
}

#endif  // !__HIPCC_RTC__

// ------------------------------------------------------------------------------------------
// Multi-workgroup filter: one launch per time step
// ------------------------------------------------------------------------------------------
constexpr int GRID_NT = 256;
// (the step kernel itself: pf_grid_v2.h; the round-1 kernel that lived here until round 5 -- seven workgroup barriers, one ancestor search
// per particle -- is gone: every model class steps through pf_grid_step_v2 / pf_grid_step_v2_fn or the row family, pf_row_step.h)

}  // namespace bssm
#include "pf_grid_v2.h"
#include "pf_row_step.h"
#ifndef __HIPCC_RTC__
#ifndef BSSM_USER_MODEL
#include "pf_big.h"
#endif
namespace bssm {

template <int MP, int KP>
hipError_t launch_grid_mk(const FilterBatch& b, const GridWorkspace& ws, hipStream_t stream, StepTiming* timing) {
  if (ws.chunk == CANON_ROW) {  // the row family (pf_row_step.h): a workspace carved by row_workspace_carve
#ifndef BSSM_USER_MODEL
    // EXPERIMENT (round 5, BSSM_AMD_FORCE_PATH=row): an eight-dimensional linear state through the fragment-native matrix-core
    // transition of pf_big.h -- one 16-row block, half of it padding -- instead of the step kernel's LDS transposes
    if constexpr (MP == 8 && KP % 4 == 0) { if (!b.fpar) return launch_big_row<1, 8>(b, ws, stream, timing); }
#endif
    return launch_row_mk<MP, KP>(b, ws, stream, timing);
  }
  const int ppt = ws.chunk / GRID_NT;
  constexpr int kFn = fn_kernel_id<MP>();
  if (b.fpar && kFn > 0 && b.fn_id != kFn) return hipErrorInvalidValue;
  // ONE step-kernel generation since round 5 (pf_grid_v2.h: one ancestor search per wave, weights exchange, wave-local combine) for
  // every model class: at most 256 particle blocks of 256 * {1, 2, 4} particles per filter = 262 144 particles (the reference's
  // front-end stops at 1e5).  The round-1 kernel (seven workgroup barriers, per-particle search) is gone.
  if (ws.B > V2_MAX_BLOCKS || (ppt != 1 && ppt != 2 && ppt != 4)) return hipErrorInvalidValue;
  void (*kern)(const FilterBatch, const GridWorkspace, const int) = nullptr;
  void (*kern_init)(const FilterBatch, const GridWorkspace) = nullptr;  // the step t = -1 has its own kernel
#ifdef BSSM_XCD_PERSIST
  void (*kern_persist)(const FilterBatch*, const GridWorkspace*, const PersistCtl, const int, const int) = nullptr;  // XCD-resident time loop (experiment builds)
#define BSSM_PERSIST_PICK(...) kern_persist = __VA_ARGS__
#else
#define BSSM_PERSIST_PICK(...) ((void)0)
#endif
  // weights exchange (pf_grid_v2.h): with the weights output on, blocks hand over their block-scaled weights instead of
  // the scan of them -- one array less written per particle-step
  bool v2ms = false, v2exact = false, v2fn = false;
  (void)v2fn; (void)v2exact;
  bool v2wx = b.weights != nullptr;
  if (opt().grid_v2_wx >= 0) v2wx = v2wx && opt().grid_v2_wx != 0;
  (void)v2ms;
  if (b.fpar) {
    // functor models (ssm_nlg / ssm_sde: the shipped ones have one or two states, a run-time compiled user model up to 16)
    if constexpr (MP == KP && MP <= FN_MP_MAX) {
      switch (ppt) {
        case 1: kern = v2_fn_kernels<MP, 1, kFn>(b.psi, v2wx, &kern_init); break;
        case 2: kern = v2_fn_kernels<MP, 2, kFn>(b.psi, v2wx, &kern_init); break;
        default: kern = v2_fn_kernels<MP, 4, kFn>(b.psi, v2wx, &kern_init); break;
      }
      v2fn = true;
    } else {
      return hipErrorInvalidValue;
    }
  }
#ifndef BSSM_USER_MODEL
  else {
    if constexpr (MP <= 16) {
      v2exact = b.N % ws.chunk == 0 && b.m == MP;  // complete blocks, no padding components: the variants without tail guards
      if (opt().grid_v2_exact >= 0) v2exact = v2exact && opt().grid_v2_exact != 0;
      if constexpr (MP <= 4) {
        if (b.p == 1) {
          switch (ppt) {
            case 1: kern = v2_step_kernel<MP, KP, 1, false>(v2wx, v2exact); kern_init = pf_grid_init_v2<MP, 1>; BSSM_PERSIST_PICK(v2_persist_kernel<MP, KP, 1, false>(v2wx, v2exact)); break;
            case 2: kern = v2_step_kernel<MP, KP, 2, false>(v2wx, v2exact); kern_init = pf_grid_init_v2<MP, 2>; BSSM_PERSIST_PICK(v2_persist_kernel<MP, KP, 2, false>(v2wx, v2exact)); break;
            default: kern = v2_step_kernel<MP, KP, 4, false>(v2wx, v2exact); kern_init = pf_grid_init_v2<MP, 4>; BSSM_PERSIST_PICK(v2_persist_kernel<MP, KP, 4, false>(v2wx, v2exact)); break;
          }
        }
      }
      // rows of several observation series (ssm_mng) and every state of 8 or 16 dimensions: the multi-series variants (one or two
      // particles per lane of an 8-dimensional state and every 16-dimensional one: fp64 MFMA transition)
      if (!kern_init) {
        v2ms = true;
        if (ppt == 1) { kern = v2_step_kernel<MP, KP, 1, true>(v2wx, v2exact); kern_init = pf_grid_init_v2<MP, 1, true>; if constexpr (MP <= 8) BSSM_PERSIST_PICK(v2_persist_kernel<MP, KP, 1, true>(v2wx, v2exact)); }
        else if (ppt == 2) { kern = v2_step_kernel<MP, KP, 2, true>(v2wx, v2exact); kern_init = pf_grid_init_v2<MP, 2, true>; if constexpr (MP <= 8) BSSM_PERSIST_PICK(v2_persist_kernel<MP, KP, 2, true>(v2wx, v2exact)); }
        else { kern = v2_step_kernel<MP, KP, 4, true>(v2wx, v2exact); kern_init = pf_grid_init_v2<MP, 4, true>; }
      }
    }
  }
#endif
  if (!kern || !kern_init) return hipErrorInvalidValue;
#ifndef BSSM_USER_MODEL
  {
    char nm[96];
    if (v2fn) std::snprintf(nm, sizeof nm, "pf_grid_step_v2_fn<%d,%d,%s,%s,%d>", MP, ppt, v2wx ? "true" : "false", kFn > 0 ? "model" : "any", b.psi);
    else std::snprintf(nm, sizeof nm, "pf_grid_step_v2<%d,%d,%d,%s,%s,%s>", MP, KP, ppt, v2ms ? "true" : "false", v2wx ? "true" : "false", v2exact ? "true" : "false");
    note_filter_kernel(nm);
  }
#endif
  hipError_t e;
  if ((e = hipMemsetAsync(b.dead_at, 0, sizeof(int) * b.F, stream)) != hipSuccess) return e;
  if ((e = hipMemsetAsync(ws.llinc, 0, sizeof(double) * b.F * (b.n + 1), stream)) != hipSuccess) return e;
  // A launch cannot start before every workgroup of the previous step has finished, so each step ends with SIMDs idling
  // behind the slowest waves.  A batch of at least 512 workgroups per step is therefore stepped as TWO interleaved launch
  // sequences (the two halves of its filters, independent of each other) on two streams: the tail of one half's step
  // runs under the body of the other's (measured +13 % particle-steps/s at 24 filters of 65 536 particles, m = 4; +19 % at
  // m = 1; +25 % at 12 filters; below ~400 workgroups the second sequence costs more than it hides).
  // (with two or more particles per lane a workgroup carries twice the work: from 384 workgroups -- three 65 536-particle
  // filters at two per lane 1.80e10 -> 1.92e10, m = 1 2.12e10 -> 2.21e10; two filters, 256 workgroups: 1.41e10 -> 1.33e10, stays one)
  int parts = ((long long)b.F * ws.B >= (ws.chunk >= 2 * GRID_NT ? 384 : 512) && b.F >= 2) ? 2 : 1;
  constexpr int kMaxParts = 4;  // (more than two sequences: measurement option, BSSM_AMD_GRID_STREAMS=3|4 -- extra side streams per host thread)
  if (opt().grid_streams > 0) parts = std::max(1, std::min({opt().grid_streams, kMaxParts, b.F}));
  const dim3 block(GRID_NT);
  if (!ws.side) parts = 1;
  hipStream_t lanes[kMaxParts] = {stream, stream, stream, stream};
  hipEvent_t fork = nullptr, join = nullptr;
  GridWorkspace wsp[kMaxParts] = {ws, ws, ws, ws};
  const int fw = (v2wx && kern_init && ws.fw && b.weights) ? 1 : 0;  // factored weights output: only with the weights-exchange kernels
  for (int q = 0; q < kMaxParts; q++) { wsp[q].wx = v2wx ? 1 : 0; wsp[q].fw = fw; }
  if (timing) timing->factored = fw;
  if (fw && (e = hipMemsetAsync(ws.wscale, 0, sizeof(double) * b.F * (b.n + 1) * ws.B, stream)) != hipSuccess) return e;  // (columns behind a degenerate step stay zero)
  int fcount[kMaxParts] = {b.F, 0, 0, 0};
  if (parts >= 2) {  // first part on the caller's stream, the others on the library's side streams
    static thread_local TlsStream extra_side[kMaxParts - 2];
    lanes[1] = ws.side;
    for (int q = 2; q < parts; q++) { lanes[q] = extra_side[q - 2].get(); if (!lanes[q]) { parts = q; break; } }
    int f_at = 0;
    for (int q = 0; q < parts; q++) { fcount[q] = (b.F * (q + 1)) / parts - (b.F * q) / parts; wsp[q].f0 = f_at; f_at += fcount[q]; }
    if ((e = hipEventCreateWithFlags(&fork, hipEventDisableTiming)) != hipSuccess) return e;
    (void)hipEventRecord(fork, stream);
    for (int q = 1; q < parts; q++) (void)hipStreamWaitEvent(lanes[q], fork, 0);
  }
  if (opt().xcd_remap >= 0) { for (int q = 0; q < kMaxParts; q++) wsp[q].xcd_remap = opt().xcd_remap; }
#if !defined(BSSM_USER_MODEL) && defined(BSSM_XCD_PERSIST)
  {  // experiment builds only (experiments/xcd_persist.h): ONE launch for the whole time loop of every filter of the batch
    char pname[112];
    std::snprintf(pname, sizeof pname, "pf_xcd_persist_v2<%d,%d,%d,%s,%s,%s>", MP, KP, ppt, v2ms ? "true" : "false", v2wx ? "true" : "false", v2exact ? "true" : "false");
    hipError_t pe = hipSuccess;
    if (xcd_persist_try(b, ws, wsp[0], stream, timing, kern_persist, kern_init, fw, pname, &pe)) return pe;
  }
#endif
  if (timing) timing->concurrent = parts;
  auto step = [&](int q, int t) {
    if (t < 0 && kern_init) hipLaunchKernelGGL(kern_init, dim3(ws.B, fcount[q]), block, 0, lanes[q], b, wsp[q]);
    else hipLaunchKernelGGL(kern, dim3(ws.B, fcount[q]), block, 0, lanes[q], b, wsp[q], t);
  };
  if (!timing || timing->every <= 0) {
    const int pe = (timing && parts == 1) ? timing->progress_every : 0;
    for (int t = -1; t < b.n; t++) {
      for (int q = 0; q < parts; q++) step(q, t);
      // launch t completes time point t + 1 of the history and time point t of weights / ancestors
      if (pe > 0 && t >= 0 && (t + 1) % pe == 0 && t + 1 < b.n && timing->progress_count < StepTiming::kMaxProgress) {
        hipEvent_t ev;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) {
          (void)hipEventRecord(ev, stream);
          timing->progress_events[timing->progress_count] = ev;
          timing->progress_steps[timing->progress_count++] = t + 1;
        }
      }
    }
  } else {
    // windows of kWin consecutive step launches (of the first sequence) bracketed by one event pair each: the event
    // overhead (a few us) is amortised over the window, so total/count is the per-launch kernel cadence
    constexpr int kMaxSamples = 64, kWin = 8;
    hipEvent_t ev[2 * kMaxSamples];
    int ns = 0;
    const int period = timing->every > kWin ? timing->every : kWin + 1;
    int open_t = -1;  // first launch of the window being timed, -1 = none
    for (int t = -1; t < b.n; t++) {
      if (open_t < 0 && t >= 0 && t % period == 0 && t + kWin <= b.n && ns < kMaxSamples) {
        (void)hipEventCreate(&ev[2 * ns]);
        (void)hipEventCreate(&ev[2 * ns + 1]);
        (void)hipEventRecord(ev[2 * ns], lanes[0]);
        open_t = t;
      }
      for (int q = 0; q < parts; q++) step(q, t);
      if (open_t >= 0 && t == open_t + kWin - 1) {
        (void)hipEventRecord(ev[2 * ns + 1], lanes[0]);
        ns++;
        open_t = -1;
      }
    }
    for (int q = 0; q < parts; q++)
      if ((e = hipStreamSynchronize(lanes[q])) != hipSuccess) return e;
    for (int i = 0; i < ns; i++) {
      float ms = 0;
      (void)hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]);
      timing->total_ms += ms;
      timing->count += kWin;
      (void)hipEventDestroy(ev[2 * i]);
      (void)hipEventDestroy(ev[2 * i + 1]);
    }
  }
  if (parts >= 2) {  // join: the caller's stream continues after the side sequences too
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


// per-MP entry points, one translation unit each (pf_inst_mp*.hip) so the build parallelises
template <int MP> hipError_t launch_wg_mp(const FilterBatch& b, hipStream_t stream);
template <int MP> hipError_t launch_grid_mp(const FilterBatch& b, const GridWorkspace& ws, hipStream_t stream, StepTiming* timing);

}  // namespace bssm
#endif  // !__HIPCC_RTC__
