// lg_smoother_wave.h -- the Kalman filter + covariance smoother of lg_smoother.h (LgSmoother<M>::run without the conditional factors)
// for states of 17 ... 64 dimensions (padded to M = 32 or 64): ONE WORKGROUP of NW waves per 64 / M draws, lane r of every wave owns ROW r,
// wave w the columns [w M / NW, (w + 1) M / NW) of it, and every M x M matrix of the recursion lives in LDS (rows padded to M + 1 doubles: a
// column read is as conflict-free as a row read).
//
// Why not lg_smoother_rows.h: its rows live in registers with every loop unrolled -- seven row arrays of 32 doubles no longer fit a lane's 512
// registers, a 64-dimensional state not by far, and a completely unrolled 32 x 32 product is 2.5 minutes of compilation per kernel.  Here a
// product C = A B keeps only the lane's part of the OUTPUT row in registers: c[j] += A(r, q) * B(q, j) for q = 0, 1, ... (the q loop is a real
// loop, the j loop unrolled) -- A(r, q) is the row's own operand (from its LDS row), B(q, j) a broadcast read that is the same address in every
// lane of the group.  Four matrix slots per group (A B C D) carry a step.  The matrices of a 64-dimensional state fill the CU's LDS (136 KB),
// so one draw is all a CU holds: its four waves (NW = M / 16, one per SIMD) split the columns, each an independent chain of multiply-adds.
//
// The arithmetic follows LgSmoother<M>::run term by term -- the same products in the same association, every sum over the inner index in
// ascending order from zero -- with the two differences lg_smoother_rows.h has as well: a step without an update runs the update formulas
// with K = 0 (products with exact zeros and ones), and a non-finite value is carried to the end and reported there.
// Series at a time (LgModel::sub = p > 1, round 6: ssm_mng draws of 17 ... 64 states): the n p virtual steps run through the SAME formulas, a step
// inside a time point with T = I, R R' = 0, C = 0 in place of the model's (products with exact ones and zeros: the values LgSmoother's
// special-cased branch has).  Device only; blockDim.x = 64 NW; every lane of the workgroup must call it.
#pragma once
#include "lg_smoother.h"
#include "pf_device.h"   // wave_sum (run_mfma)

namespace bssm {

// analysis builds (-DLGW_STAMPS, scripts/lgwave_check.hip): clocks (s_memtime, 100 MHz) thread 0 of workgroup 0 spends between the marks of run_mfma
#ifdef LGW_STAMPS
__device__ unsigned long long lgw_stamps[32];
#define LGW_T(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) { const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); lgw_acc[k] += t1_ - lgw_t0; lgw_t0 = t1_; } } while (0)
#else
#define LGW_T(k)
#endif

#ifndef LGW_NW32
#define LGW_NW32 4   // waves per workgroup at 32 (padded) states: four (eight columns each, 222 registers: two workgroups = two waves per SIMD; 512 draws 10.2 -> 5.2 ms against two)
#endif
// (M = 16, late round 6: states of 9 ... 16 dimensions -- monthly seasonals -- four draws per workgroup, a wave = one draw's 16 x 16 products, four matrix
// instructions each: the rows-in-registers form of lg_smoother_rows.h took 127 us per time point, every multiply-add behind a cross-lane fetch)
template <int M, int NW = (M == 64 ? 4 : LGW_NW32)>
struct LgWave {
  static_assert(M == 16 || M == 32 || M == 64, "M lanes per draw must tile the wave");
  static_assert(M % NW == 0, "the waves split the columns evenly");
  static constexpr int MM = M * M, G = 64 / M, LD = M + 1, MAT = M * LD, JW = M / NW;
  static constexpr int GS = 4 * MAT + 6 * M + 2;  // per group: four matrices, six vectors (+ 2: groups start in different banks)
  static constexpr size_t kLdsDoubles = (size_t)G * GS;
  static constexpr int kThreads = 64 * NW;
  static constexpr double kZeroTol = 1e-12;

  double *SA, *SB, *SC, *SD;          // this group's matrix slots
  double *vZ, *va, *vK, *vx, *vy, *vr;  // this group's vectors
  int r;                              // the row this lane works on
  int j0;                             // the first of the JW columns this lane owns
  int w;                              // its wave
  int mr = M;                         // the states the draws really have (the rest of the padded dimension is exact zeros): the inner loops stop there
  __device__ LgWave(double* lds_wg) {
    const int lane = threadIdx.x & 63;
    w = (int)threadIdx.x >> 6;
    r = lane % M;
    j0 = w * JW;
    double* base = lds_wg + (size_t)(lane / M) * GS;
    SA = base; SB = base + MAT; SC = base + 2 * MAT; SD = base + 3 * MAT;
    vZ = base + 4 * MAT; va = vZ + M; vK = va + M; vx = vK + M; vy = vx + M; vr = vy + M;
  }
  __device__ LG_INL static void sync() { __syncthreads(); }
  // a barrier that orders the workgroup's LDS traffic only: loads from / stores to memory stay in flight across it (run_mfma: the operands of the
  // next step are requested a step ahead; nothing a wave of the workgroup stores to memory inside a pass is read by another wave inside that pass)
  __device__ LG_INL static void sync_lds() { __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

  // c[jj] = sum over q (ascending, from 0.0) of own(q) * bop(q, j0 + jj)
  template <class Own, class Bop>
  __device__ LG_INL void prod(Own own, Bop bop, double (&c)[JW]) const {
#pragma unroll
    for (int jj = 0; jj < JW; jj++) c[jj] = 0.0;
#pragma unroll 2
    for (int q = 0; q < M; q++) {
      const double a = own(q);
#pragma unroll
      for (int jj = 0; jj < JW; jj++) c[jj] += a * bop(q, j0 + jj);
    }
  }
  __device__ LG_INL void put_row(double* S, const double (&c)[JW]) const {
#pragma unroll
    for (int jj = 0; jj < JW; jj++) S[r * LD + j0 + jj] = c[jj];
  }
  // this lane's columns of row r of a column-major M x M matrix in memory
  __device__ LG_INL void load_row(const double* p, size_t es, double (&c)[JW]) const {
#pragma unroll
    for (int jj = 0; jj < JW; jj++) c[jj] = p[(size_t)(r + M * (j0 + jj)) * es];
  }
  __device__ LG_INL void store_row(double* p, size_t es, const double (&c)[JW]) const {
#pragma unroll
    for (int jj = 0; jj < JW; jj++) p[(size_t)(r + M * (j0 + jj)) * es] = c[jj];
  }

  // at [(n+1)][M], Pt [(n+1)][MM], cc [(n+1)][MM], K [n][M], v [n], Fs [n]: element k at ptr[k * es] (draw-interleaved), as LgSmoother::run
  // leaves them with with_factors = false.  Returns the draw's verdict (the same in every lane of a group).
  // (round 6) `em`: the element stride of the MATRIX work arrays at / Pt / cc / K -- es (draw-interleaved, as the model inputs and v / Fs always are), or
  // 1 for a draw-major layout: a 64-dimensional draw's row-distributed stores are then 512-byte segments instead of 8 bytes per 2 KB stride
  __device__ LG_INL bool run(const LgModel& g, double* at, double* Pt, double* cc, double* K, double* v, double* Fs, size_t es, size_t em = 0) {
    if (em == 0) em = es;
#ifndef LGW_NO_MFMA
    if constexpr ((M == 64 || M == 32 || M == 16) && NW == 4) return run_mfma(g, at, Pt, cc, K, v, Fs, es, em);
#endif
    const int n = g.n, sub = g.sub;
    int bad = 0;
    double c[JW];
    const double a1r = g.a1[(size_t)r * es];
    load_row(g.P1, es, c);
    sync();
    va[r] = a1r;
    put_row(SA, c);
    if (w == 0) at[(size_t)r * em] = a1r;
    store_row(Pt, em, c);
    // ---- forward pass (kf_predict, univariate branch, Joseph form): SA = P, SD = T
    bool have_T = false;
    for (int t = 0; t < n; t++) {
      const int tr = sub == 1 ? t : t / sub, js = sub == 1 ? 0 : t - tr * sub;
      const bool moves = sub == 1 || js == sub - 1;   // (a step inside a time point: identity transition, no noise)
      const double zr = g.Z[((size_t)tr * g.Zs + (size_t)js * M + r) * es];
      const double yt = g.y[(size_t)t * es], sd = g.H[(size_t)t * g.Hs * es], Dt = g.D[((size_t)tr * g.Ds + js) * es];
      const double Cr = moves ? g.C[((size_t)tr * g.Cs + r) * es] : 0.0;
      const bool load_T = !have_T || g.Ts != 0 || sub != 1;
      if (load_T) {
        if (moves) load_row(g.T + (size_t)tr * g.Ts * es, es, c);
        else {
#pragma unroll
          for (int jj = 0; jj < JW; jj++) c[jj] = (r == j0 + jj) ? 1.0 : 0.0;
        }
      }
      sync();
      vZ[r] = zr;
      if (load_T) { put_row(SD, c); have_T = true; }
      sync();
      const bool na = !lg_finite(yt);
      const double HH = na ? 1.0 : sd * sd;
      // (P Z)_r: P is symmetric to the last bit (symmatu below), so row r times Z is column r times Z
      double pz = 0.0;
#pragma unroll 4
      for (int q = 0; q < M; q++) pz += vZ[q] * SA[r * LD + q];
      vx[r] = pz;
      sync();
      double F = 0.0;
      if (!na) {
#pragma unroll 4
        for (int q = 0; q < M; q++) F += vx[q] * vZ[q];
      }
      F += HH;
      const bool upd = !na && F > kZeroTol;
      double za = 0.0;
#pragma unroll 4
      for (int q = 0; q < M; q++) za += vZ[q] * va[q];
      const double vt = upd ? yt - Dt - za : 0.0;
      if (r == 0 && w == 0) { Fs[(size_t)t * es] = upd ? F : 0.0; v[(size_t)t * es] = vt; }
      const double Kr = upd ? pz / F : 0.0;
      const double aur = upd ? va[r] + Kr * vt : va[r];
      if (w == 0) K[((size_t)t * M + r) * em] = Kr;
      vK[r] = Kr;
      vy[r] = aur;
      // I - K Z' into SB
#pragma unroll
      for (int jj = 0; jj < JW; jj++) SB[r * LD + j0 + jj] = ((r == j0 + jj) ? 1.0 : 0.0) - Kr * vZ[j0 + jj];
      sync();
      double s = 0.0;
#pragma unroll 4
      for (int q = 0; q < M; q++) s += SD[r * LD + q] * vy[q];
      const double a1n = Cr + s;
      // Joseph form: inner = (I - K Z') P (I - K Z')' + K HH K'
      prod([&](int q) { return SB[r * LD + q]; }, [&](int q, int j) { return SA[q * LD + j]; }, c);
      put_row(SC, c);
      sync();           // the row is complete (the waves each wrote their columns)
      prod([&](int q) { return SC[r * LD + q]; }, [&](int q, int j) { return SB[j * LD + q]; }, c);
#pragma unroll
      for (int jj = 0; jj < JW; jj++) c[jj] = upd ? c[jj] + Kr * HH * vK[j0 + jj] : c[jj];
      sync();           // every lane is through with P and with the first product's rows
      put_row(SA, c);   // inner
      sync();
      // T inner T' + R R'
      prod([&](int q) { return SD[r * LD + q]; }, [&](int q, int j) { return SA[q * LD + j]; }, c);
      put_row(SC, c);
      sync();
      prod([&](int q) { return SC[r * LD + q]; }, [&](int q, int j) { return SD[j * LD + q]; }, c);
#pragma unroll
      for (int jj = 0; jj < JW; jj++) c[jj] = c[jj] + (moves ? g.RR[((size_t)tr * g.Rs + r + M * (j0 + jj)) * es] : 0.0);
      put_row(SA, c);   // (every lane was through with inner before the barrier above)
      sync();
      // symmatu: the lower triangle takes the upper one's values
#pragma unroll
      for (int jj = 0; jj < JW; jj++) if (j0 + jj < r) c[jj] = SA[(j0 + jj) * LD + r];
#pragma unroll
      for (int jj = 0; jj < JW; jj++) if (j0 + jj < r) SA[r * LD + j0 + jj] = c[jj];
      bad |= !lg_finite(a1n);
#pragma unroll
      for (int jj = 0; jj < JW; jj++) bad |= !lg_finite(c[jj]);
      if (w == 0) at[((size_t)(t + 1) * M + r) * em] = a1n;
      store_row(Pt + (size_t)(t + 1) * MM * em, em, c);
      store_row(cc + (size_t)t * MM * em, em, c);
      va[r] = a1n;
    }
    // ---- backward pass (smoother_ccov): SA = N, vr = r_t; T, P and the cross-covariance are staged from memory row by row
    sync();
    vr[r] = 0.0;
#pragma unroll
    for (int jj = 0; jj < JW; jj++) SA[r * LD + j0 + jj] = 0.0;
    __threadfence_block();
    sync();   // (also: the K, F and v stores of the forward pass are visible to the other waves of the workgroup)
    for (int t = n - 1; t >= 0; t--) {
      const double F = Fs[(size_t)t * es], vt = v[(size_t)t * es];
      const bool upd = F > 0.0;
      const int tr = sub == 1 ? t : t / sub, js = sub == 1 ? 0 : t - tr * sub;
      const bool moves = sub == 1 || js == sub - 1;
      const double zr = g.Z[((size_t)tr * g.Zs + (size_t)js * M + r) * es];
      const double Kr = upd ? K[((size_t)t * M + r) * em] : 0.0;
      const double ar = at[((size_t)t * M + r) * em];
      // L = T (I - K Z'): T row into SD, I - K Z' into SB, L into SC
      if (moves) load_row(g.T + (size_t)tr * g.Ts * es, es, c);
      else {
#pragma unroll
        for (int jj = 0; jj < JW; jj++) c[jj] = (r == j0 + jj) ? 1.0 : 0.0;
      }
      sync();
      vZ[r] = zr;
      vK[r] = Kr;
      put_row(SD, c);
      sync();
#pragma unroll
      for (int jj = 0; jj < JW; jj++) SB[r * LD + j0 + jj] = ((r == j0 + jj) ? 1.0 : 0.0) - Kr * vZ[j0 + jj];
      sync();
      prod([&](int q) { return SD[r * LD + q]; }, [&](int q, int j) { return SB[q * LD + j]; }, c);
      put_row(SC, c);
      // cc[t] = Pt[t] L' (I - N cc[t]): cc[t] into SB, then I - N cc[t] into SB; Pt[t] row into SD, then (Pt[t] L') row into SD
      load_row(cc + (size_t)t * MM * em, em, c);
      sync();           // every lane is through with I - K Z' and the T row; L is complete
      put_row(SB, c);
      sync();
      prod([&](int q) { return SA[r * LD + q]; }, [&](int q, int j) { return SB[q * LD + j]; }, c);
#pragma unroll
      for (int jj = 0; jj < JW; jj++) c[jj] = ((r == j0 + jj) ? 1.0 : 0.0) - c[jj];
      sync();
      put_row(SB, c);
      load_row(Pt + (size_t)t * MM * em, em, c);
      put_row(SD, c);
      sync();
      prod([&](int q) { return SD[r * LD + q]; }, [&](int q, int j) { return SC[j * LD + q]; }, c);
      sync();           // every lane is through with the Pt[t] row
      put_row(SD, c);
      sync();
      prod([&](int q) { return SD[r * LD + q]; }, [&](int q, int j) { return SB[q * LD + j]; }, c);
#pragma unroll
      for (int jj = 0; jj < JW; jj++) bad |= !lg_finite(c[jj]);
      store_row(cc + (size_t)t * MM * em, em, c);
      // r_{t-1} = L' r_t, N_{t-1} = L' N L (column r of L is the row's operand)
      double rn = 0.0;
#pragma unroll 4
      for (int q = 0; q < M; q++) rn += SC[q * LD + r] * vr[q];
      prod([&](int q) { return SC[q * LD + r]; }, [&](int q, int j) { return SA[q * LD + j]; }, c);
      sync();           // every lane is through with the (Pt[t] L') row
      put_row(SD, c);
      sync();
      prod([&](int q) { return SD[r * LD + q]; }, [&](int q, int j) { return SC[q * LD + j]; }, c);
      if (upd) {
        rn = zr / F * vt + rn;
#pragma unroll
        for (int jj = 0; jj < JW; jj++) c[jj] = zr * vZ[j0 + jj] / F + c[jj];
      }
      sync();           // every lane is through with N, r_t and I - N cc[t]
      vr[r] = rn;
      put_row(SA, c);
      // Pt[t] row into SB (it is a broadcast operand below)
      load_row(Pt + (size_t)t * MM * em, em, c);
      put_row(SB, c);
      sync();
      // symmatu(N)
#pragma unroll
      for (int jj = 0; jj < JW; jj++) if (j0 + jj < r) SA[r * LD + j0 + jj] = SA[(j0 + jj) * LD + r];
      sync();
      double pr = 0.0;
#pragma unroll 4
      for (int q = 0; q < M; q++) pr += SB[r * LD + q] * vr[q];
      const double as = ar + pr;
      bad |= !lg_finite(as);
      if (w == 0) at[((size_t)t * M + r) * em] = as;
      // Pt[t] - symmatu(Pt[t] N Pt[t])
      prod([&](int q) { return SB[r * LD + q]; }, [&](int q, int j) { return SA[q * LD + j]; }, c);
      put_row(SD, c);   // (every lane was through with the (L' N) row before the barriers above)
      sync();
      prod([&](int q) { return SD[r * LD + q]; }, [&](int q, int j) { return SB[q * LD + j]; }, c);
      put_row(SC, c);   // (and with L)
      sync();
#pragma unroll
      for (int jj = 0; jj < JW; jj++) {
        const int j = j0 + jj;
        const double x = (j < r) ? SC[j * LD + r] : c[jj];
        const double pn = SB[r * LD + j] - x;
        bad |= !lg_finite(pn);
        Pt[((size_t)t * MM + r + M * j) * em] = pn;
      }
    }
#pragma unroll
    for (int jj = 0; jj < JW; jj++) cc[((size_t)n * MM + r + M * (j0 + jj)) * em] = 0.0;
    // any lane of the group bad -> false in every lane of the group
    sync();
    vx[r] = 0.0;
    sync();
    if (bad) vx[r] = 1.0;
    sync();
    double any = 0.0;
#pragma unroll 4
    for (int j = 0; j < M; j++) any += vx[j];
    return any == 0.0;
  }
  // =================================================================================================================================
  // Round 6, late: the 64-dimensional step with its matrix products on the MATRIX CORES (one draw per workgroup, four waves).
  // The row form above is bound by LDS reads: a product asks for 17 LDS reads per 16 multiply-adds and lane, 4 x 64 x 17 reads of the one LDS
  // per product against 64 x 16 multiply-adds per SIMD.  v_mfma_f64_16x16x4_f64 takes ONE A and ONE B element per lane for 64 x 16 x 4
  // multiply-adds; wave w computes the 64 x 16 column strip [16 w, 16 w + 16) of a product in four accumulators (the B fragment of a k-chunk
  // serves the four row blocks): 5 LDS reads per 4 matrix instructions.  Two more changes of FORM, none of value:
  //   * (I - K Z') P (I - K Z')' + K HH K' with K = P Z / F is P - (P Z)(P Z)' / F: a rank-one update of the rows in place (two products less);
  //   * L = T (I - K Z') = T - (T K) Z': a matrix-vector product and a rank-one update (one product less).
  // 9 products on the matrix cores per time point instead of 12 row products.  The elementwise steps keep the row form (lane r: row r, wave w:
  // columns 16 w ...), the products hand their results over through the LDS slots.  Sums run in the matrix cores' order (k-chunks of four), so
  // the results differ from LgSmoother<M>::run in the last places (the set-up tests hold the kernels to the oracle at 1e-9 n).
  typedef double d4 __attribute__((ext_vector_type(4)));
  // acc = op(X) op(Y) for this wave's 16 columns and the four row blocks; XT: X(i, k) = Xs[k][i], YT: Y(k, j) = Ys[j][k]
  static constexpr int IB = M / 16;          // row blocks of a product (= column blocks = waves per draw)
  static_assert(NW % G == 0 && (NW / G == IB || !(M == 64 || M == 32)), "a draw's column blocks are its waves");
  // the draw whose product this WAVE works on and its column block there (M = 64: the workgroup's one draw, block w; M = 32: two draws per
  // workgroup, waves 0-1 the first one's two blocks, waves 2-3 the second one's) -- for half of the lanes not the draw whose row they hold in the
  // elementwise steps: `goff` leads from a lane's own slots to the wave's
  __device__ LG_INL int mm_draw() const { return G == 1 ? 0 : w / (NW / G); }
  __device__ LG_INL int mm_block() const { return G == 1 ? w : w % (NW / G); }
  __device__ LG_INL int mm_goff() const { return (mm_draw() - (int)((threadIdx.x & 63) / M)) * GS; }
  template <bool XT, bool YT>
  __device__ LG_INL void mm_acc(const double* Xs, const double* Ys, d4 (&acc)[IB]) const {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const int goff = mm_goff(), cb = mm_block();
    // this lane's operand addresses of k-chunk 0; a chunk further is 4 columns (rows when transposed) on
    const double* xp = Xs + goff + (XT ? lk * LD + li : li * LD + lk);
    const double* yp = Ys + goff + (YT ? (16 * cb + li) * LD + lk : lk * LD + 16 * cb + li);
    constexpr int XB = XT ? 16 : 16 * LD, XK = XT ? 4 * LD : 4, YK = YT ? 4 : 4 * LD;   // row-block and k-chunk strides
#pragma unroll
    for (int ib = 0; ib < IB; ib++) acc[ib] = d4{0.0, 0.0, 0.0, 0.0};
    const int kcn = (mr + 3) >> 2;   // (the k-chunks of the real states: the padding is exact zeros)
    // the operands of chunk kc + 1 are requested before the matrix instructions of chunk kc are issued
    double bv = yp[0], av[IB];
#pragma unroll
    for (int ib = 0; ib < IB; ib++) av[ib] = xp[ib * XB];
    for (int kc = 0; kc < kcn; kc++) {
      const int kn = kc + 1 < kcn ? kc + 1 : kc;   // (the last chunk asks for itself again: no branch around the loads)
      const double bn = yp[kn * YK];
      double an[IB];
#pragma unroll
      for (int ib = 0; ib < IB; ib++) an[ib] = xp[kn * XK + ib * XB];
#pragma unroll
      for (int ib = 0; ib < IB; ib++) acc[ib] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ib], bv, acc[ib], 0, 0, 0);
      bv = bn;
#pragma unroll
      for (int ib = 0; ib < IB; ib++) av[ib] = an[ib];
    }
  }
  // D fragment: element q of row block ib is row 16 ib + (lane >> 4) + 4 q, column 16 (column block) + (lane & 15)
  __device__ LG_INL void mm_store(double* Dst, const d4 (&acc)[IB]) const {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    double* dp = Dst + mm_goff() + lk * LD + 16 * mm_block() + li;
#pragma unroll
    for (int ib = 0; ib < IB; ib++)
#pragma unroll
      for (int q = 0; q < 4; q++) dp[(16 * ib + 4 * q) * LD] = acc[ib][q];
  }
  // sum over the M lanes that hold a draw's rows, in every one of them (fixed shape)
  __device__ LG_INL static double draw_sum(double x) {
    if constexpr (G == 1) return wave_sum(x);
    else {
#pragma unroll
      for (int o = 1; o < M; o <<= 1) x += __shfl_xor(x, o, 64);
      return x;
    }
  }
  __device__ LG_INL void get_row(const double* S, double (&c)[JW]) const {
#pragma unroll
    for (int jj = 0; jj < JW; jj++) c[jj] = S[r * LD + j0 + jj];
  }
  __device__ LG_INL bool run_mfma(const LgModel& g, double* at, double* Pt, double* cc, double* K, double* v, double* Fs, size_t es, size_t em) {
    const int n = g.n, sub = g.sub;
    int bad = 0;
    double c[JW];
    d4 acc[IB];
#ifdef LGW_STAMPS
    unsigned long long lgw_acc[32] = {0}, lgw_t0 = __builtin_amdgcn_s_memtime();
#endif
    const double a1r = g.a1[(size_t)r * es];
    load_row(g.P1, es, c);
    sync();
    double a_reg = a1r;   // a_t of this lane's row
    put_row(SA, c);
    if (w == 0) at[(size_t)r * em] = a1r;
    store_row(Pt, em, c);
    // ---- forward pass: SA = P, SD = T
    bool have_T = false;
    // R R' of the step: element (min, max) of the pair (the lower triangle takes the upper one's values below); a time-invariant one is read
    // ONCE (the model arrays are draw-interleaved: a row-distributed read of a matrix is 4 096 scattered 8-byte loads)
    const bool rr_const = g.Rs == 0 && sub == 1;
    double rrv[JW];
    // a step's results (a_{t+1}, P_{t+1}) go to memory from the NEXT step, behind its operand loads: a store issued at the end of a step is still
    // in flight when the next step waits for its operands, and the counter that tracks both cannot tell them apart (~1 us per step)
    double pend[JW];
    auto flush_fwd = [&](const int t) {   // the stores of step t: a_reg and pend hold a_{t+1}, P_{t+1}
      if (w == 0) at[((size_t)(t + 1) * M + r) * em] = a_reg;
      store_row(Pt + (size_t)(t + 1) * MM * em, em, pend);
      store_row(cc + (size_t)t * MM * em, em, pend);
    };
    if (rr_const) {
#pragma unroll
      for (int jj = 0; jj < JW; jj++) {
        const int j = j0 + jj, lo = j < r ? j : r, hi = j < r ? r : j;
        rrv[jj] = g.RR[(size_t)(lo + M * hi) * es];
      }
    }
    for (int t = 0; t < n; t++) {
      const int tr = sub == 1 ? t : t / sub, js = sub == 1 ? 0 : t - tr * sub;
      const bool moves = sub == 1 || js == sub - 1;   // (a step inside a time point: identity transition, no noise)
      const double zr = g.Z[((size_t)tr * g.Zs + (size_t)js * M + r) * es];
      const double yt = g.y[(size_t)t * es], sd = g.H[(size_t)t * g.Hs * es], Dt = g.D[((size_t)tr * g.Ds + js) * es];
      const double Cr = moves ? g.C[((size_t)tr * g.Cs + r) * es] : 0.0;
      const bool load_T = moves && (!have_T || g.Ts != 0 || sub != 1);
      if (load_T) load_row(g.T + (size_t)tr * g.Ts * es, es, c);
      if (!rr_const) {
#pragma unroll
        for (int jj = 0; jj < JW; jj++) {
          const int j = j0 + jj, lo = j < r ? j : r, hi = j < r ? r : j;
          rrv[jj] = moves ? g.RR[((size_t)tr * g.Rs + lo + M * hi) * es] : 0.0;
        }
      }
      sync_lds();
      LGW_T(0);
      vZ[r] = zr;
      if (load_T) { put_row(SD, c); have_T = true; }
      sync_lds();
      const bool na = !lg_finite(yt);
      const double HH = na ? 1.0 : sd * sd;
      if (t > 0) flush_fwd(t - 1);   // (this step's operands have arrived: the stores of the step before go out behind them)
      double pz = 0.0;
#pragma unroll 4
      for (int q = 0; q < mr; q++) pz += vZ[q] * SA[r * LD + q];
      vx[r] = pz;
      // Z' P Z and Z' a: lane r holds its term of both (one draw per wave: fixed-shape sums over the lanes instead of two loops over LDS)
      double F = na ? 0.0 : draw_sum(pz * zr);
      const double za = draw_sum(zr * a_reg);
      sync_lds();
      LGW_T(1);
      F += HH;
      const bool upd = !na && F > kZeroTol;
      const double vt = upd ? yt - Dt - za : 0.0;
      if (r == 0 && w == 0) { Fs[(size_t)t * es] = upd ? F : 0.0; v[(size_t)t * es] = vt; }
      const double Kr = upd ? pz / F : 0.0;
      const double aur = upd ? a_reg + Kr * vt : a_reg;
      if (w == 0) K[((size_t)t * M + r) * em] = Kr;
      vy[r] = aur;
      // inner = P - K (P Z)'  (this lane's elements of its row)
      if (upd) {
#pragma unroll
        for (int jj = 0; jj < JW; jj++) SA[r * LD + j0 + jj] -= Kr * vx[j0 + jj];
      }
      sync_lds();
      LGW_T(2);
      double a1n = aur + Cr;
      if (moves) {
        double s = 0.0;
#pragma unroll 4
        for (int q = 0; q < mr; q++) s += SD[r * LD + q] * vy[q];
        a1n = Cr + s;
        mm_acc<false, false>(SD, SA, acc);   // T inner
        mm_store(SC, acc);
        sync_lds();
        LGW_T(3);
        mm_acc<false, true>(SC, SD, acc);    // (T inner) T'
        mm_store(SB, acc);
        sync_lds();
        LGW_T(4);
        // + R R', symmatu: the lower triangle takes the upper one's values
#pragma unroll
        for (int jj = 0; jj < JW; jj++) { const int j = j0 + jj; c[jj] = (j < r ? SB[j * LD + r] : SB[r * LD + j]) + rrv[jj]; }
      } else {
#pragma unroll
        for (int jj = 0; jj < JW; jj++) { const int j = j0 + jj; c[jj] = j < r ? SA[j * LD + r] : SA[r * LD + j]; }
        sync_lds();   // every lane has read the upper triangle
      }
      put_row(SA, c);
      bad |= !lg_finite(a1n);
#pragma unroll
      for (int jj = 0; jj < JW; jj++) { bad |= !lg_finite(c[jj]); pend[jj] = c[jj]; }
      a_reg = a1n;
      LGW_T(5);
    }
    if (n > 0) flush_fwd(n - 1);
    // ---- backward pass (smoother_ccov): SA = N, vr = r_t
    sync();
    vr[r] = 0.0;
#pragma unroll
    for (int jj = 0; jj < JW; jj++) SA[r * LD + j0 + jj] = 0.0;
    __threadfence_block();
    sync();   // (also: the K, F and v stores of the forward pass are visible to the other waves of the workgroup)
    const bool t_const = g.Ts == 0;   // (a time-invariant T: its row is read once)
    double trow[JW];
    if (t_const) load_row(g.T, es, trow);
    // the operands a step reads from memory (written by the forward pass, never by a later backward step) are requested ONE STEP AHEAD -- in the
    // middle of the step before, behind its last use of the registers they arrive in: a round trip to memory (~2 us) is otherwise paid at the top of
    // every step by the one wave a SIMD has
    double nF = 0.0, nvt = 0.0, nzr = 0.0, nKr = 0.0, nar = 0.0, ntrow[JW], nccrow[JW], nptrow[JW];
    auto request = [&](const int t) {
      const int tr = sub == 1 ? t : t / sub, js = sub == 1 ? 0 : t - tr * sub;
      const bool moves = sub == 1 || js == sub - 1;
      nF = Fs[(size_t)t * es]; nvt = v[(size_t)t * es];
      nzr = g.Z[((size_t)tr * g.Zs + (size_t)js * M + r) * es];
      nKr = K[((size_t)t * M + r) * em];
      nar = at[((size_t)t * M + r) * em];
      if (moves && !t_const) load_row(g.T + (size_t)tr * g.Ts * es, es, ntrow);
      load_row(cc + (size_t)t * MM * em, em, nccrow);
      load_row(Pt + (size_t)t * MM * em, em, nptrow);
    };
    if (n > 0) request(n - 1);
    for (int t = n - 1; t >= 0; t--) {
      const double F = nF, vt = nvt;
      const bool upd = F > 0.0;
      const int tr = sub == 1 ? t : t / sub, js = sub == 1 ? 0 : t - tr * sub;
      const bool moves = sub == 1 || js == sub - 1;
      const double zr = nzr;
      const double Kr = upd ? nKr : 0.0;
      const double ar = nar;
      double ptrow[JW], ccrow[JW];
#pragma unroll
      for (int jj = 0; jj < JW; jj++) { ptrow[jj] = nptrow[jj]; ccrow[jj] = nccrow[jj]; }
      if (moves) {
#pragma unroll
        for (int jj = 0; jj < JW; jj++) c[jj] = t_const ? trow[jj] : ntrow[jj];
      } else {
#pragma unroll
        for (int jj = 0; jj < JW; jj++) c[jj] = (r == j0 + jj) ? 1.0 : 0.0;
      }
      if (t < n - 1) store_row(Pt + (size_t)(t + 1) * MM * em, em, pend);   // the smoothed covariance of the step before (see flush_fwd)
      if (t > 0) request(t - 1);
      sync_lds();           // the step before is through with every slot
      LGW_T(6);
      vZ[r] = zr;
      vK[r] = Kr;
      put_row(SD, c);       // T
      put_row(SB, ccrow);   // cc[t] (the predicted covariance of t + 1)
      sync_lds();
      LGW_T(7);
      // L = T (I - K Z') = T - (T K) Z' into SC
      double tk = 0.0;
#pragma unroll 4
      for (int q = 0; q < mr; q++) tk += SD[r * LD + q] * vK[q];
#pragma unroll
      for (int jj = 0; jj < JW; jj++) SC[r * LD + j0 + jj] = SD[r * LD + j0 + jj] - tk * vZ[j0 + jj];
      mm_acc<false, false>(SA, SB, acc);   // N cc[t]
      sync_lds();           // every lane is through with T; L is complete
      LGW_T(8);
      mm_store(SD, acc);
      sync_lds();
      LGW_T(9);
#pragma unroll
      for (int jj = 0; jj < JW; jj++) SD[r * LD + j0 + jj] = ((r == j0 + jj) ? 1.0 : 0.0) - SD[r * LD + j0 + jj];   // I - N cc[t]
      put_row(SB, ptrow);   // Pt[t]
      sync_lds();
      LGW_T(10);
      mm_acc<false, true>(SB, SC, acc);    // Pt[t] L'
      sync_lds();           // every lane is through with Pt[t]
      LGW_T(11);
      mm_store(SB, acc);
      sync_lds();
      LGW_T(12);
      mm_acc<false, false>(SB, SD, acc);   // cc[t] = Pt[t] L' (I - N cc[t])
      sync_lds();           // every lane is through with I - N cc[t]
      LGW_T(13);
      mm_store(SD, acc);
      mm_acc<true, false>(SC, SA, acc);    // L' N
      double rn = 0.0;  // r_{t-1} = L' r_t
#pragma unroll 4
      for (int q = 0; q < mr; q++) rn += SC[q * LD + r] * vr[q];
      mm_store(SB, acc);
      sync_lds();
      LGW_T(14);
      get_row(SD, c);
#pragma unroll
      for (int jj = 0; jj < JW; jj++) bad |= !lg_finite(c[jj]);
      store_row(cc + (size_t)t * MM * em, em, c);
      mm_acc<false, false>(SB, SC, acc);   // N_{t-1} = L' N L (+ Z Z' / F)
      mm_store(SA, acc);
      sync_lds();
      LGW_T(15);
      if (upd) {
        const double zf = zr / F;   // (one division for the row: Z Z' / F in the last place, as everything behind the matrix cores)
        rn = zf * vt + rn;
#pragma unroll
        for (int jj = 0; jj < JW; jj++) if (j0 + jj >= r) SA[r * LD + j0 + jj] = zf * vZ[j0 + jj] + SA[r * LD + j0 + jj];
      }
      vr[r] = rn;
      put_row(SB, ptrow);   // Pt[t] again (a broadcast operand below)
      sync_lds();
      LGW_T(16);
      // symmatu(N)
#pragma unroll
      for (int jj = 0; jj < JW; jj++) if (j0 + jj < r) SA[r * LD + j0 + jj] = SA[(j0 + jj) * LD + r];
      double pr = 0.0;
#pragma unroll 4
      for (int q = 0; q < mr; q++) pr += SB[r * LD + q] * vr[q];
      const double as = ar + pr;
      bad |= !lg_finite(as);
      if (w == 0) at[((size_t)t * M + r) * em] = as;
      sync_lds();
      LGW_T(17);
      // Pt[t] - symmatu(Pt[t] N Pt[t])
      mm_acc<false, false>(SB, SA, acc);
      mm_store(SD, acc);
      sync_lds();
      LGW_T(18);
      mm_acc<false, false>(SD, SB, acc);
      mm_store(SC, acc);
      sync_lds();
      LGW_T(19);
#pragma unroll
      for (int jj = 0; jj < JW; jj++) {
        const int j = j0 + jj;
        const double x = (j < r) ? SC[j * LD + r] : SC[r * LD + j];
        const double pn = ptrow[jj] - x;
        bad |= !lg_finite(pn);
        pend[jj] = pn;
      }
      LGW_T(20);
    }
    if (n > 0) store_row(Pt, em, pend);
#ifdef LGW_STAMPS
    if (threadIdx.x == 0 && blockIdx.x == 0) for (int k = 0; k < 32; k++) atomicAdd(&lgw_stamps[k], lgw_acc[k]);
#endif
#pragma unroll
    for (int jj = 0; jj < JW; jj++) cc[((size_t)n * MM + r + M * (j0 + jj)) * em] = 0.0;
    sync();
    vx[r] = 0.0;
    sync();
    if (bad) vx[r] = 1.0;
    sync();
    double any = 0.0;
#pragma unroll 4
    for (int j = 0; j < M; j++) any += vx[j];
    return any == 0.0;
  }
};

}  // namespace bssm
