// lg_smoother_wave.h -- the Kalman filter + covariance smoother of lg_smoother.h (LgSmoother<M>::run without the conditional factors)
// for states of 17 ... 64 dimensions (padded to M = 32 or 64): ONE WAVE per 64 / M draws, lane r owns ROW r, and every M x M matrix of the
// recursion lives in LDS (rows padded to M + 1 doubles: a column read is as conflict-free as a row read).
//
// Why not lg_smoother_rows.h: its rows live in registers with every loop unrolled -- seven row arrays of 32 doubles no longer fit a lane's 512
// registers, a 64-dimensional state not by far, and a completely unrolled 32 x 32 product is 2.5 minutes of compilation per kernel.  Here a
// product C = A B keeps only the OUTPUT row in registers: c[j] += A(r, q) * B(q, j) for q = 0, 1, ... (the q loop is a real loop, the j loop
// unrolled) -- A(r, q) is this lane's own operand (from its LDS row), B(q, j) a broadcast read that is the same address in every lane of the
// group.  Four matrix slots per group (A B C D) carry a step; slot D only ever holds rows that their own lane reads.
//
// The arithmetic follows LgSmoother<M>::run term by term -- the same products in the same association, every sum over the inner index in
// ascending order from zero -- with the two differences lg_smoother_rows.h has as well: a step without an update runs the update formulas
// with K = 0 (products with exact zeros and ones), and a non-finite value is carried to the end and reported there.  Univariate observations
// only (LgModel::sub == 1).  Device only; one wave per workgroup (blockDim.x = 64); every lane of the wave must call it.
#pragma once
#include "lg_smoother.h"

namespace bssm {

template <int M>
struct LgWave {
  static_assert(M == 32 || M == 64, "M lanes per draw must tile the wave");
  static constexpr int MM = M * M, G = 64 / M, LD = M + 1, MAT = M * LD;
  static constexpr int GS = 4 * MAT + 6 * M + 2;  // per group: four matrices, six vectors (+ 2: groups start in different banks)
  static constexpr size_t kLdsDoubles = (size_t)G * GS;
  static constexpr double kZeroTol = 1e-12;

  double *SA, *SB, *SC, *SD;          // this group's matrix slots
  double *vZ, *va, *vK, *vx, *vy, *vr;  // this group's vectors
  int r;                              // the row this lane owns
  __device__ LgWave(double* lds_wave) {
    const int lane = threadIdx.x & 63;
    r = lane % M;
    double* base = lds_wave + (size_t)(lane / M) * GS;
    SA = base; SB = base + MAT; SC = base + 2 * MAT; SD = base + 3 * MAT;
    vZ = base + 4 * MAT; va = vZ + M; vK = va + M; vx = vK + M; vy = vx + M; vr = vy + M;
  }
  __device__ LG_INL static void sync() { __syncthreads(); }

  // c[j] = sum over q (ascending, from 0.0) of own(q) * bop(q, j)
  template <class Own, class Bop>
  __device__ LG_INL static void prod(Own own, Bop bop, double (&c)[M]) {
#pragma unroll
    for (int j = 0; j < M; j++) c[j] = 0.0;
#pragma unroll 1
    for (int q = 0; q < M; q++) {
      const double a = own(q);
#pragma unroll
      for (int j = 0; j < M; j++) c[j] += a * bop(q, j);
    }
  }
  __device__ LG_INL void put_row(double* S, const double (&c)[M]) {
#pragma unroll
    for (int j = 0; j < M; j++) S[r * LD + j] = c[j];
  }

  // at [(n+1)][M], Pt [(n+1)][MM], cc [(n+1)][MM], K [n][M], v [n], Fs [n]: element k at ptr[k * es] (draw-interleaved), as LgSmoother::run
  // leaves them with with_factors = false.  Returns the draw's verdict (the same in the M lanes of a group).
  __device__ LG_INL bool run(const LgModel& g, double* at, double* Pt, double* cc, double* K, double* v, double* Fs, size_t es) {
    const int n = g.n;
    int bad = 0;
    double c[M];
    const double a1r = g.a1[(size_t)r * es];
#pragma unroll
    for (int j = 0; j < M; j++) c[j] = g.P1[(size_t)(r + M * j) * es];
    sync();
    va[r] = a1r;
    put_row(SA, c);
    at[(size_t)r * es] = a1r;
#pragma unroll
    for (int j = 0; j < M; j++) Pt[(size_t)(r + M * j) * es] = c[j];
    // ---- forward pass (kf_predict, univariate branch, Joseph form): SA = P, SD = T
    bool have_T = false;
    for (int t = 0; t < n; t++) {
      const double zr = g.Z[((size_t)t * g.Zs + r) * es];
      const double yt = g.y[(size_t)t * es], sd = g.H[(size_t)t * g.Hs * es], Dt = g.D[(size_t)t * g.Ds * es];
      const double Cr = g.C[((size_t)t * g.Cs + r) * es];
      if (!have_T || g.Ts != 0) {
#pragma unroll
        for (int j = 0; j < M; j++) c[j] = g.T[((size_t)t * g.Ts + r + M * j) * es];
      }
      sync();
      vZ[r] = zr;
      if (!have_T || g.Ts != 0) { put_row(SD, c); have_T = true; }
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
      if (r == 0) { Fs[(size_t)t * es] = upd ? F : 0.0; v[(size_t)t * es] = vt; }
      const double Kr = upd ? pz / F : 0.0;
      const double aur = upd ? va[r] + Kr * vt : va[r];
      K[((size_t)t * M + r) * es] = Kr;
      vK[r] = Kr;
      vy[r] = aur;
      // I - K Z' into SB
#pragma unroll
      for (int j = 0; j < M; j++) SB[r * LD + j] = ((r == j) ? 1.0 : 0.0) - Kr * vZ[j];
      sync();
      double s = 0.0;
#pragma unroll 4
      for (int q = 0; q < M; q++) s += SD[r * LD + q] * vy[q];
      const double a1n = Cr + s;
      // Joseph form: inner = (I - K Z') P (I - K Z')' + K HH K'
      prod([&](int q) { return SB[r * LD + q]; }, [&](int q, int j) { return SA[q * LD + j]; }, c);
      put_row(SC, c);
      prod([&](int q) { return SC[r * LD + q]; }, [&](int q, int j) { return SB[j * LD + q]; }, c);
#pragma unroll
      for (int j = 0; j < M; j++) c[j] = upd ? c[j] + Kr * HH * vK[j] : c[j];
      sync();           // every lane is through with P
      put_row(SA, c);   // inner
      sync();
      // T inner T' + R R'
      prod([&](int q) { return SD[r * LD + q]; }, [&](int q, int j) { return SA[q * LD + j]; }, c);
      put_row(SC, c);
      prod([&](int q) { return SC[r * LD + q]; }, [&](int q, int j) { return SD[j * LD + q]; }, c);
#pragma unroll
      for (int j = 0; j < M; j++) c[j] = c[j] + g.RR[((size_t)t * g.Rs + r + M * j) * es];
      sync();           // every lane is through with inner
      put_row(SA, c);
      sync();
      // symmatu: the lower triangle takes the upper one's values
#pragma unroll
      for (int j = 0; j < M; j++) if (j < r) c[j] = SA[j * LD + r];
#pragma unroll
      for (int j = 0; j < M; j++) if (j < r) SA[r * LD + j] = c[j];
      bad |= !lg_finite(a1n);
#pragma unroll
      for (int j = 0; j < M; j++) bad |= !lg_finite(c[j]);
      at[((size_t)(t + 1) * M + r) * es] = a1n;
#pragma unroll
      for (int j = 0; j < M; j++) {
        Pt[((size_t)(t + 1) * MM + r + M * j) * es] = c[j];
        cc[((size_t)t * MM + r + M * j) * es] = c[j];
      }
      va[r] = a1n;
    }
    // ---- backward pass (smoother_ccov): SA = N, vr = r_t; T, P and the cross-covariance are staged from memory row by row
    sync();
    vr[r] = 0.0;
#pragma unroll
    for (int j = 0; j < M; j++) SA[r * LD + j] = 0.0;
    sync();   // (also: the K, F and v stores of the last forward step are visible to the other lanes of the group)
    for (int t = n - 1; t >= 0; t--) {
      const double F = Fs[(size_t)t * es], vt = v[(size_t)t * es];
      const bool upd = F > 0.0;
      const double zr = g.Z[((size_t)t * g.Zs + r) * es];
      const double Kr = upd ? K[((size_t)t * M + r) * es] : 0.0;
      const double ar = at[((size_t)t * M + r) * es];
      // L = T (I - K Z'): T row into SD, I - K Z' into SB, L into SC
#pragma unroll
      for (int j = 0; j < M; j++) c[j] = g.T[((size_t)t * g.Ts + r + M * j) * es];
      sync();
      vZ[r] = zr;
      vK[r] = Kr;
      put_row(SD, c);
      sync();
#pragma unroll
      for (int j = 0; j < M; j++) SB[r * LD + j] = ((r == j) ? 1.0 : 0.0) - Kr * vZ[j];
      sync();
      prod([&](int q) { return SD[r * LD + q]; }, [&](int q, int j) { return SB[q * LD + j]; }, c);
      put_row(SC, c);
      // cc[t] = Pt[t] L' (I - N cc[t]): cc[t] into SB, then I - N cc[t] into SB; Pt[t] row into SD, then (Pt[t] L') row into SD
#pragma unroll
      for (int j = 0; j < M; j++) c[j] = cc[((size_t)t * MM + r + M * j) * es];
      sync();           // every lane is through with I - K Z'; L is complete
      put_row(SB, c);
      sync();
      prod([&](int q) { return SA[r * LD + q]; }, [&](int q, int j) { return SB[q * LD + j]; }, c);
#pragma unroll
      for (int j = 0; j < M; j++) c[j] = ((r == j) ? 1.0 : 0.0) - c[j];
      sync();
      put_row(SB, c);
#pragma unroll
      for (int j = 0; j < M; j++) c[j] = Pt[((size_t)t * MM + r + M * j) * es];
      put_row(SD, c);
      sync();
      prod([&](int q) { return SD[r * LD + q]; }, [&](int q, int j) { return SC[j * LD + q]; }, c);
      put_row(SD, c);
      prod([&](int q) { return SD[r * LD + q]; }, [&](int q, int j) { return SB[q * LD + j]; }, c);
#pragma unroll
      for (int j = 0; j < M; j++) { bad |= !lg_finite(c[j]); cc[((size_t)t * MM + r + M * j) * es] = c[j]; }
      // r_{t-1} = L' r_t, N_{t-1} = L' N L (column r of L is this lane's operand)
      double rn = 0.0;
#pragma unroll 4
      for (int q = 0; q < M; q++) rn += SC[q * LD + r] * vr[q];
      prod([&](int q) { return SC[q * LD + r]; }, [&](int q, int j) { return SA[q * LD + j]; }, c);
      put_row(SD, c);
      prod([&](int q) { return SD[r * LD + q]; }, [&](int q, int j) { return SC[q * LD + j]; }, c);
      if (upd) {
        rn = zr / F * vt + rn;
#pragma unroll
        for (int j = 0; j < M; j++) c[j] = zr * vZ[j] / F + c[j];
      }
      sync();           // every lane is through with N, r_t and I - N cc[t]
      vr[r] = rn;
      put_row(SA, c);
      // Pt[t] row into SB (it is a broadcast operand below)
#pragma unroll
      for (int j = 0; j < M; j++) c[j] = Pt[((size_t)t * MM + r + M * j) * es];
      put_row(SB, c);
      sync();
      // symmatu(N)
#pragma unroll
      for (int j = 0; j < M; j++) if (j < r) SA[r * LD + j] = SA[j * LD + r];
      sync();
      double pr = 0.0;
#pragma unroll 4
      for (int q = 0; q < M; q++) pr += SB[r * LD + q] * vr[q];
      const double as = ar + pr;
      bad |= !lg_finite(as);
      at[((size_t)t * M + r) * es] = as;
      // Pt[t] - symmatu(Pt[t] N Pt[t])
      prod([&](int q) { return SB[r * LD + q]; }, [&](int q, int j) { return SA[q * LD + j]; }, c);
      put_row(SD, c);
      prod([&](int q) { return SD[r * LD + q]; }, [&](int q, int j) { return SB[q * LD + j]; }, c);
      sync();           // every lane is through with L
      put_row(SC, c);
      sync();
#pragma unroll
      for (int j = 0; j < M; j++) {
        const double x = (j < r) ? SC[j * LD + r] : c[j];
        const double pn = SB[r * LD + j] - x;
        bad |= !lg_finite(pn);
        Pt[((size_t)t * MM + r + M * j) * es] = pn;
      }
    }
#pragma unroll
    for (int j = 0; j < M; j++) cc[((size_t)n * MM + r + M * j) * es] = 0.0;
    // any lane of the group bad -> false in every lane of the group
    sync();
    vx[r] = bad ? 1.0 : 0.0;
    sync();
    double any = 0.0;
#pragma unroll 4
    for (int j = 0; j < M; j++) any += vx[j];
    return any == 0.0;
  }
};

}  // namespace bssm
