// lg_smoother_rows.h -- the Kalman filter + covariance smoother of lg_smoother.h (LgSmoother<M>::run without the conditional
// factors) for ONE WAVE per 64 / M draws: M lanes share a draw, lane r holds ROW r of every M x M matrix of the recursion.
//
// Why: with one lane per draw an 8-dimensional state needs five or six live 8 x 8 matrices = 640-768 VGPRs per lane; the kernel
// owns 512, so part of every product went through scratch memory and the sequential backward pass of the C4 shape (m = 8, p = 4,
// T = 200) took 42 ms for ANY number of draws.  Row-distributed, a matrix costs 16 VGPRs per lane, the operands other lanes need go
// through a few KB of LDS (rows written once, read by all lanes of the group), and eight times as many waves work.
//
// The arithmetic follows LgSmoother<M>::run term by term -- same products, sums over the inner index in ascending order, vectors
// replicated in the lanes of a group -- with two differences that do not change a needed result: (1) a step without an update
// (missing observation or F <= tolerance) runs the update formulas with K = 0, v = 0, which reproduce the no-update branch exactly
// (products with exact zeros and ones), so that the wave never diverges around its LDS exchanges; (2) a non-finite value does not
// end the draw at once, it is carried to the end and reported there.  Device only; every lane of the wave must call it.
#pragma once
#include "lg_smoother.h"

namespace bssm {

// (M = 16: the products stay completely unrolled -- static register indices -- but a scheduling fence ends every output element, so that
// the compiler does not hoist the 256 LDS reads of a product in front of its multiply-adds: 920 spilt registers otherwise)
#define LGR_FENCE do { if constexpr (M > 8) __builtin_amdgcn_sched_barrier(0); } while (0)

template <int M>
struct LgRows {
  static_assert(64 % M == 0, "M lanes per draw must tile the wave");
  static constexpr int MM = M * M, G = 64 / M;
  static constexpr int GS = 2 * MM + M + 2;  // doubles of LDS per group: two matrices, one vector (+ 2: groups start in different banks)
  static constexpr size_t kLdsDoubles = (size_t)G * GS;
  static constexpr double kZeroTol = 1e-12;

  double* sh;  // this group's LDS block
  int r;       // the row this lane holds
  __device__ LgRows(double* lds_wave) {
    const int lane = threadIdx.x & 63;
    r = lane % M;
    sh = lds_wave + (size_t)(lane / M) * GS;
  }
  // rows of a matrix into LDS buffer `b` (0 / 1); the barriers fence the previous readers and the new ones (one wave per workgroup)
  __device__ LG_INL void put(int b, const double (&row)[M]) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < M; j++) sh[b * MM + r * M + j] = row[j];
    __syncthreads();
  }
  __device__ LG_INL double get(int b, int i, int j) const { return sh[b * MM + i * M + j]; }
  // one value per lane -> the vector in every lane
  __device__ LG_INL void gather(double x, double (&out)[M]) {
    __syncthreads();
    sh[2 * MM + r] = x;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < M; q++) out[q] = sh[2 * MM + q];
  }
  // symmatu: the lower triangle takes the upper one's values (row r, column j < r  <-  row j, column r)
  __device__ LG_INL void symmatu(double (&row)[M]) {
    put(0, row);
#pragma unroll
    for (int j = 0; j < M; j++) if (j < r) row[j] = get(0, j, r);
  }

  // at [(n+1)][M], Pt [(n+1)][MM], cc [(n+1)][MM], K [n][M], v [n], Fs [n]: element k at ptr[k * es] (draw-interleaved), as LgSmoother::run
  // leaves them with with_factors = false.  Returns the draw's verdict (the same in the M lanes of a group).
  __device__ LG_INL bool run(const LgModel& g, double* at, double* Pt, double* cc, double* K, double* v, double* Fs, size_t es) {
    const int n = g.n, sub = g.sub;
    int bad = 0;
    double a[M], P[M], Z[M], T[M], RR[M], Cv[M], Kt[M];
#pragma unroll
    for (int q = 0; q < M; q++) a[q] = g.a1[(size_t)q * es];
#pragma unroll
    for (int j = 0; j < M; j++) P[j] = g.P1[(size_t)(r + M * j) * es];
    at[(size_t)r * es] = a[r];
#pragma unroll
    for (int j = 0; j < M; j++) Pt[(size_t)(r + M * j) * es] = P[j];
    // ---- forward pass (kf_predict, univariate branch, Joseph form)
    // the operands of a step that do not depend on the recursion are requested one step ahead (their round trip runs under the
    // previous step's LDS exchanges and arithmetic)
    struct FwdIn { double Z[M], T[M], RR[M], Cv[M], yt, sd, Dv; };
    auto load_fwd = [&](int t, FwdIn& I) {
      const int tr = sub == 1 ? t : t / sub, js = sub == 1 ? 0 : t - tr * sub;
#pragma unroll
      for (int q = 0; q < M; q++) I.Z[q] = g.Z[((size_t)tr * g.Zs + (size_t)js * M + q) * es];
      if (sub == 1 || js == sub - 1) {
#pragma unroll
        for (int j = 0; j < M; j++) { I.T[j] = g.T[((size_t)tr * g.Ts + r + M * j) * es]; I.RR[j] = g.RR[((size_t)tr * g.Rs + r + M * j) * es]; }
#pragma unroll
        for (int q = 0; q < M; q++) I.Cv[q] = g.C[((size_t)tr * g.Cs + q) * es];
      }
      I.yt = g.y[(size_t)t * es];
      I.sd = g.H[(size_t)t * g.Hs * es];
      I.Dv = g.D[((size_t)tr * g.Ds + js) * es];
    };
    // (M = 16: the operands are loaded where they are used -- two sets of 67 doubles on top of the seven row arrays overflow the 512 registers)
    constexpr bool kAhead = M <= 8;
    FwdIn fin, fnx;
    if (kAhead && n > 0) load_fwd(0, fin);
    for (int t = 0; t < n; t++) {
      const int tr = sub == 1 ? t : t / sub, js = sub == 1 ? 0 : t - tr * sub;
      const bool moves = sub == 1 || js == sub - 1;
      (void)tr;
      if constexpr (kAhead) { if (t + 1 < n) load_fwd(t + 1, fnx); } else load_fwd(t, fin);
#pragma unroll
      for (int q = 0; q < M; q++) { Z[q] = fin.Z[q]; T[q] = fin.T[q]; RR[q] = fin.RR[q]; Cv[q] = fin.Cv[q]; }
      const double yt = fin.yt;
      const bool na = !lg_finite(yt);
      const double sd = fin.sd;
      const double Dt = fin.Dv;
      if constexpr (kAhead) fin = fnx;
      const double HH = na ? 1.0 : sd * sd;
      // zp = Z' P = P Z (P is symmetric to the last bit: symmatu below): row r times Z
      double pz = 0.0, zp[M];
#pragma unroll
      for (int q = 0; q < M; q++) pz += Z[q] * P[q];
      gather(pz, zp);
      double F = 0.0;
      if (!na) {
#pragma unroll
        for (int q = 0; q < M; q++) F += zp[q] * Z[q];
      }
      F += HH;
      const bool upd = !na && F > kZeroTol;
      double za = 0.0;
#pragma unroll
      for (int c = 0; c < M; c++) za += Z[c] * a[c];
      const double vt = upd ? yt - Dt - za : 0.0;
      if (r == 0) { Fs[(size_t)t * es] = upd ? F : 0.0; v[(size_t)t * es] = vt; }
      // K = P Z / F (zero without an update): lane r has (P Z)_r = pz
      gather(upd ? pz / F : 0.0, Kt);
      double au[M], a1n[M];
#pragma unroll
      for (int q = 0; q < M; q++) au[q] = upd ? a[q] + Kt[q] * vt : a[q];
      if (moves) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < M; q++) s += T[q] * au[q];
        gather(Cv[r] + s, a1n);
      } else {
#pragma unroll
        for (int q = 0; q < M; q++) a1n[q] = au[q];
      }
      // Joseph form: inner = (I - K Z') P (I - K Z')' + K HH K'
      double tmp[M], inner[M], P1n[M];
      put(0, P);
#pragma unroll
      for (int j = 0; j < M; j++) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < M; q++) s += ((r == q ? 1.0 : 0.0) - Kt[r] * Z[q]) * get(0, q, j);
        tmp[j] = s;
        LGR_FENCE;
      }
#pragma unroll
      for (int j = 0; j < M; j++) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < M; q++) s += tmp[q] * ((j == q ? 1.0 : 0.0) - Kt[j] * Z[q]);
        inner[j] = upd ? s + Kt[r] * HH * Kt[j] : s;
      }
      if (moves) {  // T inner T' + R R'
        put(0, inner);
#pragma unroll
        for (int j = 0; j < M; j++) {
          double s = 0.0;
#pragma unroll
          for (int q = 0; q < M; q++) s += T[q] * get(0, q, j);
          tmp[j] = s;
          LGR_FENCE;
        }
        put(1, T);
#pragma unroll
        for (int j = 0; j < M; j++) {
          double s = 0.0;
#pragma unroll
          for (int q = 0; q < M; q++) s += tmp[q] * get(1, j, q);
          P1n[j] = s + RR[j];
          LGR_FENCE;
        }
      } else {
#pragma unroll
        for (int j = 0; j < M; j++) P1n[j] = inner[j];
      }
      symmatu(P1n);
      K[((size_t)t * M + r) * es] = Kt[r];
      bad |= !lg_finite(a1n[r]);
#pragma unroll
      for (int j = 0; j < M; j++) bad |= !lg_finite(P1n[j]);
      at[((size_t)(t + 1) * M + r) * es] = a1n[r];
#pragma unroll
      for (int j = 0; j < M; j++) {
        Pt[((size_t)(t + 1) * MM + r + M * j) * es] = P1n[j];
        cc[((size_t)t * MM + r + M * j) * es] = P1n[j];
      }
#pragma unroll
      for (int q = 0; q < M; q++) a[q] = a1n[q];
#pragma unroll
      for (int j = 0; j < M; j++) P[j] = P1n[j];
    }
    // ---- backward pass (smoother_ccov)
    double rt[M], N[M], L[M], ct[M];
#pragma unroll
    for (int q = 0; q < M; q++) { rt[q] = 0.0; N[q] = 0.0; ct[q] = 0.0; }
    // (one step ahead here too: what step t reads was written by the forward pass, and step t + 1 only rewrites its own slots)
    struct BwdIn { double Z[M], T[M], P[M], ct[M], Kt[M], ar, F, vt; };
    auto load_bwd = [&](int t, BwdIn& I) {
      const int tr = sub == 1 ? t : t / sub, js = sub == 1 ? 0 : t - tr * sub;
#pragma unroll
      for (int q = 0; q < M; q++) I.Z[q] = g.Z[((size_t)tr * g.Zs + (size_t)js * M + q) * es];
      if (sub == 1 || js == sub - 1) {
#pragma unroll
        for (int j = 0; j < M; j++) { I.T[j] = g.T[((size_t)tr * g.Ts + r + M * j) * es]; I.ct[j] = cc[((size_t)t * MM + r + M * j) * es]; }
      }
      I.ar = at[((size_t)t * M + r) * es];
#pragma unroll
      for (int j = 0; j < M; j++) I.P[j] = Pt[((size_t)t * MM + r + M * j) * es];
      I.F = Fs[(size_t)t * es];
#pragma unroll
      for (int q = 0; q < M; q++) I.Kt[q] = K[((size_t)t * M + q) * es];  // (zeros where the step had no update)
      I.vt = v[(size_t)t * es];
    };
    BwdIn bin, bnx;
    // the first backward load reads what OTHER lanes of this draw stored in the last forward iteration (K, Fs, v): make those
    // stores visible first (the M lanes of a draw share a wave, one wave per workgroup here, so the barrier is cheap)
    __syncthreads();
    if (kAhead && n > 0) load_bwd(n - 1, bin);
    for (int t = n - 1; t >= 0; t--) {
      const int tr = sub == 1 ? t : t / sub, js = sub == 1 ? 0 : t - tr * sub;
      (void)tr;
      const bool movesb = sub == 1 || js == sub - 1;
      const bool need_P = sub == 1 || js == 0 || js == sub - 1, need_cc = sub == 1 || js == sub - 1;
      if constexpr (kAhead) { if (t > 0) load_bwd(t - 1, bnx); } else load_bwd(t, bin);
      const double F = bin.F, vt_b = bin.vt;
      const bool upd = F > 0.0;
      const double ar = bin.ar;
#pragma unroll
      for (int q = 0; q < M; q++) { Z[q] = bin.Z[q]; T[q] = bin.T[q]; P[q] = bin.P[q]; Kt[q] = upd ? bin.Kt[q] : 0.0; }
      if (need_cc) {
#pragma unroll
        for (int j = 0; j < M; j++) ct[j] = bin.ct[j];
      }
      if constexpr (kAhead) bin = bnx;
      double rn[M], N2[M], tmp[M], tmp2[M];
      if (movesb) {
        // L = T (I - K Z'), row r
#pragma unroll
        for (int j = 0; j < M; j++) {
          double s = 0.0;
#pragma unroll
          for (int q = 0; q < M; q++) s += T[q] * ((q == j ? 1.0 : 0.0) - Kt[q] * Z[j]);
          L[j] = upd ? s : T[j];
        }
        put(1, L);
        if (need_cc) {  // cc[t] = Pt[t] L' (I - N cc[t])
          put(0, ct);
#pragma unroll
          for (int j = 0; j < M; j++) {
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < M; q++) s += N[q] * get(0, q, j);
            tmp[j] = (r == j ? 1.0 : 0.0) - s;
            LGR_FENCE;
          }
#pragma unroll
          for (int j = 0; j < M; j++) {
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < M; q++) s += P[q] * get(1, j, q);
            tmp2[j] = s;
            LGR_FENCE;
          }
          put(0, tmp);
#pragma unroll
          for (int j = 0; j < M; j++) {
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < M; q++) s += tmp2[q] * get(0, q, j);
            ct[j] = s;
            LGR_FENCE;
          }
        }
        // r_{t-1} = L' r_t ; N_{t-1} = L' N L  (column r of L from buffer 1)
        double s0 = 0.0;
#pragma unroll
        for (int q = 0; q < M; q++) s0 += get(1, q, r) * rt[q];
        gather(s0, rn);
        put(0, N);
#pragma unroll
        for (int j = 0; j < M; j++) {
          double s = 0.0;
#pragma unroll
          for (int q = 0; q < M; q++) s += get(1, q, r) * get(0, q, j);
          tmp[j] = s;
          LGR_FENCE;
        }
#pragma unroll
        for (int j = 0; j < M; j++) {
          double s = 0.0;
#pragma unroll
          for (int q = 0; q < M; q++) s += tmp[q] * get(1, q, j);
          N2[j] = s;
          LGR_FENCE;
        }
      } else {
        // a virtual step inside a time point: L = I - K Z' (the identity without an update); with w = N K, kappa = K' w:
        //   L' r = r - Z (K' r),   L' N L = N - Z w' - w Z' + kappa Z Z'
        double w[M], wr = 0.0, kr = 0.0, kappa = 0.0;
#pragma unroll
        for (int q = 0; q < M; q++) wr += N[q] * Kt[q];
        gather(wr, w);
#pragma unroll
        for (int q = 0; q < M; q++) { kr += Kt[q] * rt[q]; kappa += Kt[q] * w[q]; }
#pragma unroll
        for (int i = 0; i < M; i++) rn[i] = upd ? rt[i] - Z[i] * kr : rt[i];
#pragma unroll
        for (int j = 0; j < M; j++) N2[j] = upd ? N[j] - Z[r] * w[j] - w[r] * Z[j] + kappa * Z[r] * Z[j] : N[j];
      }
      if (upd) {
        const double vt = vt_b;
#pragma unroll
        for (int i = 0; i < M; i++) rn[i] = Z[i] / F * vt + rn[i];
#pragma unroll
        for (int j = 0; j < M; j++) N2[j] = Z[r] * Z[j] / F + N2[j];
      }
#pragma unroll
      for (int q = 0; q < M; q++) { rt[q] = rn[q]; N[q] = N2[q]; }
      symmatu(N);
      double pr = 0.0;
#pragma unroll
      for (int q = 0; q < M; q++) pr += P[q] * rt[q];
      const double as = ar + pr;
      bad |= !lg_finite(as);
      at[((size_t)t * M + r) * es] = as;
      if (need_cc) {
#pragma unroll
        for (int j = 0; j < M; j++) { bad |= !lg_finite(ct[j]); cc[((size_t)t * MM + r + M * j) * es] = ct[j]; }
      }
      if (need_P) {  // P - P N P
        put(0, N);
#pragma unroll
        for (int j = 0; j < M; j++) {
          double s = 0.0;
#pragma unroll
          for (int q = 0; q < M; q++) s += P[q] * get(0, q, j);
          tmp[j] = s;
          LGR_FENCE;
        }
        put(1, P);
#pragma unroll
        for (int j = 0; j < M; j++) {
          double s = 0.0;
#pragma unroll
          for (int q = 0; q < M; q++) s += tmp[q] * get(1, q, j);
          tmp2[j] = s;
          LGR_FENCE;
        }
        symmatu(tmp2);
#pragma unroll
        for (int j = 0; j < M; j++) { P[j] = P[j] - tmp2[j]; bad |= !lg_finite(P[j]); Pt[((size_t)t * MM + r + M * j) * es] = P[j]; }
      }
    }
    if (r == 0) {
#pragma unroll
      for (int q = 0; q < MM; q++) cc[((size_t)n * MM + q) * es] = 0.0;
    }
    // the verdict of the draw: any lane of the group saw a non-finite value
    double flags[M];
    gather(bad ? 1.0 : 0.0, flags);
    double any = 0.0;
#pragma unroll
    for (int q = 0; q < M; q++) any += flags[q];
    return any == 0.0;
  }
};

}  // namespace bssm
