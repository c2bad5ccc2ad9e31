// lg_smoother.h -- Kalman filter + covariance smoother + conditional factors of a univariate-observation
// linear-Gaussian model with an M-dimensional state, written once for host and device.
//
// This is the O(n M^3) core of the psi-APF set-up:
//   forward   ssm_ulg::smoother_ccov's Kalman pass        src/model_ssm_ulg.cpp:419-447 (update :122-136 in Joseph form)
//   backward  r_t / N_t recursions, smoothed mean, variance and lag-one cross-covariance   :448-473
//   factors   conditional_cov (use_svd = true)             src/conditional_dist.cpp:4-31
// The operation order follows the generic host routines in host_model.cpp (kf_predict / smoother_ccov /
// conditional_cov / sym_singular) term by term, so both give the same numbers; here everything lives in fixed-size
// arrays, which is what makes it (a) several times faster on the host and (b) runnable as one GPU lane per theta-draw.
#pragma once
#include <float.h>
#include <math.h>

#ifndef __HIPCC__
#define LG_HD
#else
#include <hip/hip_runtime.h>
#define LG_HD __host__ __device__
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// (complete unrolling up to 8 x 8; a 16 x 16 product -- states padded from 9 ... 16, round 5 -- stays a loop nest over scratch memory)
#define LG_UNROLL _Pragma("unroll kUnroll")
// the big routines are inlined into the set-up kernels: as separate device functions they get the default register budget of a
// callee (256 VGPRs) instead of the 512 a one-wave-per-SIMD kernel owns, and keep their 8 x 8 work arrays in scratch memory
#define LG_INL __attribute__((always_inline))
#else
#define LG_UNROLL
#define LG_INL
#endif
// analysis builds: the includer may define LG_STAMP(i) to record a clock at the phase boundaries of LgSmoother::run
#ifndef LG_STAMP
#define LG_STAMP(i)
#endif

namespace bssm {

// One draw's model, as plain pointers.  "(1|n)": stride 0 = time-invariant.
struct LgModel {
  int n;
  const double* Z;  int Zs;   // [(1|n)][M]      observation row
  const double* T;  int Ts;   // [(1|n)][M*M]    column-major
  const double* RR; int Rs;   // [(1|n)][M*M]    R R'
  const double* D;  int Ds;   // [(1|n)]
  const double* C;  int Cs;   // [(1|n)][M]
  const double* y;            // [n]  observations minus x'beta, NaN = missing
  const double* H;  int Hs;   // [(1|n)] observation sd
  const double* a1;           // [M]
  const double* P1;           // [M*M]
  // Multivariate observations with independent series (ssm_mng's approximating model: H diagonal), treated one series
  // at a time: `sub` = p virtual steps per time point, n = p * (time points).  Virtual step t observes series t % sub of
  // time point t / sub (Z row at Z + (t / sub) * Zs + (t % sub) * M, D at D + (t / sub) * Ds + t % sub, y and H at [t]);
  // the state moves (T, RR, C of time point t / sub) only behind the last series of a time point, in between the
  // transition is the identity without noise.  Smoothed moments of time point s sit at index s * sub; the conditional
  // factors are computed at those indices only.  sub = 1: the plain univariate model.
  int sub = 1;
};

LG_HD inline bool lg_finite(double x) { return x == x && fabs(x) <= DBL_MAX; }

template <int M>
struct LgSmoother {
  static constexpr int MM = M * M;
  static constexpr int kUnroll = M <= 8 ? M : 1;  // (LG_UNROLL)
  static constexpr double kZeroTol = 1e-12;  // src/model_ssm_ulg.h:19
  static constexpr double kEps = DBL_EPSILON;
#if defined(__HIP_DEVICE_COMPILE__)
  static constexpr bool kAhead = M <= 2;  // operands of the next step requested while this one computes (device, small states: registers to spare)
#else
  static constexpr bool kAhead = false;
#endif

  // C = A * B (all M x M, column-major), accumulation over the inner index in ascending order from 0
  // (LG_UNROLL: on the device the three loops are unrolled completely, so that the operands of a product are read in one batch
  // and the M^3 multiply-adds run on registers; left as loops, the arrays are indexed dynamically and every element access of an
  // 8 x 8 product is a dependent round trip to scratch memory)
  LG_HD static void mul(const double* A, const double* B, double* C) {
    LG_UNROLL
    for (int j = 0; j < M; j++) {
      LG_UNROLL
      for (int i = 0; i < M; i++) {
        double s = 0.0;
        LG_UNROLL
        for (int q = 0; q < M; q++) s += A[i + M * q] * B[q + M * j];
        C[i + M * j] = s;
      }
    }
  }
  LG_HD static void mul_bt(const double* A, const double* B, double* C) {  // A * B'
    LG_UNROLL
    for (int j = 0; j < M; j++) {
      LG_UNROLL
      for (int i = 0; i < M; i++) {
        double s = 0.0;
        LG_UNROLL
        for (int q = 0; q < M; q++) s += A[i + M * q] * B[j + M * q];
        C[i + M * j] = s;
      }
    }
  }
  LG_HD static void mul_at(const double* A, const double* B, double* C) {  // A' * B
    LG_UNROLL
    for (int j = 0; j < M; j++) {
      LG_UNROLL
      for (int i = 0; i < M; i++) {
        double s = 0.0;
        LG_UNROLL
        for (int q = 0; q < M; q++) s += A[q + M * i] * B[q + M * j];
        C[i + M * j] = s;
      }
    }
  }
  LG_HD static void symmatu(double* A) { for (int j = 0; j < M; j++) for (int i = j + 1; i < M; i++) A[i + M * j] = A[j + M * i]; }

  // |eigenvalues| (descending) and vectors of a symmetric matrix by cyclic Jacobi, sign convention: the largest
  // component of every vector is positive -- what arma::svd_econ(..., "left") yields for a PSD input up to sign
  LG_HD LG_INL static void sym_singular(const double* Ain, double* U, double* s) {
    double A[MM], W[MM];
    for (int j = 0; j < M; j++) for (int i = 0; i < M; i++) A[i + M * j] = 0.5 * (Ain[i + M * j] + Ain[j + M * i]);
    for (int q = 0; q < MM; q++) W[q] = 0.0;
    for (int i = 0; i < M; i++) W[i + M * i] = 1.0;
    for (int sweep = 0; sweep < 60; sweep++) {
      double off = 0.0, dg = 0.0;
      for (int j = 0; j < M; j++) for (int i = 0; i < j; i++) off += A[i + M * j] * A[i + M * j];
      for (int j = 0; j < M; j++) dg += A[j + M * j] * A[j + M * j];
      // converged: the off-diagonal mass is below 1e-30 of the diagonal's, so every further rotation has |sin| < 1e-30 and
      // leaves the eigenvalues and vectors unchanged in double precision (the sweeps go 1e-17, 1e-33, 1e-49, ... relative;
      // running them down to the absolute 1e-300 cost 6-9 sweeps per call where 2 do the work)
      if (off < 1e-300 || off < 1e-60 * dg) break;
      for (int p = 0; p < M - 1; p++)
        for (int q = p + 1; q < M; q++) {
          const double apq = A[p + M * q];
          if (apq == 0.0) continue;
          const double tau = (A[q + M * q] - A[p + M * p]) / (2.0 * apq);
          const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
          const double cs = 1.0 / sqrt(1.0 + t * t), sn = t * cs;
          for (int r = 0; r < M; r++) { const double x = A[r + M * p], y = A[r + M * q]; A[r + M * p] = cs * x - sn * y; A[r + M * q] = sn * x + cs * y; }
          for (int r = 0; r < M; r++) { const double x = A[p + M * r], y = A[q + M * r]; A[p + M * r] = cs * x - sn * y; A[q + M * r] = sn * x + cs * y; }
          for (int r = 0; r < M; r++) { const double x = W[r + M * p], y = W[r + M * q]; W[r + M * p] = cs * x - sn * y; W[r + M * q] = sn * x + cs * y; }
        }
    }
    double sv[M];
    int ord[M];
    for (int j = 0; j < M; j++) { sv[j] = fabs(A[j + M * j]); ord[j] = j; }
    for (int a = 1; a < M; a++) {  // stable insertion sort, descending
      const int o = ord[a];
      int b = a - 1;
      while (b >= 0 && sv[ord[b]] < sv[o]) { ord[b + 1] = ord[b]; b--; }
      ord[b + 1] = o;
    }
    for (int j = 0; j < M; j++) {
      s[j] = sv[ord[j]];
      int big = 0;
      for (int r = 1; r < M; r++) if (fabs(W[r + M * ord[j]]) > fabs(W[big + M * ord[j]])) big = r;
      const double sg = W[big + M * ord[j]] < 0 ? -1.0 : 1.0;
      for (int r = 0; r < M; r++) U[r + M * j] = sg * W[r + M * ord[j]];
    }
  }

  // Outputs (and working storage): alphahat [(n+1)][M], Vt [(n+1)][M*M], Ct [(n+1)][M*M].
  // Scratch: K [n][M], v [n], Fs [n] (0 where the update was skipped).
  // Every array element k of the model, the outputs and the scratch lives at ptr[k * es]: es = 1 on the host; the
  // device batch interleaves the draws (es = number of draws, lane d starts at ptr + d), so that the 64 lanes of a
  // wave touch consecutive doubles -- per-draw contiguous arrays cost 64 cache lines per wave instruction and were
  // 3x slower end to end.  Each step loads its operands into locals, computes as the generic host routines do, and
  // stores the results back.
  // Returns false when a Kalman step or a result is not finite (the filter then returns -Inf, :988-990 analogue).
  LG_HD static void ldv(const double* p, size_t es, int cnt, double* out) { for (int q = 0; q < cnt; q++) out[q] = p[(size_t)q * es]; }
  LG_HD static void stv(double* p, size_t es, int cnt, const double* in) { for (int q = 0; q < cnt; q++) p[(size_t)q * es] = in[q]; }

  // with_factors = false stops behind the backward pass: at / Pt / cc then hold the smoothed means, covariances and lag-one
  // cross-covariances, and the caller turns them into factors itself, one call of cond_step per time point in any order
  // (the set-up kernels split the time points over their waves)
  // conditional_cov (src/conditional_dist.cpp:4-31) at one time point t: from the smoothed covariances Pt[t - 1], Pt[t] and the
  // cross-covariance cc[t - 1] to Vout[t] = U sqrt(s) of Var(alpha_t | alpha_{t-1}, y) and Cout[t] = Cov' pinv(V_{t-1}); t = 0 has
  // the factor of Pt[0] only.  Vout / Cout may be Pt / cc themselves when the caller goes from the last time point down.
  // Returns whether everything it wrote is finite.
  LG_HD LG_INL static bool cond_step(int t, const double* Pt, const double* cc, double* Vout, double* Cout, size_t es) {
    double U[MM], s[M], W[MM], Vp[MM], Vc[MM], Cp[MM], tmp[MM], tmp2[MM];
    int bad = 0;
    // (round 6: factor columns with s_c <= eps M trace(the covariance before the subtraction) are zero -- the rule of
    // oracle/bssm_oracle.c: orc_conditional_cov, host_model.cpp: conditional_cov)
    if (t == 0) {
      ldv(Pt, es, MM, Vp);
      double tr0 = 0.0;
      for (int i = 0; i < M; i++) tr0 += Vp[i + M * i];
      sym_singular(Vp, U, s);
      for (int c = 0; c < M; c++) { const double f = s[c] > kEps * M * tr0 ? sqrt(s[c]) : 0.0; for (int i = 0; i < M; i++) Vp[i + M * c] = f == 0.0 ? 0.0 : U[i + M * c] * f; }
      stv(Vout, es, MM, Vp);
      for (int q = 0; q < MM; q++) bad |= !lg_finite(Vp[q]);
      return bad == 0;
    }
    ldv(Pt + (size_t)(t - 1) * MM * es, es, MM, Vp);
    ldv(Pt + (size_t)t * MM * es, es, MM, Vc);
    ldv(cc + (size_t)(t - 1) * MM * es, es, MM, Cp);
    sym_singular(Vp, U, s);
    for (int q = 0; q < MM; q++) W[q] = 0.0;
    for (int c = 0; c < M; c++) {
      if (!(s[c] > kEps * M * s[0])) continue;
      for (int j = 0; j < M; j++) for (int i = 0; i < M; i++) W[i + M * j] += U[i + M * c] * U[j + M * c] / s[c];
    }
    mul_at(Cp, W, tmp);       // tr(Ct[t-1]) * W
    mul(tmp, Cp, tmp2);
    double trc = 0.0;
    for (int i = 0; i < M; i++) trc += Vc[i + M * i];
    for (int q = 0; q < MM; q++) Vc[q] = Vc[q] - tmp2[q];
    stv(Cout + (size_t)t * MM * es, es, MM, tmp);
    for (int q = 0; q < MM; q++) bad |= !lg_finite(tmp[q]);
    sym_singular(Vc, U, s);
    for (int c = 0; c < M; c++) { const double f = s[c] > kEps * M * trc ? sqrt(s[c]) : 0.0; for (int i = 0; i < M; i++) Vc[i + M * c] = f == 0.0 ? 0.0 : U[i + M * c] * f; }
    stv(Vout + (size_t)t * MM * es, es, MM, Vc);
    for (int q = 0; q < MM; q++) bad |= !lg_finite(Vc[q]);
    return bad == 0;
  }

  LG_HD LG_INL static bool run(const LgModel& g, double* at, double* Pt, double* cc, double* K, double* v, double* Fs, size_t es = 1,
                        bool with_factors = true) {
    const int n = g.n;
    int bad = 0;
    double a[M], P[MM], a1n[M], P1n[MM];
    double tmp[MM], tmp2[MM], inner[MM], IKZ[MM];
    ldv(g.a1, es, M, a); ldv(g.P1, es, MM, P);
    stv(at, es, M, a); stv(Pt, es, MM, P);
    // ---- forward pass (kf_predict, univariate branch, Joseph form)
    const int sub = g.sub;
    // the operands of a step that do not depend on the recursion (model matrices, observation): on the device and for small states
    // they are requested ONE STEP AHEAD, so that their round trip runs under the previous step's arithmetic -- with one lane per
    // draw a step is otherwise a load round trip followed by a dependent chain, nothing else to switch to
    struct FwdIn { double Z[M], T[MM], RR[MM], Cv[M], yt, sd, Dv; };
    auto load_fwd = [&](int t, FwdIn& I) {
      const int tr = sub == 1 ? t : t / sub, js = sub == 1 ? 0 : t - tr * sub;
      ldv(g.Z + ((size_t)tr * g.Zs + (size_t)js * M) * es, es, M, I.Z);
      if (sub == 1 || js == sub - 1) {
        ldv(g.T + (size_t)tr * g.Ts * es, es, MM, I.T);
        ldv(g.RR + (size_t)tr * g.Rs * es, es, MM, I.RR);
        ldv(g.C + (size_t)tr * g.Cs * es, es, M, I.Cv);
      } else {
        for (int q = 0; q < MM; q++) { I.T[q] = 0.0; I.RR[q] = 0.0; }
        for (int i = 0; i < M; i++) { I.T[i + M * i] = 1.0; I.Cv[i] = 0.0; }
      }
      I.yt = g.y[(size_t)t * es];
      I.sd = g.H[(size_t)t * g.Hs * es];
      I.Dv = g.D[((size_t)tr * g.Ds + js) * es];
    };
    FwdIn fin, fnx;
    if (kAhead && n > 0) load_fwd(0, fin);
    for (int t = 0; t < n; t++) {
      const int tr = sub == 1 ? t : t / sub, js = sub == 1 ? 0 : t - tr * sub;
      const bool moves = sub == 1 || js == sub - 1;  // the state moves behind this (virtual) step
      (void)tr;
      if (kAhead) { if (t + 1 < n) load_fwd(t + 1, fnx); } else load_fwd(t, fin);
      double (&Z)[M] = fin.Z; double (&T)[MM] = fin.T; double (&RR)[MM] = fin.RR; double (&Cv)[M] = fin.Cv;
      const double yt = fin.yt;
      const bool na = !lg_finite(yt);
      const double sd = fin.sd;
      const double Dt = fin.Dv;
      const double HH = na ? 1.0 : sd * sd;
      double F = 0.0;
      if (!na) {
        double zp[M];
        for (int j = 0; j < M; j++) { double s = 0.0; for (int q = 0; q < M; q++) s += Z[q] * P[q + M * j]; zp[j] = s; }
        for (int q = 0; q < M; q++) F += zp[q] * Z[q];
      }
      F += HH;
      const bool upd = !na && F > kZeroTol;
      Fs[(size_t)t * es] = upd ? F : 0.0;
      double Kt[M];
      if (!upd) {
        v[(size_t)t * es] = 0.0;
        for (int i = 0; i < M; i++) Kt[i] = 0.0;
        if (moves) {
          for (int i = 0; i < M; i++) { double s = 0.0; for (int q = 0; q < M; q++) s += T[i + M * q] * a[q]; a1n[i] = Cv[i] + s; }
          mul(T, P, tmp); mul_bt(tmp, T, P1n);
          for (int q = 0; q < MM; q++) P1n[q] = P1n[q] + RR[q];
        } else {  // identity transition without noise: the products with I and the sums with 0 are exact, so they are skipped
          for (int i = 0; i < M; i++) a1n[i] = a[i];
          for (int q = 0; q < MM; q++) P1n[q] = P[q];
        }
        symmatu(P1n);
      } else {
        double za = 0.0;
        for (int c = 0; c < M; c++) za += Z[c] * a[c];
        const double vt = yt - Dt - za;
        v[(size_t)t * es] = vt;
        double au[M];
        for (int i = 0; i < M; i++) { double s = 0.0; for (int q = 0; q < M; q++) s += P[i + M * q] * Z[q]; Kt[i] = s / F; }
        for (int i = 0; i < M; i++) au[i] = a[i] + Kt[i] * vt;
        if (moves) { for (int i = 0; i < M; i++) { double s = 0.0; for (int q = 0; q < M; q++) s += T[i + M * q] * au[q]; a1n[i] = Cv[i] + s; } }
        else for (int i = 0; i < M; i++) a1n[i] = au[i];
        for (int j = 0; j < M; j++) for (int i = 0; i < M; i++) IKZ[i + M * j] = (i == j ? 1.0 : 0.0) - Kt[i] * Z[j];
        mul(IKZ, P, tmp); mul_bt(tmp, IKZ, inner);
        for (int j = 0; j < M; j++) for (int i = 0; i < M; i++) inner[i + M * j] = inner[i + M * j] + Kt[i] * HH * Kt[j];
        if (moves) {
          mul(T, inner, tmp); mul_bt(tmp, T, P1n);
          for (int q = 0; q < MM; q++) P1n[q] = P1n[q] + RR[q];
        } else {
          for (int q = 0; q < MM; q++) P1n[q] = inner[q];
        }
        symmatu(P1n);
      }
      stv(K + (size_t)t * M * es, es, M, Kt);
      // (a non-finite value is carried to the end of the pass and reported there: NaN / Inf only propagate, and twelve early-exit branches
      // per time point sat on the dependent chain of a kernel that is nothing but that chain)
      for (int i = 0; i < M; i++) bad |= !lg_finite(a1n[i]);
      for (int q = 0; q < MM; q++) bad |= !lg_finite(P1n[q]);
      stv(at + (size_t)(t + 1) * M * es, es, M, a1n);
      stv(Pt + (size_t)(t + 1) * MM * es, es, MM, P1n);
      stv(cc + (size_t)t * MM * es, es, MM, P1n);
      for (int i = 0; i < M; i++) a[i] = a1n[i];
      for (int q = 0; q < MM; q++) P[q] = P1n[q];
      if (kAhead) fin = fnx;
    }
    if (bad) return false;
    LG_STAMP(2);
    // ---- backward pass (smoother_ccov)
    double rt[M], Nt[MM], L[MM], N2[MM];
    for (int i = 0; i < M; i++) rt[i] = 0.0;
    for (int q = 0; q < MM; q++) Nt[q] = 0.0;
    // (the same one step ahead: what step t reads was written by the forward pass, and step t + 1 only rewrites its own slots)
    struct BwdIn { double Z[M], T[MM], a[M], P[MM], ct[MM], Kt[M], F, vt; };
    auto load_bwd = [&](int t, BwdIn& I) {
      const int tr = sub == 1 ? t : t / sub, js = sub == 1 ? 0 : t - tr * sub;
      ldv(g.Z + ((size_t)tr * g.Zs + (size_t)js * M) * es, es, M, I.Z);
      if (sub == 1 || js == sub - 1) {
        ldv(g.T + (size_t)tr * g.Ts * es, es, MM, I.T);
      } else {
        for (int q = 0; q < MM; q++) I.T[q] = 0.0;
        for (int i = 0; i < M; i++) I.T[i + M * i] = 1.0;
      }
      ldv(at + (size_t)t * M * es, es, M, I.a);
      ldv(Pt + (size_t)t * MM * es, es, MM, I.P);
      if (sub == 1 || js == sub - 1) ldv(cc + (size_t)t * MM * es, es, MM, I.ct);
      I.F = Fs[(size_t)t * es];
      ldv(K + (size_t)t * M * es, es, M, I.Kt);  // (zeros where the step had no update)
      I.vt = v[(size_t)t * es];
    };
    BwdIn bin, bnx;
    if (kAhead && n > 0) load_bwd(n - 1, bin);
    for (int t = n - 1; t >= 0; t--) {
      const int tr = sub == 1 ? t : t / sub, js = sub == 1 ? 0 : t - tr * sub;
      (void)tr;
      const bool movesb = sub == 1 || js == sub - 1;
      // Series at a time (sub > 1): the conditional factors are taken at the time points only (virtual steps s * sub), from the
      // covariances of the steps s * sub - 1 and s * sub and the cross-covariance of step s * sub - 1 -- the other virtual steps only
      // carry r_t / N_t through, so their smoothed covariance and cross-covariance (five of a step's seven matrix products) are
      // not formed; what is needed comes out bit for bit as before
      const bool need_P = sub == 1 || js == 0 || js == sub - 1, need_cc = sub == 1 || js == sub - 1;
      if (kAhead) { if (t > 0) load_bwd(t - 1, bnx); } else load_bwd(t, bin);
      double (&Z)[M] = bin.Z; double (&T)[MM] = bin.T; double (&a)[M] = bin.a; double (&P)[MM] = bin.P; double (&ct)[MM] = bin.ct;
      double (&Kt)[M] = bin.Kt;
      const double F = bin.F, vt_b = bin.vt;
      const double* Lm = T;
      if (F > 0.0 && movesb) {
        for (int j = 0; j < M; j++) for (int i = 0; i < M; i++) IKZ[i + M * j] = (i == j ? 1.0 : 0.0) - Kt[i] * Z[j];
        mul(T, IKZ, L); Lm = L;
      }
      // cc[t] = Pt[t] L' (I - Nt cc[t])
      if (need_cc) {
        mul(Nt, ct, tmp);
        for (int j = 0; j < M; j++) for (int i = 0; i < M; i++) tmp[i + M * j] = (i == j ? 1.0 : 0.0) - tmp[i + M * j];
        mul_bt(P, Lm, tmp2);
        mul(tmp2, tmp, ct);
      }
      double rn[M];
      if (movesb) {
        for (int i = 0; i < M; i++) { double s = 0.0; for (int q = 0; q < M; q++) s += Lm[q + M * i] * rt[q]; rn[i] = s; }
        mul_at(Lm, Nt, tmp); mul(tmp, Lm, N2);
      } else if (F > 0.0) {
        // a virtual step inside a time point: L = I - K Z' is a rank-one change of the identity, so L' r and L' N L are formed in
        // O(M^2) instead of two M^3 products: with w = N K (N symmetric) and kappa = K' w,
        //   L' r = r - Z (K' r),   L' N L = N - Z w' - w Z' + kappa Z Z'
        double w[M], kr = 0.0, kappa = 0.0;
        for (int i = 0; i < M; i++) { double s = 0.0; for (int q = 0; q < M; q++) s += Nt[i + M * q] * Kt[q]; w[i] = s; }
        for (int q = 0; q < M; q++) { kr += Kt[q] * rt[q]; kappa += Kt[q] * w[q]; }
        for (int i = 0; i < M; i++) rn[i] = rt[i] - Z[i] * kr;
        for (int j = 0; j < M; j++) for (int i = 0; i < M; i++) N2[i + M * j] = Nt[i + M * j] - Z[i] * w[j] - w[i] * Z[j] + kappa * Z[i] * Z[j];
      } else {  // no update at this virtual step: L = I
        for (int i = 0; i < M; i++) rn[i] = rt[i];
        for (int q = 0; q < MM; q++) N2[q] = Nt[q];
      }
      if (F > 0.0) {
        const double vt = vt_b;
        for (int i = 0; i < M; i++) rn[i] = Z[i] / F * vt + rn[i];
        for (int j = 0; j < M; j++) for (int i = 0; i < M; i++) N2[i + M * j] = Z[i] * Z[j] / F + N2[i + M * j];
      }
      for (int i = 0; i < M; i++) rt[i] = rn[i];
      for (int q = 0; q < MM; q++) Nt[q] = N2[q];
      symmatu(Nt);
      for (int i = 0; i < M; i++) { double s = 0.0; for (int q = 0; q < M; q++) s += P[i + M * q] * rt[q]; a[i] = a[i] + s; }
      for (int i = 0; i < M; i++) bad |= !lg_finite(a[i]);
      stv(at + (size_t)t * M * es, es, M, a);
      if (need_cc) {
        for (int q = 0; q < MM; q++) bad |= !lg_finite(ct[q]);
        stv(cc + (size_t)t * MM * es, es, MM, ct);
      }
      if (need_P) {
        mul(P, Nt, tmp); mul(tmp, P, tmp2);
        symmatu(tmp2);
        for (int q = 0; q < MM; q++) P[q] = P[q] - tmp2[q];
        for (int q = 0; q < MM; q++) bad |= !lg_finite(P[q]);
        stv(Pt + (size_t)t * MM * es, es, MM, P);
      }
      if (kAhead) bin = bnx;
    }
    for (int q = 0; q < MM; q++) cc[((size_t)n * MM + q) * es] = 0.0;
    if (bad) return false;
    LG_STAMP(3);
    if (!with_factors) return true;
    // ---- conditional_cov: Vt <- U sqrt(s) of Var(alpha_t | alpha_{t-1}, y), Ct <- Cov' pinv(V_{t-1}); in place, so from the
    // last time point down (step t reads the covariances of t - 1 and t and the cross-covariance of t - 1)
    bool ok = true;
    for (int t = n; t >= 0; t--) {
      if (sub > 1 && t % sub != 0) continue;  // a virtual step inside a time point: no factor of its own
      const bool okt = cond_step(t, Pt, cc, Pt, cc, es);
      ok = ok && okt;
    }
    LG_STAMP(4);
    // (every smoothed mean, covariance and cross-covariance was tested where the backward pass produced it, every factor in
    // cond_step; the predicted mean of slot n in the forward pass: a last sweep over the arrays has nothing left to find)
    return ok;
  }

  // Smoothed state means only (ssm_ulg::fast_smoother, src/model_ssm_ulg.cpp:252-305): the inner step of the mode
  // search.  at [(n+1)][M] out; scratch rts [n][M], K [n][M], v [n], Fs [n].
  LG_HD static bool run_mean(const LgModel& g, double* at, double* rts, double* K, double* v, double* Fs) {
    const int n = g.n;
    double P[MM], Pn[MM], tmp[MM], inner[MM], IKZ[MM], L[MM];
    for (int i = 0; i < M; i++) at[i] = g.a1[i];
    for (int q = 0; q < MM; q++) P[q] = g.P1[q];
    for (int t = 0; t < n; t++) {
      const double* Z = g.Z + (size_t)t * g.Zs;
      const double* T = g.T + (size_t)t * g.Ts;
      const double* RR = g.RR + (size_t)t * g.Rs;
      const double* Cv = g.C + (size_t)t * g.Cs;
      const double* a = at + (size_t)t * M;
      double* a1n = at + (size_t)(t + 1) * M;
      const double yt = g.y[t];
      const bool na = !lg_finite(yt);
      const double sd = g.H[(size_t)t * g.Hs];
      const double HH = na ? 1.0 : sd * sd;
      double F = 0.0;
      if (!na) {
        double zp[M];
        for (int j = 0; j < M; j++) { double s = 0.0; for (int q = 0; q < M; q++) s += Z[q] * P[q + M * j]; zp[j] = s; }
        for (int q = 0; q < M; q++) F += zp[q] * Z[q];
      }
      F += HH;
      const bool upd = !na && F > kZeroTol;
      Fs[t] = upd ? F : 0.0;
      if (!upd) {
        v[t] = 0.0;
        for (int i = 0; i < M; i++) K[(size_t)t * M + i] = 0.0;
        for (int i = 0; i < M; i++) { double s = 0.0; for (int q = 0; q < M; q++) s += T[i + M * q] * a[q]; a1n[i] = Cv[i] + s; }
        mul(T, P, tmp); mul_bt(tmp, T, Pn);
      } else {
        double za = 0.0;
        for (int c = 0; c < M; c++) za += Z[c] * a[c];
        const double vt = yt - g.D[(size_t)t * g.Ds] - za;
        v[t] = vt;
        double Kt[M], au[M];
        for (int i = 0; i < M; i++) { double s = 0.0; for (int q = 0; q < M; q++) s += P[i + M * q] * Z[q]; Kt[i] = s / F; K[(size_t)t * M + i] = Kt[i]; }
        for (int i = 0; i < M; i++) au[i] = a[i] + Kt[i] * vt;
        for (int i = 0; i < M; i++) { double s = 0.0; for (int q = 0; q < M; q++) s += T[i + M * q] * au[q]; a1n[i] = Cv[i] + s; }
        for (int j = 0; j < M; j++) for (int i = 0; i < M; i++) IKZ[i + M * j] = (i == j ? 1.0 : 0.0) - Kt[i] * Z[j];
        mul(IKZ, P, tmp); mul_bt(tmp, IKZ, inner);
        for (int j = 0; j < M; j++) for (int i = 0; i < M; i++) inner[i + M * j] = inner[i + M * j] + Kt[i] * HH * Kt[j];
        mul(T, inner, tmp); mul_bt(tmp, T, Pn);
      }
      for (int q = 0; q < MM; q++) P[q] = Pn[q] + RR[q];
      symmatu(P);
    }
    double rprev[M];
    for (int i = 0; i < M; i++) { rprev[i] = 0.0; if (n > 0) rts[(size_t)(n - 1) * M + i] = 0.0; }
    for (int t = n - 1; t >= 0; t--) {
      const double* Z = g.Z + (size_t)t * g.Zs;
      const double* T = g.T + (size_t)t * g.Ts;
      const double* rt = rts + (size_t)t * M;
      const double F = Fs[t];
      const double* Lm = T;
      if (F > 0.0) {
        const double* Kt = K + (size_t)t * M;
        for (int j = 0; j < M; j++) for (int i = 0; i < M; i++) IKZ[i + M * j] = (i == j ? 1.0 : 0.0) - Kt[i] * Z[j];
        mul(T, IKZ, L);
        Lm = L;
      }
      double r0[M];
      for (int i = 0; i < M; i++) { double s = 0.0; for (int q = 0; q < M; q++) s += Lm[q + M * i] * rt[q]; r0[i] = s; }
      if (F > 0.0) for (int i = 0; i < M; i++) r0[i] = Z[i] / F * v[t] + r0[i];
      for (int i = 0; i < M; i++) { if (t > 0) rts[(size_t)(t - 1) * M + i] = r0[i]; else rprev[i] = r0[i]; }
    }
    for (int i = 0; i < M; i++) { double s = 0.0; for (int q = 0; q < M; q++) s += g.P1[i + M * q] * rprev[q]; at[i] = g.a1[i] + s; }
    for (int t = 0; t < n - 1; t++) {
      const double* T = g.T + (size_t)t * g.Ts;
      const double* RR = g.RR + (size_t)t * g.Rs;
      const double* Cv = g.C + (size_t)t * g.Cs;
      const double* a = at + (size_t)t * M;
      const double* rt = rts + (size_t)t * M;
      double* an = at + (size_t)(t + 1) * M;
      for (int i = 0; i < M; i++) {
        double s = 0.0, q2 = 0.0;
        for (int q = 0; q < M; q++) s += T[i + M * q] * a[q];
        for (int q = 0; q < M; q++) q2 += RR[i + M * q] * rt[q];
        an[i] = Cv[i] + s + q2;
      }
    }
    return true;
  }
};

// run-time dispatch over the state dimension (1..8)
LG_HD inline bool lg_smoother_run(int m, const LgModel& g, double* at, double* Pt, double* cc, double* K, double* v, double* Fs) {
  switch (m) {
    case 1: return LgSmoother<1>::run(g, at, Pt, cc, K, v, Fs);
    case 2: return LgSmoother<2>::run(g, at, Pt, cc, K, v, Fs);
    case 3: return LgSmoother<3>::run(g, at, Pt, cc, K, v, Fs);
    case 4: return LgSmoother<4>::run(g, at, Pt, cc, K, v, Fs);
    case 5: return LgSmoother<5>::run(g, at, Pt, cc, K, v, Fs);
    case 6: return LgSmoother<6>::run(g, at, Pt, cc, K, v, Fs);
    case 7: return LgSmoother<7>::run(g, at, Pt, cc, K, v, Fs);
    case 8: return LgSmoother<8>::run(g, at, Pt, cc, K, v, Fs);
  }
  return false;
}
LG_HD inline bool lg_smoother_mean(int m, const LgModel& g, double* at, double* rts, double* K, double* v, double* Fs) {
  switch (m) {
    case 1: return LgSmoother<1>::run_mean(g, at, rts, K, v, Fs);
    case 2: return LgSmoother<2>::run_mean(g, at, rts, K, v, Fs);
    case 3: return LgSmoother<3>::run_mean(g, at, rts, K, v, Fs);
    case 4: return LgSmoother<4>::run_mean(g, at, rts, K, v, Fs);
    case 5: return LgSmoother<5>::run_mean(g, at, rts, K, v, Fs);
    case 6: return LgSmoother<6>::run_mean(g, at, rts, K, v, Fs);
    case 7: return LgSmoother<7>::run_mean(g, at, rts, K, v, Fs);
    case 8: return LgSmoother<8>::run_mean(g, at, rts, K, v, Fs);
  }
  return false;
}

}  // namespace bssm
