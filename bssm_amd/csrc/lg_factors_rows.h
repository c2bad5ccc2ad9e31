// lg_factors_rows.h -- the conditional factors of the psi set-up (LgSmoother<M>::cond_step: conditional_cov with use_svd = true,
// src/conditional_dist.cpp:4-31) for states of 16 and more dimensions, ROW-DISTRIBUTED: M lanes share one (draw, time point), lane r owns
// row r, and the M x M matrices of the step live in LDS (rows padded to M + 1 doubles: a column read is as conflict-free as a row read).
//
// Why: with one lane per (draw, time point) the work matrices of a 16 x 16 step are 18 KB of scratch memory per LANE (74 KB at 32 x 32),
// every element access a dependent round trip -- 172 ms per 2 048 monthly-seasonal draws, 83 % of that set-up -- and 32 x 32 does not run at
// all.  Here a cyclic Jacobi rotation (p, q) is ONE LDS round trip that is parallel over the M lanes: every lane reads the column pair of ITS
// row (A[r][p], A[r][q], the eigenvector rows W[r][p], W[r][q]) and the row pair in ITS column (A[p][r], A[q][r]) together -- the column
// update leaves the row pair of a lane outside {p, q} untouched, and the lanes p and q form the 2 x 2 block's intermediate values from the four
// block elements themselves -- and then writes its results of both updates.  Two matrix slots per group: X (the Jacobi matrix, then U, then the
// own-row operands of the products) and Y (the eigenvector accumulator, then the pseudo-inverse, then the cross-covariance): 68 KB at M = 64,
// two workgroups per CU.  At 64 (padded) states a workgroup is NW = 4 waves: lane r of every wave works on row r, wave w owns the columns
// [w M / NW, (w + 1) M / NW) of it in the products and every NW-th rotation of a round-robin step.
//
// The arithmetic follows LgSmoother<M>::sym_singular / cond_step term by term: the same rotations in the same cyclic order from the same
// formulas, products summed over the inner index in ascending order, the same sign and ordering conventions (the stable descending sort is
// evaluated as ranks: #{larger} + #{equal, earlier}).  Two differences that do not change a needed result: the off-diagonal / diagonal masses of
// the stopping test are summed row by row (they only decide when to stop, and a further sweep is a no-op in double precision), and a
// non-finite value is carried to the end and reported there.
// Device only; one wave per workgroup (blockDim.x = 64); every lane of the wave must call it.
#pragma once
#include "lg_smoother.h"
#include "pf_device.h"   // lane_broadcast, wave_max_key

namespace bssm {

// time points >= 1 whose workgroup took the full (Jacobi) path, and all of them: a diagnostic of the thin path's coverage (bssm_debug_thin_counts)
__device__ unsigned long long lgf_thin_counts[2];
#ifdef LGF_STATS
__device__ unsigned long long lgf_stats[8];   // analysis builds (scripts/lgwave_check.hip): calls, sweeps, clocks of the Jacobi loops / of cond_step
#endif

// (NW: four waves at 64 states -- 256 draws of 53 states 328 -> 244 ms, and 148 ms once the kernel was held to 256 registers so that two workgroups
// share a CU; two waves at 32 states: at 347 registers they were SLOWER in a full batch than one, 38 -> 47 ms per 512 draws -- half the workgroups
// per CU, every wave repeating the per-step parameters -- and at 256 registers (nine spilt) faster, 39.6 -> 29.5 ms)
#ifndef LGF_NW32
#define LGF_NW32 2   // waves per workgroup at 32 (padded) states
#endif
template <int M, int NW = (M >= 64 ? M / 16 : (M == 32 ? LGF_NW32 : 1))>
struct LgFactorRows {
  static_assert(64 % M == 0 && M >= 16, "M lanes per matrix must tile the wave");
  static_assert(M % NW == 0, "the waves split the columns evenly");
  static constexpr int MM = M * M, G = 64 / M, LD = M + 1, MAT = M * LD, JW = M / NW;
  static constexpr int GS = 2 * MAT + 4 * M + 2;   // per group: two matrices, four vectors (+ 2: groups start in different banks)
  static constexpr size_t kLdsDoubles = (size_t)G * GS;
  static constexpr int kThreads = 64 * NW;
  static constexpr double kEps = DBL_EPSILON;

  double *A, *W;              // this group's matrix slots: X (Jacobi work matrix -> U -> own-row operands), Y (eigenvector accumulator -> pseudo-inverse -> cross-covariance)
  double *va, *vb, *vc, *vd;  // this group's vectors
  int r;                      // the row (and, in a rotation, also the column) this lane works on
  int w, j0;                  // its wave, the first of the JW columns the wave owns
  double chol_ratio = 1.0;    // smallest / first pivot of the last RELATIVE pivoted Cholesky (the thin path's factor of V_{t-1}): ~ 1 / condition number
  bool skip = false;          // this group's draw has already failed: its decompositions take no rotations (the result is discarded)
  int mr = M;                 // rows / columns from mr on are exact zeros (states padded to M): no rotation touches them
  __device__ LgFactorRows(double* lds_wg) {
    const int lane = threadIdx.x & 63;
    r = lane % M;
    w = (int)threadIdx.x >> 6;
    j0 = w * JW;
    double* base = lds_wg + (size_t)(lane / M) * GS;
    A = base; W = base + MAT;
    va = base + 2 * MAT; vb = va + M; vc = vb + M; vd = vc + M;
  }

  // |eigenvalues| (descending) left in vd[], eigenvectors U (sign: the largest component of every vector positive) left in A, of the symmetric
  // part of the matrix whose row r (this lane's columns of it) is passed in.  Uses A, W and the four vectors.
  __device__ LG_INL void sym_singular(const double (&row)[JW]) {
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < JW; jj++) { A[r * LD + j0 + jj] = row[jj]; W[r * LD + j0 + jj] = (j0 + jj == r) ? 1.0 : 0.0; }
    __syncthreads();
    {
      double sym[JW];
#pragma unroll
      for (int jj = 0; jj < JW; jj++) sym[jj] = 0.5 * (row[jj] + A[(j0 + jj) * LD + r]);
      __syncthreads();
#pragma unroll
      for (int jj = 0; jj < JW; jj++) A[r * LD + j0 + jj] = sym[jj];
    }
    __syncthreads();
    bool done = skip;
#ifdef LGF_STATS
    const unsigned long long lgf_c0 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) atomicAdd(&lgf_stats[0], 1ull);
#endif
    for (int sweep = 0; sweep < 60; sweep++) {
#ifdef LGF_STATS
      if (threadIdx.x == 0) atomicAdd(&lgf_stats[1], 1ull);
#endif
      // stopping test of the sweep (LgSmoother::sym_singular): off-diagonal mass below 1e-60 of the diagonal's
      double po = 0.0;
      for (int j = r + 1; j < M; j++) { const double x = A[r * LD + j]; po += x * x; }
      const double dr = A[r * LD + r];
      __syncthreads();
      va[r] = po; vb[r] = dr * dr;   // (every wave of the workgroup stores the same values)
      __syncthreads();
      double off = 0.0, dg = 0.0;
      for (int j = 0; j < M; j++) { off += va[j]; dg += vb[j]; }
      done = done || off < 1e-300 || off < 1e-60 * dg;
      if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;   // (every group has converged: the same verdict in every wave)
      if constexpr (M >= 32) {
        // ROUND-ROBIN ordering (the tournament schedule): a step is mr / 2 rotations on disjoint index pairs -- they commute, so the step is
        // A <- J' A J with J their product: the parameters of pair i by lane i from the matrix as the step finds it, then the column pairs of
        // every row are rotated (of A and of the eigenvector accumulator), then the row pairs in every column; wave w takes the pairs w, w + NW, ...
        // A sweep is mr - 1 steps instead of mr (mr - 1) / 2 dependent rotations; the same rotations as the cyclic order up to the order of
        // roundings (the host takes the cyclic order: the decompositions agree to the last places, and differ as any two exact methods do
        // where eigenvalues repeat).
        const int ne = (mr + 1) & ~1, half = ne / 2, ring = ne - 1;
        auto pair_of = [&](int k, int i, int& p, int& q) {   // (k and i are the same in every lane: scalar arithmetic)
          int a = k + i, b = k - i;
          a = a >= ring ? a - ring : a; b = b < 0 ? b + ring : b;
          if (i == 0) { a = ring; b = k; }
          p = a < b ? a : b; q = a < b ? b : a;
        };
        constexpr int B = 8;   // pairs per batch: their operands are requested together (a rotation is otherwise one LDS round trip after the other)
        const int mine = (half - w + NW - 1) / NW;   // pairs of this wave: w, w + NW, ...
        for (int k = 0; k < ring; k++) {
          __syncthreads();
          if (r < half) {   // (every wave computes all parameters: the same values into the same places)
            int p, q;
            pair_of(k, r, p, q);
            double cs = 1.0, sn = 0.0;
            if (q < mr && !done) {
              const double apq = A[p * LD + q];
              if (apq != 0.0) {
                // tangent and cosine from three long operations: with h = aqq - app, g = 2 apq:  t = sign(h g) |g| / (|h| + sqrt(h^2 + g^2)),  cs = rsqrt(1 + t^2)
                const double h = A[q * LD + q] - A[p * LD + p], g = 2.0 * apq;
                const double sgn = (h == 0.0 || (h > 0) == (g > 0)) ? 1.0 : -1.0;
                const double t = sgn * fabs(g) / (fabs(h) + sqrt(h * h + g * g));
                cs = rsqrt(1.0 + t * t);
                sn = t * cs;
              }
            }
            va[r] = cs; vb[r] = sn;
          }
          __syncthreads();
          for (int i0 = 0; i0 < mine; i0 += B) {
            double cs[B], sn[B], x[B], y[B], wx[B], wy[B];
            int pp[B], qq[B];
#pragma unroll
            for (int u = 0; u < B; u++) {
              const int i = w + NW * (i0 + u < mine ? i0 + u : mine - 1);
              pair_of(k, i, pp[u], qq[u]);
              cs[u] = va[i]; sn[u] = vb[i];
              x[u] = A[r * LD + pp[u]]; y[u] = A[r * LD + qq[u]]; wx[u] = W[r * LD + pp[u]]; wy[u] = W[r * LD + qq[u]];
            }
#pragma unroll
            for (int u = 0; u < B; u++) {
              if (i0 + u < mine && sn[u] != 0.0) {
                A[r * LD + pp[u]] = cs[u] * x[u] - sn[u] * y[u]; A[r * LD + qq[u]] = sn[u] * x[u] + cs[u] * y[u];
                W[r * LD + pp[u]] = cs[u] * wx[u] - sn[u] * wy[u]; W[r * LD + qq[u]] = sn[u] * wx[u] + cs[u] * wy[u];
              }
            }
          }
          __syncthreads();
          for (int i0 = 0; i0 < mine; i0 += B) {
            double cs[B], sn[B], x[B], y[B];
            int pp[B], qq[B];
#pragma unroll
            for (int u = 0; u < B; u++) {
              const int i = w + NW * (i0 + u < mine ? i0 + u : mine - 1);
              pair_of(k, i, pp[u], qq[u]);
              cs[u] = va[i]; sn[u] = vb[i];
              x[u] = A[pp[u] * LD + r]; y[u] = A[qq[u] * LD + r];
            }
#pragma unroll
            for (int u = 0; u < B; u++) {
              if (i0 + u < mine && sn[u] != 0.0) { A[pp[u] * LD + r] = cs[u] * x[u] - sn[u] * y[u]; A[qq[u] * LD + r] = sn[u] * x[u] + cs[u] * y[u]; }
            }
          }
        }
        __syncthreads();
      } else {
      // (a pair with a padded index has apq == 0 exactly and is no rotation in LgSmoother::sym_singular either: the loops stop at mr)
      for (int p = 0; p < mr - 1; p++)
        for (int q = p + 1; q < mr; q++) {
          const double apq = A[p * LD + q], app = A[p * LD + p], aqq = A[q * LD + q], aqp = A[q * LD + p];
          // this lane's column pair (of A and of the eigenvector accumulator) and the row pair in its column, all as they are BEFORE the rotation
          const double x = A[r * LD + p], y = A[r * LD + q], wx = W[r * LD + p], wy = W[r * LD + q];
          double x2 = A[p * LD + r], y2 = A[q * LD + r];
          const bool act = !done && apq != 0.0;
          const double tau = (aqq - app) / (2.0 * (act ? apq : 1.0));
          const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
          const double cs = 1.0 / sqrt(1.0 + t * t), sn = t * cs;
          // the column update A[., p], A[., q] <- (cs x - sn y, sn x + cs y) leaves A[p][r], A[q][r] of a lane r outside {p, q} as read above;
          // in the columns p and q the row update reads what the column update made of the 2 x 2 block
          if (r == p) { x2 = cs * app - sn * apq; y2 = cs * aqp - sn * aqq; }
          if (r == q) { x2 = sn * app + cs * apq; y2 = sn * aqp + cs * aqq; }
          __syncthreads();
          if (act) {
            if (r != p && r != q) { A[r * LD + p] = cs * x - sn * y; A[r * LD + q] = sn * x + cs * y; }
            A[p * LD + r] = cs * x2 - sn * y2; A[q * LD + r] = sn * x2 + cs * y2;
            W[r * LD + p] = cs * wx - sn * wy; W[r * LD + q] = sn * wx + cs * wy;
          }
          __syncthreads();
        }
      }
    }
#ifdef LGF_STATS
    if (threadIdx.x == 0) atomicAdd(&lgf_stats[2], (unsigned long long)(__builtin_readcyclecounter() - lgf_c0));
#endif
    // singular values in stable descending order, as ranks: lane r's value goes to place #{larger} + #{equal and earlier}
    const double svr = fabs(A[r * LD + r]);
    __syncthreads();
    va[r] = svr;
    __syncthreads();
    int rank = 0;
    for (int i = 0; i < M; i++) { const double o = va[i]; rank += (o > svr || (o == svr && i < r)) ? 1 : 0; }
    __syncthreads();
    vd[rank] = svr; vb[rank] = (double)r;   // (a NaN on the diagonal leaves places unfilled: the result is non-finite and reported anyway)
    __syncthreads();
    // lane j owns output column j: its source column, its sign
    const int src = min(max((int)vb[r], 0), M - 1);
    int big = 0;
    for (int i = 1; i < M; i++) if (fabs(W[i * LD + src]) > fabs(W[big * LD + src])) big = i;
    const double sg = W[big * LD + src] < 0 ? -1.0 : 1.0;
    __syncthreads();
    va[r] = sg; vc[r] = (double)src;
    __syncthreads();
    // U[r][j] = sg_j W[r][src_j]: into A (the Jacobi matrix is spent)
#pragma unroll 4
    for (int jj = 0; jj < JW; jj++) A[r * LD + j0 + jj] = va[j0 + jj] * W[r * LD + (int)vc[j0 + jj]];
    __syncthreads();
  }

  // c[jj] = sum over q (ascending, from 0.0) of A[r][q] * W[q][j0 + jj]: the row's own operand in slot X, the broadcast operand in slot Y
  __device__ LG_INL void prod_xy(double (&c)[JW]) {
#pragma unroll
    for (int jj = 0; jj < JW; jj++) c[jj] = 0.0;
#pragma unroll 2
    for (int q = 0; q < M; q++) {
      const double a = A[r * LD + q];
#pragma unroll
      for (int jj = 0; jj < JW; jj++) c[jj] += a * W[q * LD + j0 + jj];
    }
  }


  // ================================================================================================================================
  // Round 6: the THIN path of a time point t >= 1.  The two decompositions of conditional_cov are what the set-up of 17 ... 64 states
  // spends its time in (2.9 ms per (draw, time point) at 53 states: 72 % of a seasonal post-correction), and neither needs a full Jacobi:
  //   * pinv(V_{t-1}) only ever multiplies the cross-covariance.  V_{t-1} is a smoothed covariance, positive definite unless a state is
  //     deterministic: a diagonal-pivoted Cholesky factor (rank-revealing: a pivot below eps m x the first one ends it) and two triangular
  //     solves per column of the cross-covariance give pinv(V_{t-1}) C_{t-1} in ~3 m^2 LDS operations per lane instead of ~10 sweeps
  //     of m - 1 rotation steps;  a rank-deficient V_{t-1} takes the full path below, decomposition and all (same numbers as before);
  //   * Var(alpha_t | alpha_{t-1}, y) has the rank of the noise of one step (3 of 53 for bsm_ng(period = 52)): a pivoted Cholesky factor
  //     L (m x r, r <= RT = 4; more: the full path) and the eigen-decomposition of the r x r Gram matrix L'L = Q diag(lambda) Q' give
  //     U sqrt(s) = L Q directly -- the same columns the full decomposition finds for the non-zero part, and EXACT zeros where it returns
  //     rounding noise (the cut every implementation applies since round 6: oracle/bssm_oracle.c, orc_conditional_cov).
  // Wave 0 does the sequential parts (the other waves keep the barriers); the products use every wave's columns.
  static constexpr int RT = 4;
  // sum of va-published diagonal entries (ascending): every lane passes the diagonal element it holds (or anything where it holds none)
  __device__ LG_INL double trace_of(const double mine) {
    __syncthreads();
    if (r >= j0 && r < j0 + JW) va[r] = (r < mr) ? mine : 0.0;
    __syncthreads();
    double tr = 0.0;
    for (int i = 0; i < M; i++) tr += va[i];
    __syncthreads();
    return tr;
  }
  // symmetric part of the matrix whose row r (this wave's columns of it) is passed in, into slot X
  __device__ LG_INL void store_sym(const double (&row)[JW]) {
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < JW; jj++) A[r * LD + j0 + jj] = row[jj];
    __syncthreads();
    double sym[JW];
#pragma unroll
    for (int jj = 0; jj < JW; jj++) sym[jj] = 0.5 * (row[jj] + A[(j0 + jj) * LD + r]);
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < JW; jj++) A[r * LD + j0 + jj] = sym[jj];
    __syncthreads();
  }
  // Diagonal-pivoted Cholesky of the symmetric matrix in slot X: `steps` columns of the factor into slot Y (W[i][j]: rows in their own
  // order, column j belongs to pivot j; columns behind the rank are zero), the pivot rows into vc[0 ... rank).  A pivot that is not
  // above the threshold (relative to the first pivot, or absolute) ends the factor; the loop keeps its trip count.  Returns the rank.
  __device__ LG_INL int pivoted_cholesky(const int steps, const double thr, const bool relative) {
    double dg = (r < mr) ? A[r * LD + r] : -1.0;
    bool picked = r >= mr, open = true;
    int rank = 0;
    double thr_abs = thr, piv0 = 1.0, pivmin = 1.0;
    for (int j = 0; j < steps; j++) {
      double* buf = (j & 1) ? vb : va;   // (two buffers: a wave may still scan the previous step's while wave 0 publishes this one's)
      if (w == 0) buf[r] = picked ? -1.0 : dg;
      __syncthreads();
      int best = 0;
      double dmax = buf[0];
#pragma unroll 16
      for (int i = 1; i < M; i++) { const double v = buf[i]; if (v > dmax) { dmax = v; best = i; } }
      if (j == 0 && relative) thr_abs = thr * dmax;
      open = open && dmax > thr_abs;
      if (relative) { if (j == 0) { piv0 = dmax; pivmin = dmax; } else if (open) pivmin = fmin(pivmin, dmax); }
      rank += open ? 1 : 0;
      if (w == 0) {
        double l = 0.0;
        if (open) {
          // (four partial sums: the loads of a step are independent, the subtractions of one sum are not)
          double c0 = A[r * LD + best], c1 = 0.0, c2 = 0.0, c3 = 0.0;
          int k = 0;
          for (; k + 4 <= j; k += 4) {
            c0 -= W[r * LD + k] * W[best * LD + k];
            c1 -= W[r * LD + k + 1] * W[best * LD + k + 1];
            c2 -= W[r * LD + k + 2] * W[best * LD + k + 2];
            c3 -= W[r * LD + k + 3] * W[best * LD + k + 3];
          }
          for (; k < j; k++) c0 -= W[r * LD + k] * W[best * LD + k];
          const double c = (c0 + c1) + (c2 + c3);
          l = picked ? 0.0 : c / sqrt(dmax);
          if (r == best) { l = sqrt(dmax); picked = true; }
          vc[j] = (double)best;
        }
        W[r * LD + j] = l;
        dg -= l * l;
      }
    }
    __syncthreads();
    if (relative) chol_ratio = piv0 > 0.0 ? pivmin / piv0 : 0.0;
    return rank;
  }
  // ---- Round 6, late: the sequential parts of the thin path in REGISTERS, on every wave (G == 1: one matrix per workgroup, M = 64).
  // Wave 0 alone ran the left-looking factorisation and the substitutions out of LDS while three waves kept the barriers: ~300 us per (draw,
  // time point).  Here every wave holds its JW columns of row r: the factorisation is right-looking (the Schur complement in registers: a pivot
  // step is one maximum over the wave, one column through LDS, one rank-1 update whose row operand comes by v_readlane), and the substitutions
  // are column-oriented -- a wave solves ITS columns of the right-hand side, no barrier at all.  Same pivots and the same factor up to the
  // order of the roundings (sums of the same products, subtracted one pivot at a time instead of in four partial sums).
  // 1 / sqrt(d) for d > 0 (v_rsq_f64 + three Newton steps: the last place or two), d * that = sqrt(d)
  __device__ LG_INL static double rsqrt_nr(const double d) {
    double y = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
#pragma unroll
    for (int it = 0; it < 3; it++) {
      const double e = __builtin_fma(-(h * y), y, 0.5);
      y = __builtin_fma(y, e, y);
    }
    return y;
  }
  __device__ LG_INL int pivoted_cholesky_regs(const int steps, const double thr, const bool relative) {
    static_assert(G == 1, "one matrix per workgroup: the pivot is wave-uniform");
    const int j0s = __builtin_amdgcn_readfirstlane(j0);
    double S[JW];
#pragma unroll
    for (int jj = 0; jj < JW; jj++) S[jj] = A[r * LD + j0 + jj];
    double dg = (r < mr) ? A[r * LD + r] : -1.0;
    bool picked = r >= mr, open = true;
    int rank = 0;
    double thr_abs = thr, piv0 = 1.0, pivmin = 1.0;
    for (int j = 0; j < steps; j++) {
      double* buf = (j & 1) ? vb : va;
      const double key = picked ? -1.0 : dg;
      const double dmax = wave_max_key(key);
      const unsigned long long at = __builtin_amdgcn_ballot_w64(key == dmax);
      const int best = at ? __builtin_ctzll(at) : 0;   // (the first of equals, as the scan of pivoted_cholesky; no lane: a NaN, carried to the end)
      if (j == 0 && relative) thr_abs = thr * dmax;
      open = open && dmax > thr_abs;
      if (relative) { if (j == 0) { piv0 = dmax; pivmin = dmax; } else if (open) pivmin = fmin(pivmin, dmax); }
      rank += open ? 1 : 0;
      if ((best / JW) == w) {   // this wave owns column `best` of the Schur complement (wave-uniform)
        double x = S[0];
#pragma unroll
        for (int jj = 1; jj < JW; jj++) x = ((best % JW) == jj) ? S[jj] : x;
        buf[r] = x;
      }
      __syncthreads();
      double l = 0.0;
      if (open) {
        const double rs = rsqrt_nr(dmax);
        l = picked ? 0.0 : buf[r] * rs;
        if (r == best) { l = dmax * rs; picked = true; }
      }
      if (w == 0) {
        W[r * LD + j] = l;
        if (r == 0 && open) vc[j] = (double)best;
      }
#pragma unroll
      for (int jj = 0; jj < JW; jj++) S[jj] -= l * lane_broadcast(l, j0s + jj);
      dg -= l * l;
    }
    __syncthreads();
    if (relative) chol_ratio = piv0 > 0.0 ? pivmin / piv0 : 0.0;
    return rank;
  }
  // X <- (L L')^{-1} X for this wave's JW columns: L (lower triangular, rows in pivot order) in slot X of LDS, the right-hand side's row r in
  // registers; forward, then backward substitution, column-oriented: a row is kept DIVIDED by its diagonal element, so that the component a
  // step finishes is simply lane a's row -- broadcast by v_readlane and eliminated from the other rows with the factor's entry divided the same
  // way (one multiplication per step; the entries on the other side of the diagonal are exact zeros, the row's own entry is masked).
  __device__ LG_INL void chol_solve_regs(double (&X)[JW]) {
    const double dinv = (r < mr) ? 1.0 / A[r * LD + r] : 0.0;
#pragma unroll
    for (int jj = 0; jj < JW; jj++) X[jj] *= dinv;
    for (int a = 0; a < mr; a++) {
      const double la = (r == a) ? 0.0 : A[r * LD + a] * dinv;
#pragma unroll
      for (int jj = 0; jj < JW; jj++) X[jj] -= la * lane_broadcast(X[jj], a);
    }
#pragma unroll
    for (int jj = 0; jj < JW; jj++) X[jj] *= dinv;
    for (int a = mr - 1; a >= 0; a--) {
      const double ua = (r == a) ? 0.0 : A[a * LD + r] * dinv;
#pragma unroll
      for (int jj = 0; jj < JW; jj++) X[jj] -= ua * lane_broadcast(X[jj], a);
    }
  }
  // Returns whether THIS GROUP's matrices need the full path (a singular V_{t-1}, or more than RT noise directions): a property of the
  // draw alone -- which draws share a workgroup must not show in anybody's bits (the result of a post-correction call is the same for
  // every device list).  Every group runs all of it (the barriers are the workgroup's); a group that needs the full path writes nothing
  // here, the others write their results and their `bad` flag.
  __device__ LG_INL bool thin_step(int t, const double* Pt, const double* cc, double* Vout, double* Cout, size_t es, size_t d, int* bad_out) {
    int bad = 0;
    double row[JW], cout[JW];
    // ---- pinv(V_{t-1}) C_{t-1} by a pivoted Cholesky factor
#pragma unroll
    for (int jj = 0; jj < JW; jj++) row[jj] = Pt[((size_t)(t - 1) * MM + r + M * (j0 + jj)) * es + d];
    store_sym(row);
    int rank1;
    if constexpr (G == 1) rank1 = pivoted_cholesky_regs(mr, kEps * mr, true);
    else rank1 = pivoted_cholesky(mr, kEps * mr, true);
    bool need_full = rank1 < mr;   // a singular V_{t-1}: what follows is computed (on a factor with zero columns: not finite) and discarded
    const int pr = (r < mr) ? min(max((int)vc[r], 0), M - 1) : r;   // the row of the matrix that is pivot r (clamped: places behind the rank of a singular matrix hold stale values)
#pragma unroll
    for (int jj = 0; jj < JW; jj++) row[jj] = (r < mr) ? W[pr * LD + j0 + jj] : 0.0;
    __syncthreads();   // every lane has read its pivot row of the factor (and vc)
    if (w == 0 && r < mr) vd[pr] = (double)r;      // inverse permutation: the place of row pr
#pragma unroll
    for (int jj = 0; jj < JW; jj++) A[r * LD + j0 + jj] = row[jj];    // slot X: the factor with its rows in pivot order (lower triangular)
    double c0[JW];   // C_{t-1}, rows in pivot order
#pragma unroll
    for (int jj = 0; jj < JW; jj++) c0[jj] = (r < mr) ? cc[((size_t)(t - 1) * MM + pr + M * (j0 + jj)) * es + d] : 0.0;
    if constexpr (G == 1) {
      __syncthreads();   // the factor is in slot X
      double xs[JW];
#pragma unroll
      for (int jj = 0; jj < JW; jj++) xs[jj] = c0[jj];
      chol_solve_regs(xs);
#pragma unroll
      for (int jj = 0; jj < JW; jj++) W[r * LD + j0 + jj] = xs[jj];   // slot Y: the solution (the factor's columns there are spent: every lane read its pivot row above)
    } else {
#pragma unroll
    for (int jj = 0; jj < JW; jj++) W[r * LD + j0 + jj] = c0[jj];  // slot Y: C_{t-1}, rows in pivot order
    __syncthreads();
    if (w == 0 && r < mr) {   // lane c solves column c: forward, then backward substitution, in place
      const int c = r;
      for (int a = 0; a < mr; a++) {
        double s0 = W[a * LD + c], s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int k = 0;
        for (; k + 4 <= a; k += 4) {
          s0 -= A[a * LD + k] * W[k * LD + c];
          s1 -= A[a * LD + k + 1] * W[(k + 1) * LD + c];
          s2 -= A[a * LD + k + 2] * W[(k + 2) * LD + c];
          s3 -= A[a * LD + k + 3] * W[(k + 3) * LD + c];
        }
        for (; k < a; k++) s0 -= A[a * LD + k] * W[k * LD + c];
        W[a * LD + c] = ((s0 + s1) + (s2 + s3)) / A[a * LD + a];
      }
      for (int a = mr - 1; a >= 0; a--) {
        double s0 = W[a * LD + c], s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int k = a + 1;
        for (; k + 4 <= mr; k += 4) {
          s0 -= A[k * LD + a] * W[k * LD + c];
          s1 -= A[(k + 1) * LD + a] * W[(k + 1) * LD + c];
          s2 -= A[(k + 2) * LD + a] * W[(k + 2) * LD + c];
          s3 -= A[(k + 3) * LD + a] * W[(k + 3) * LD + c];
        }
        for (; k < mr; k++) s0 -= A[k * LD + a] * W[k * LD + c];
        W[a * LD + c] = ((s0 + s1) + (s2 + s3)) / A[a * LD + a];
      }
    }
    }
    __syncthreads();
    // C_t = (pinv(V_{t-1}) C_{t-1})': element [r][j] is the solution's [place of j][r]
#pragma unroll
    for (int jj = 0; jj < JW; jj++) {
      const int j = j0 + jj;
      const int pl = min(max((int)vd[j], 0), M - 1);
      cout[jj] = (r < mr && j < mr) ? W[pl * LD + r] : 0.0;
    }
    // ---- Var(alpha_t | alpha_{t-1}, y) = V_t - C_t C_{t-1}: the cross-covariance again (rows in pivot order) into slot X
    // (every wave is through with the factor in slot X: the barrier behind the substitutions)
#pragma unroll
    for (int jj = 0; jj < JW; jj++) A[r * LD + j0 + jj] = c0[jj];
    __syncthreads();
    {
      double acc[JW];
#pragma unroll
      for (int jj = 0; jj < JW; jj++) acc[jj] = 0.0;
#pragma unroll 2
      for (int a = 0; a < mr; a++) {
        const double y = W[a * LD + r];
#pragma unroll
        for (int jj = 0; jj < JW; jj++) acc[jj] += y * A[a * LD + j0 + jj];
      }
      double dgc = 0.0;
#pragma unroll
      for (int jj = 0; jj < JW; jj++) {
        const double pt = Pt[((size_t)t * MM + r + M * (j0 + jj)) * es + d];
        if (j0 + jj == r) dgc = pt;
        row[jj] = (r < mr) ? pt - acc[jj] : 0.0;
      }
      const double trc = trace_of(dgc);
      const double thr = kEps * mr * trc;
      store_sym(row);
      const int rank2 = pivoted_cholesky(RT + 1, thr, false);
      need_full = need_full || rank2 > RT;   // more noise directions than the thin form takes
      // ---- Gram matrix of the factor's RT columns (every lane the same sums), its eigen-decomposition by cyclic Jacobi in registers
      double g[RT][RT], q[RT][RT];
#pragma unroll
      for (int a = 0; a < RT; a++)
#pragma unroll
        for (int b = 0; b < RT; b++) { g[a][b] = 0.0; q[a][b] = a == b ? 1.0 : 0.0; }
      for (int i = 0; i < mr; i++) {
        double l[RT];
#pragma unroll
        for (int a = 0; a < RT; a++) l[a] = W[i * LD + a];
#pragma unroll
        for (int a = 0; a < RT; a++)
#pragma unroll
          for (int b = a; b < RT; b++) g[a][b] += l[a] * l[b];
      }
#pragma unroll
      for (int a = 0; a < RT; a++)
#pragma unroll
        for (int b = 0; b < a; b++) g[a][b] = g[b][a];
      for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0, dgs = 0.0;
#pragma unroll
        for (int b = 0; b < RT; b++) {
#pragma unroll
          for (int a = 0; a < b; a++) off += g[a][b] * g[a][b];
          dgs += g[b][b] * g[b][b];
        }
        if (off < 1e-300 || off < 1e-60 * dgs) break;
#pragma unroll
        for (int p = 0; p < RT - 1; p++)
#pragma unroll
          for (int qq = p + 1; qq < RT; qq++) {
            const double apq = g[p][qq];
            if (apq == 0.0) continue;
            const double tau = (g[qq][qq] - g[p][p]) / (2.0 * apq);
            const double tt = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
            const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = tt * cs;
#pragma unroll
            for (int i = 0; i < RT; i++) { const double x = g[i][p], y = g[i][qq]; g[i][p] = cs * x - sn * y; g[i][qq] = sn * x + cs * y; }
#pragma unroll
            for (int i = 0; i < RT; i++) { const double x = g[p][i], y = g[qq][i]; g[p][i] = cs * x - sn * y; g[qq][i] = sn * x + cs * y; }
#pragma unroll
            for (int i = 0; i < RT; i++) { const double x = q[i][p], y = q[i][qq]; q[i][p] = cs * x - sn * y; q[i][qq] = sn * x + cs * y; }
          }
      }
      // eigenvalues in stable descending order (ranks), this row of L Q in that order
      double lam[RT], f[RT], lam_at[RT];
#pragma unroll
      for (int a = 0; a < RT; a++) lam[a] = fabs(g[a][a]);
      double lrow[RT];
#pragma unroll
      for (int a = 0; a < RT; a++) lrow[a] = (r < mr) ? W[r * LD + a] : 0.0;
#pragma unroll
      for (int c = 0; c < RT; c++) { f[c] = 0.0; lam_at[c] = 0.0; }
#pragma unroll
      for (int a = 0; a < RT; a++) {
        int rk = 0;
#pragma unroll
        for (int b = 0; b < RT; b++) rk += (lam[b] > lam[a] || (lam[b] == lam[a] && b < a)) ? 1 : 0;
        double fa = 0.0;
#pragma unroll
        for (int b = 0; b < RT; b++) fa += lrow[b] * q[b][a];
#pragma unroll
        for (int c = 0; c < RT; c++) if (rk == c) { f[c] = fa; lam_at[c] = lam[a]; }
      }
      // sign of every column: its largest component positive (the first of equals)
      double sg[RT];
      if constexpr (G == 1) {   // (one matrix per wave row set: the column's largest |component| is a maximum over the wave, lane r holding row r's)
#pragma unroll
        for (int c = 0; c < RT; c++) {
          const double k = fabs(f[c]);
          const double mx = wave_max_key(k);
          const unsigned long long at = __builtin_amdgcn_ballot_w64(k == mx);
          const int big = at ? __builtin_ctzll(at) : 0;
          sg[c] = lane_broadcast(f[c], big) < 0 ? -1.0 : 1.0;
        }
      } else {
      __syncthreads();   // every lane is through with the factor's columns in slot Y and with the Jacobi matrix in slot X
      if (w == 0) {
#pragma unroll
        for (int c = 0; c < RT; c++) A[r * LD + c] = f[c];
      }
      __syncthreads();
#pragma unroll
      for (int c = 0; c < RT; c++) {
        int big = 0;
#pragma unroll 4
        for (int i = 1; i < M; i++) if (fabs(A[i * LD + c]) > fabs(A[big * LD + c])) big = i;
        sg[c] = A[big * LD + c] < 0 ? -1.0 : 1.0;
      }
      }
#pragma unroll
      for (int jj = 0; jj < JW; jj++) {
        const int j = j0 + jj;
        double v = 0.0;
#pragma unroll
        for (int c = 0; c < RT; c++) if (j == c && lam_at[c] > thr) v = sg[c] * f[c];
        if (!need_full && !skip) {
          Vout[((size_t)t * MM + r + M * j) * es + d] = v;
          Cout[((size_t)t * MM + r + M * j) * es + d] = cout[jj];
          bad |= !lg_finite(v) || !lg_finite(cout[jj]);
        }
      }
    }
    *bad_out = bad;
    return need_full && !skip;   // (a draw that has already failed is discarded whatever is computed for it)
  }

  // conditional_cov at time point t of draw d (arrays draw-interleaved, element k at ptr[k * es + d]); returns whether everything written is finite
  // Round 6, second half: the thin path and the full path are TWO kernels (the set-up launches the thin one over the time points >= 1, then the
  // full one over everything): in one kernel the register allocation was the union of both (659 registers spilt at M = 64, 2.2 KB of scratch
  // per lane, 320 us per thin time point instead of ~80).  `flag` (one int per (draw, time point), written by cond_step_thin) says which groups
  // the full kernel still has to do; flag == nullptr: the full path at this time point whatever the matrices are (BSSM_AMD_FACTORS_FULL, and t = 0).
  __device__ LG_INL bool cond_step_thin(int t, const double* Pt, const double* cc, double* Vout, double* Cout, size_t es, size_t d, int* flag, double* ratio_out = nullptr) {
    int bad_thin = 0;
    const bool mine_full = thin_step(t, Pt, cc, Vout, Cout, es, d, &bad_thin);
    if (r == 0 && w == 0) { *flag = mine_full ? 1 : 0; if (ratio_out) *ratio_out = chol_ratio; }
    if (threadIdx.x == 0) atomicAdd(&lgf_thin_counts[1], 1ull);
    return reduce_bad(mine_full ? 0 : bad_thin);
  }
  __device__ LG_INL bool cond_step(int t, const double* Pt, const double* cc, double* Vout, double* Cout, size_t es, size_t d, const int* flag = nullptr) {
    int bad = 0;
    double row[JW];
#ifdef LGF_STATS
    const unsigned long long lgf_t0 = __builtin_readcyclecounter();
#endif
    if (t == 0) {
      double dg0 = 0.0;
#pragma unroll
      for (int jj = 0; jj < JW; jj++) { row[jj] = Pt[(size_t)(r + M * (j0 + jj)) * es + d]; if (j0 + jj == r) dg0 = row[jj]; }
      const double tr0 = trace_of(dg0);
      sym_singular(row);
#pragma unroll
      for (int jj = 0; jj < JW; jj++) {
        const double sv = vd[j0 + jj];
        const double v = sv > kEps * mr * tr0 ? A[r * LD + j0 + jj] * sqrt(sv) : 0.0;   // (the cut of orc_conditional_cov, oracle/bssm_oracle.c)
        Vout[(size_t)(r + M * (j0 + jj)) * es + d] = v;
        bad |= !lg_finite(v);
      }
      return reduce_bad(bad);
    }
    bool mine_full = true;   // this group's results come from the full path below
    int bad_thin = 0;
    if (flag) {   // the thin kernel ran before: only the groups it flagged are done here (every group of the workgroup runs the code: barriers)
      mine_full = *flag != 0;
      const bool any_full = __syncthreads_or(mine_full) != 0;
      if (threadIdx.x == 0 && any_full) atomicAdd(&lgf_thin_counts[0], 1ull);
      if (!any_full) return true;
    }
#pragma unroll
    for (int jj = 0; jj < JW; jj++) row[jj] = Pt[((size_t)(t - 1) * MM + r + M * (j0 + jj)) * es + d];
    sym_singular(row);   // U in X, |eigenvalues| in vd
    // pseudo-inverse W2 = sum over the kept components c of U[:, c] U[:, c]' / s_c, ascending c
    const double s0 = vd[0];
#pragma unroll
    for (int jj = 0; jj < JW; jj++) row[jj] = 0.0;
#pragma unroll 1
    for (int c = 0; c < M; c++) {
      const double sc = vd[c];
      if (!(sc > kEps * M * s0)) continue;
      const double urc = A[r * LD + c];
#pragma unroll
      for (int jj = 0; jj < JW; jj++) row[jj] += urc * A[(j0 + jj) * LD + c] / sc;
    }
    __syncthreads();   // every lane is through with U (and long through with the eigenvector accumulator)
#pragma unroll
    for (int jj = 0; jj < JW; jj++) W[r * LD + j0 + jj] = row[jj];
    // tmp = Cp' W2 (row r: sum over q of Cp[q][r] W2[q][j]): column r of the cross-covariance is the row's operand
#pragma unroll
    for (int jj = 0; jj < JW; jj++) row[jj] = cc[((size_t)(t - 1) * MM + (j0 + jj) + M * r) * es + d];
#pragma unroll
    for (int jj = 0; jj < JW; jj++) A[r * LD + j0 + jj] = row[jj];
    __syncthreads();
    prod_xy(row);
#pragma unroll
    for (int jj = 0; jj < JW; jj++) {
      if (mine_full) { Cout[((size_t)t * MM + r + M * (j0 + jj)) * es + d] = row[jj]; bad |= !lg_finite(row[jj]); }
    }
    // Vc - tmp Cp: tmp's rows into X, the cross-covariance's rows into Y
    __syncthreads();   // every lane is through with W2 and with the operand rows
#pragma unroll
    for (int jj = 0; jj < JW; jj++) A[r * LD + j0 + jj] = row[jj];
#pragma unroll
    for (int jj = 0; jj < JW; jj++) row[jj] = cc[((size_t)(t - 1) * MM + r + M * (j0 + jj)) * es + d];
#pragma unroll
    for (int jj = 0; jj < JW; jj++) W[r * LD + j0 + jj] = row[jj];
    __syncthreads();
    prod_xy(row);
    double dgc = 0.0;
#pragma unroll
    for (int jj = 0; jj < JW; jj++) {
      const double pt = Pt[((size_t)t * MM + r + M * (j0 + jj)) * es + d];
      if (j0 + jj == r) dgc = pt;
      row[jj] = pt - row[jj];
    }
    const double trc = trace_of(dgc);
    sym_singular(row);
#pragma unroll
    for (int jj = 0; jj < JW; jj++) {
      const double sv = vd[j0 + jj];
      const double v = sv > kEps * mr * trc ? A[r * LD + j0 + jj] * sqrt(sv) : 0.0;
      if (mine_full) { Vout[((size_t)t * MM + r + M * (j0 + jj)) * es + d] = v; bad |= !lg_finite(v); }
    }
#ifdef LGF_STATS
    if (threadIdx.x == 0) { atomicAdd(&lgf_stats[3], (unsigned long long)(__builtin_readcyclecounter() - lgf_t0)); atomicAdd(&lgf_stats[4], 1ull); }
#endif
    return reduce_bad(mine_full ? bad : bad_thin);
  }

  // any lane of the group bad -> false in every lane of the group
  __device__ LG_INL bool reduce_bad(int bad) {
    __syncthreads();
    vc[r] = 0.0;
    __syncthreads();
    if (bad) vc[r] = 1.0;
    __syncthreads();
    double any = 0.0;
    for (int j = 0; j < M; j++) any += vc[j];
    return any == 0.0;
  }
};

}  // namespace bssm
