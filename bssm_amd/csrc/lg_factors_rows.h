// lg_factors_rows.h -- the conditional factors of the psi set-up (LgSmoother<M>::cond_step: conditional_cov with use_svd = true,
// src/conditional_dist.cpp:4-31) for states of 16 and more dimensions, ROW-DISTRIBUTED: M lanes share one (draw, time point), lane r owns
// row r, and every M x M matrix of the step lives in LDS (rows padded to M + 1 doubles: a column read is as conflict-free as a row read).
//
// Why: with one lane per (draw, time point) the work matrices of a 16 x 16 step are 18 KB of scratch memory per LANE (74 KB at 32 x 32),
// every element access a dependent round trip -- 172 ms per 2 048 monthly-seasonal draws, 83 % of that set-up -- and 32 x 32 does not run at
// all.  Here a cyclic Jacobi rotation (p, q) is three LDS phases that are parallel over the M lanes: lane r rotates the column pair of ITS row
// (A[r][p], A[r][q] and the eigenvector rows W[r][p], W[r][q]), then lane c rotates the row pair in ITS column (A[p][c], A[q][c]).
//
// The arithmetic follows LgSmoother<M>::sym_singular / cond_step term by term: the same rotations in the same cyclic order from the same
// formulas, products summed over the inner index in ascending order, the same sign and ordering conventions.  Two differences that do not
// change a needed result: the off-diagonal / diagonal masses of the stopping test are summed row by row (they only decide when to stop, and
// a further sweep is a no-op in double precision), and a non-finite value is carried to the end and reported there.
// Device only; one wave per workgroup (blockDim.x = 64); every lane of the wave must call it.
#pragma once
#include "lg_smoother.h"

namespace bssm {

template <int M>
struct LgFactorRows {
  static_assert(64 % M == 0 && M >= 16, "M lanes per matrix must tile the wave");
  static constexpr int MM = M * M, G = 64 / M, LD = M + 1, MAT = M * LD;
  static constexpr int GS = 3 * MAT + 4 * M + 2;   // per group: three matrices, four vectors (+ 2: groups start in different banks)
  static constexpr size_t kLdsDoubles = (size_t)G * GS;
  static constexpr double kEps = DBL_EPSILON;

  double *A, *W, *C;        // this group's matrices: A (Jacobi work matrix -> U), W (eigenvector accumulator -> pseudo-inverse), C (cross-covariance)
  double *va, *vb, *vc, *vd;  // this group's vectors
  int r;                    // the row (and, in the row phase of a rotation, the column) this lane owns
  bool skip = false;        // this group's draw has already failed: its decompositions take no rotations (the result is discarded)
  __device__ LgFactorRows(double* lds_wave) {
    const int lane = threadIdx.x & 63;
    r = lane % M;
    double* base = lds_wave + (size_t)(lane / M) * GS;
    A = base; W = base + MAT; C = base + 2 * MAT;
    va = base + 3 * MAT; vb = va + M; vc = vb + M; vd = vc + M;
  }

  // |eigenvalues| (descending) in s[], eigenvectors U (sign: the largest component of every vector positive) left in A, of the symmetric part
  // of the matrix whose row r this lane passes in.  Uses A, W and the four vectors; C is untouched.
  __device__ LG_INL void sym_singular(const double (&row)[M], double (&s)[M]) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < M; j++) { A[r * LD + j] = row[j]; W[r * LD + j] = (j == r) ? 1.0 : 0.0; }
    __syncthreads();
    double sym[M];
#pragma unroll
    for (int j = 0; j < M; j++) sym[j] = 0.5 * (row[j] + A[j * LD + r]);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < M; j++) A[r * LD + j] = sym[j];
    __syncthreads();
    bool done = skip;
    for (int sweep = 0; sweep < 60; sweep++) {
      // stopping test of the sweep (LgSmoother::sym_singular): off-diagonal mass below 1e-60 of the diagonal's
      double po = 0.0;
      for (int j = r + 1; j < M; j++) { const double x = A[r * LD + j]; po += x * x; }
      const double dr = A[r * LD + r];
      __syncthreads();
      va[r] = po; vb[r] = dr * dr;
      __syncthreads();
      double off = 0.0, dg = 0.0;
      for (int j = 0; j < M; j++) { off += va[j]; dg += vb[j]; }
      done = done || off < 1e-300 || off < 1e-60 * dg;
      if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;   // (every group of the wave has converged)
      for (int p = 0; p < M - 1; p++)
        for (int q = p + 1; q < M; q++) {
          const double apq = A[p * LD + q], app = A[p * LD + p], aqq = A[q * LD + q];
          const bool act = !done && apq != 0.0;
          const double tau = (aqq - app) / (2.0 * (act ? apq : 1.0));
          const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
          const double cs = 1.0 / sqrt(1.0 + t * t), sn = t * cs;
          // column pair of this lane's row, of A and of the eigenvector accumulator
          const double x = A[r * LD + p], y = A[r * LD + q], wx = W[r * LD + p], wy = W[r * LD + q];
          __syncthreads();
          if (act) {
            A[r * LD + p] = cs * x - sn * y; A[r * LD + q] = sn * x + cs * y;
            W[r * LD + p] = cs * wx - sn * wy; W[r * LD + q] = sn * wx + cs * wy;
          }
          __syncthreads();
          // row pair in this lane's column
          const double x2 = A[p * LD + r], y2 = A[q * LD + r];
          __syncthreads();
          if (act) { A[p * LD + r] = cs * x2 - sn * y2; A[q * LD + r] = sn * x2 + cs * y2; }
          __syncthreads();
        }
    }
    // singular values, stable insertion sort descending (every lane the same sort on the same values), sign convention
    double sv[M];
    int ord[M];
#pragma unroll
    for (int j = 0; j < M; j++) { sv[j] = fabs(A[j * LD + j]); ord[j] = j; }
    for (int a = 1; a < M; a++) {
      const int o = ord[a];
      const double so = sv[o];
      int b = a - 1;
      while (b >= 0 && sv[ord[b]] < so) { ord[b + 1] = ord[b]; b--; }
      ord[b + 1] = o;
    }
    // (ord[] and sv[] are indexed dynamically: they live in scratch memory, M entries each -- the sort is M^2 / 4 steps once per call)
    // lane j owns output column j: its source column, its sign
    int src = 0;
    for (int j = 0; j < M; j++) { s[j] = sv[ord[j]]; if (j == r) src = ord[j]; }
    int big = 0;
    for (int i = 1; i < M; i++) if (fabs(W[i * LD + src]) > fabs(W[big * LD + src])) big = i;
    const double sg = W[big * LD + src] < 0 ? -1.0 : 1.0;
    __syncthreads();
    va[r] = sg; vb[r] = (double)src;
    __syncthreads();
    // U[r][j] = sg_j W[r][ord_j]: into A (the Jacobi matrix is spent)
    double urow[M];
#pragma unroll
    for (int j = 0; j < M; j++) urow[j] = va[j] * W[r * LD + (int)vb[j]];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < M; j++) A[r * LD + j] = urow[j];
    __syncthreads();
  }

  // conditional_cov at time point t of draw d (arrays draw-interleaved, element k at ptr[k * es + d]); returns whether everything written is finite
  __device__ LG_INL bool cond_step(int t, const double* Pt, const double* cc, double* Vout, double* Cout, size_t es, size_t d) {
    int bad = 0;
    double row[M], s[M];
    if (t == 0) {
#pragma unroll
      for (int j = 0; j < M; j++) row[j] = Pt[(size_t)(r + M * j) * es + d];
      sym_singular(row, s);
#pragma unroll
      for (int j = 0; j < M; j++) {
        const double v = A[r * LD + j] * sqrt(s[j]);
        Vout[(size_t)(r + M * j) * es + d] = v;
        bad |= !lg_finite(v);
      }
      return reduce_bad(bad);
    }
    double vc_row[M];
#pragma unroll
    for (int j = 0; j < M; j++) {
      row[j] = Pt[((size_t)(t - 1) * MM + r + M * j) * es + d];
      vc_row[j] = Pt[((size_t)t * MM + r + M * j) * es + d];
      C[r * LD + j] = cc[((size_t)(t - 1) * MM + r + M * j) * es + d];
    }
    sym_singular(row, s);   // U in A
    // pseudo-inverse W2 = sum over the kept components c of U[:, c] U[:, c]' / s_c, ascending c (into W)
    double acc[M];
#pragma unroll
    for (int j = 0; j < M; j++) acc[j] = 0.0;
    for (int c = 0; c < M; c++) {
      if (!(s[c] > kEps * M * s[0])) continue;
      const double urc = A[r * LD + c], sc = s[c];
#pragma unroll
      for (int j = 0; j < M; j++) acc[j] += urc * A[j * LD + c] / sc;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < M; j++) W[r * LD + j] = acc[j];
    __syncthreads();
    // tmp = Cp' W2 (row r: sum over q of Cp[q][r] W2[q][j]), tmp2 = tmp Cp
    double tmp[M];
#pragma unroll
    for (int j = 0; j < M; j++) {
      double sacc = 0.0;
      for (int q = 0; q < M; q++) sacc += C[q * LD + r] * W[q * LD + j];
      tmp[j] = sacc;
    }
#pragma unroll
    for (int j = 0; j < M; j++) {
      double sacc = 0.0;
#pragma unroll
      for (int q = 0; q < M; q++) sacc += tmp[q] * C[q * LD + j];
      vc_row[j] -= sacc;
    }
#pragma unroll
    for (int j = 0; j < M; j++) {
      Cout[((size_t)t * MM + r + M * j) * es + d] = tmp[j];
      bad |= !lg_finite(tmp[j]);
    }
    sym_singular(vc_row, s);
#pragma unroll
    for (int j = 0; j < M; j++) {
      const double v = A[r * LD + j] * sqrt(s[j]);
      Vout[((size_t)t * MM + r + M * j) * es + d] = v;
      bad |= !lg_finite(v);
    }
    return reduce_bad(bad);
  }

  // any lane of the group bad -> false in every lane of the group
  __device__ LG_INL bool reduce_bad(int bad) {
    __syncthreads();
    vc[r] = bad ? 1.0 : 0.0;
    __syncthreads();
    double any = 0.0;
    for (int j = 0; j < M; j++) any += vc[j];
    return any == 0.0;
  }
};

}  // namespace bssm
