// pf_canon.h -- the ONE normalisation / resampling arithmetic of every filter that fits a workgroup (N <= 8192 particles).
//
// Two kernel families run such a filter: pf_wg_kernel (one workgroup per filter, particle system in LDS, the whole time loop in
// one launch) and pf_row_step (pf_row_step.h: the filter spread over N / 64 waves, one launch per time step -- the shape a
// sparsely filled LAST round of a post-correction shard goes through, src/approx_mcmc.cpp:364-417).  Which of the two a draw
// lands in depends on the shard layout, so both must produce the SAME BITS for a draw (round-4 review, item 1): everything
// that involves more than one particle is defined here, as a function of the particle index alone --
//
//   row g = particles 64 g ... 64 g + 63 (the lanes of one wave):
//     M_g  = max_i w_i                                 (exact: the largest weight of a time point is exp(0) = 1)
//     e_i  = tl_exp(w_i - M_g)                         (a row without mass: M_g = -inf, e = 0)
//     sc_i = inclusive scan of e over the row, fixed DPP shape (wave_scan_incl);  S_g = sc_63
//   filter (lane l of a wave holds rows l, 64 + l, ...: up to 128 rows in the workgroup kernels, 256 in the row family):
//     M    = max_g M_g;  scale_g = fast_exp(M_g - M);  Q_g = inclusive scan of S_g scale_g (fixed shape: two wave scans)
//     S    = Q_last;  log-likelihood increment M + tl_log(S / N)           (src/model_ssm_ung.cpp:675-683)
//     weights_i = e_i scale_g                                               (exp(w_i - M), :676)
//   stratified resampling (src/stratified_sample.cpp:9-28), slot j with uniform u_j:
//     v = (u_j + j) (S / N);  source row e = #{g : Q_g < v} (clamped);  vp = (v - Q_{e-1}) / scale_e;
//     ancestor = 64 e + min(#{k : sc_{64 e + k} < vp}, row length - 1)
//
// The comparisons run in the source row's own scale (the scans are never rescaled), so neither family materialises a global
// CDF and both decide every ancestor from the same operands.  Multiplications that feed an addition are kept apart from it
// (an opaque register barrier) so that no compiler context can contract one family's into a fused multiply-add.
#pragma once
#include "pf_device.h"

namespace bssm {

constexpr int CANON_ROW = 64;        // particles per row block
constexpr int CANON_MAX_ROWS = 128;  // rows of a workgroup-resident filter (8192 particles): two prefix slots per lane
constexpr int ROW_NP = 4;            // prefix slots per lane of the row family: filters of up to 256 rows (16 384 particles)
constexpr int ROW_MAX_ROWS = 64 * ROW_NP;

// Row prefixes of a filter as the lanes of a wave hold them: lane l has the inclusive, scaled prefix (and 1 / scale) of rows
// l, 64 + l, ... 64 (NP - 1) + l.  NP (1, 2 or 4 slots per lane: filters of up to 64, 128, 256 rows) is a property of the KERNEL,
// its capacity; the values of the rows a filter has are the same for every NP that holds them.
template <int NP>
struct CanonPrefix {
  double Q[NP], I[NP];
  int B;
  __device__ __forceinline__ int count_below(const double v, const int lane) const {  // number of row prefixes below v
    int c = __builtin_popcountll(__builtin_amdgcn_ballot_w64(lane < B && Q[0] < v));
#pragma unroll
    for (int k = 1; k < NP; k++)
      if (B > 64 * k) c += __builtin_popcountll(__builtin_amdgcn_ballot_w64(lane + 64 * k < B && Q[k] < v));  // wave-uniform
    return c;
  }
  __device__ __forceinline__ double q(const int e) const {  // e wave-uniform
    double r = lane_broadcast(Q[0], e & 63);
#pragma unroll
    for (int k = 1; k < NP; k++) { if (B > 64 * k) { const double rk = lane_broadcast(Q[k], e & 63); r = (e >> 6) == k ? rk : r; } }  // (wave-uniform: slots a filter does not have cost a scalar branch)
    return r;
  }
  __device__ __forceinline__ double is(const int e) const {
    double r = lane_broadcast(I[0], e & 63);
#pragma unroll
    for (int k = 1; k < NP; k++) { if (B > 64 * k) { const double rk = lane_broadcast(I[k], e & 63); r = (e >> 6) == k ? rk : r; } }
    return r;
  }
};

__device__ __forceinline__ double canon_opaque(double x) {
  __asm__ volatile("" : "+v"(x));
  return x;
}

// statistics of the row this wave's lanes hold (w = -inf in lanes without a particle)
struct CanonRow { double e, sc, M; };
__device__ __forceinline__ CanonRow canon_row_stats(const double w) {
  const double M = wave_max_key(w);
  const double Ms = M == -INFINITY ? 0.0 : M;  // a row without mass: exp(-inf - 0) = 0 instead of exp(nan)
  const double e = tl_exp(w - Ms);
  const double sc = wave_scan_incl(e);
  return CanonRow{e, sc, M};
}

// combine of the (M_g, S_g) partials; lane l passes rows l, 64 + l, ... in pmax[k], psum[k], identities (-inf, 0) beyond the
// filter's B rows.  Every lane returns the same M and S.
template <int NP>
struct CanonComb {
  CanonPrefix<NP> px;
  double sc[NP];  // scale exp(M_g - M) of rows 64 k + l
  double M, S;
};
template <int NP>
__device__ __forceinline__ CanonComb<NP> canon_combine(const double (&pmax)[NP], const double (&psum)[NP], const int B) {
  CanonComb<NP> c;
  double mx = pmax[0];
#pragma unroll
  for (int k = 1; k < NP; k++) mx = fmax(mx, pmax[k]);
  c.M = wave_max_key(mx);
  c.px.B = B;
  double run = 0.0;  // total of the slots in front (every lane)
#pragma unroll
  for (int k = 0; k < NP; k++) {
    if (k == 0 || B > 64 * k) {  // wave-uniform
      c.sc[k] = fast_exp(pmax[k] - c.M);  // rows beyond B: 0; a poisoned (NaN) partial makes S NaN
      const double Qk = wave_scan_incl(canon_opaque(psum[k] * c.sc[k]));
      c.px.Q[k] = k == 0 ? Qk : run + Qk;
      c.px.I[k] = 1.0 / c.sc[k];
    } else {
      c.sc[k] = 0.0;
      c.px.Q[k] = run;
      c.px.I[k] = 0.0;
    }
    run = lane_broadcast(c.px.Q[k], 63);
  }
  c.S = run;
  return c;
}
template <int NP>
__device__ __forceinline__ double canon_row_scale(const CanonComb<NP>& c, const int g) {  // g wave-uniform
  double r = lane_broadcast(c.sc[0], g & 63);
#pragma unroll
  for (int k = 1; k < NP; k++) { const double rk = lane_broadcast(c.sc[k], g & 63); r = (g >> 6) == k ? rk : r; }
  return r;
}
// the combine of one wave handed to the other waves of a workgroup through LDS: [3 NP x 64 + 2] doubles
template <int NP>
__device__ __forceinline__ void canon_publish(const CanonComb<NP>& c, double* s_comb, const int lane) {
#pragma unroll
  for (int k = 0; k < NP; k++) {
    s_comb[(3 * k) * 64 + lane] = c.px.Q[k];
    s_comb[(3 * k + 1) * 64 + lane] = c.px.I[k];
    s_comb[(3 * k + 2) * 64 + lane] = c.sc[k];
  }
  if (lane == 0) { s_comb[3 * NP * 64] = c.M; s_comb[3 * NP * 64 + 1] = c.S; }
}
template <int NP>
__device__ __forceinline__ CanonPrefix<NP> canon_prefix_load(const double* s_comb, const int lane, const int B) {
  CanonPrefix<NP> px;
#pragma unroll
  for (int k = 0; k < NP; k++) { px.Q[k] = s_comb[(3 * k) * 64 + lane]; px.I[k] = s_comb[(3 * k + 1) * 64 + lane]; }
  px.B = B;
  return px;
}
template <int NP> __device__ __forceinline__ double canon_published_scale(const double* s_comb, const int g) { return s_comb[(3 * (g >> 6) + 2) * 64 + (g & 63)]; }
template <int NP> __device__ __forceinline__ double canon_published_M(const double* s_comb) { return s_comb[3 * NP * 64]; }
template <int NP> __device__ __forceinline__ double canon_published_S(const double* s_comb) { return s_comb[3 * NP * 64 + 1]; }
// log-likelihood increment of a time point (src/model_ssm_ung.cpp:683) and the stratum width S / N
__device__ __forceinline__ double canon_stratum(const double S, const int N) { return S / (double)N; }
__device__ __forceinline__ double canon_llinc(const double M, const double c_scale) { return M + canon_opaque(tl_log(c_scale)); }
// stratified target of output slot j
__device__ __forceinline__ double canon_target(const uint32_t uword, const int j, const double c_scale) {
  const double pos = canon_opaque((double)uword * 0x1.0p-32 + (double)j);  // (the product is exact: fused or not, the same bits)
  return pos * c_scale;
}

// Source row of every active slot of the wave and the target in that row's own scale.  The 64 slots of a wave are consecutive,
// so their targets increase along the lanes: two ballots bound the rows [c_lo, c_hi] the wave draws from (wave-uniform), and a
// lane walks the few prefixes in between.
struct CanonSource { int e; double vp; int c_lo, c_hi; };
template <int NP>
__device__ __forceinline__ CanonSource canon_source_row(const double v, const bool act, const CanonPrefix<NP>& px, const int lane) {
  const unsigned long long amask = __builtin_amdgcn_ballot_w64(act);
  CanonSource s{0, 0.0, 0, -1};
  if (amask == 0ull) return s;
  const int l_first = __builtin_ctzll(amask), l_last = 63 - __builtin_clzll(amask);
  const double v_lo = lane_broadcast(v, l_first), v_hi = lane_broadcast(v, l_last);
  s.c_lo = min(px.count_below(v_lo, lane), px.B - 1);
  s.c_hi = min(px.count_below(v_hi, lane), px.B - 1);
  int e = s.c_lo;
  double Pm1 = s.c_lo > 0 ? px.q(s.c_lo - 1) : 0.0, isv = px.is(s.c_lo);
  for (int k = s.c_lo; k < s.c_hi; k++) {  // wave-uniform bounds; Q is non-decreasing, so `below` is monotone in k
    const double qk = px.q(k), ik = px.is(k + 1);
    const bool below = qk < v;
    e = below ? k + 1 : e;
    Pm1 = below ? qk : Pm1;
    isv = below ? ik : isv;
  }
  s.e = e;
  s.vp = canon_opaque(v - Pm1) * isv;
  return s;
}

// number of entries of row[0, slen) below vp (entries non-decreasing, slen <= 64), capped at 63
__device__ __forceinline__ int canon_count_below(const double* row, const int slen, const double vp) {
  int lo = 0;
#pragma unroll
  for (int step = 32; step >= 1; step >>= 1) {
    const int idx = lo + step - 1;
    if (idx < slen && row[idx] < vp) lo += step;
  }
  return lo;
}

}  // namespace bssm
