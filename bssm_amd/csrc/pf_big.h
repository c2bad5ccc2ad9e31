// pf_big.h -- particle filters of the linear-state models with 17 ... 64 state dimensions (bootstrap and psi-APF;
// src/model_ssm_ung.cpp:511-712, src/model_ssm_mng.cpp:461-668).
//
// The reference has no limit on the state dimension (src/model_ssm_ung.cpp:9-32 reads any m); bsm_ng / bsm_lg with a seasonal
// component of period 24 or 52 have m = 1 + slope + (period - 1) + noise = 25 ... 54 states (R/models.R:995-997).  The kernels of
// pf_filter_impl.h keep "my particle, all components" in the registers of a lane, which ends at 16 components.  Here the
// transition
//     X' (MP x 64) = C + [T R] (MP x (m + k)) . [A ; Z] ((m + k) x 64)          alpha_{t+1} = C_t + T_t alpha_t + R_t eta_t (:663-670)
// of the 64 particles of a wave is a real dense contraction and runs on the matrix cores in their NATIVE fragment layout
// (v_mfma_f64_16x16x4_f64: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15], D[row = (lane >> 4) + 4 q][col = lane & 15]):
//   * the particle system is COMPONENT-MAJOR ([component][particle], in LDS or in HBM), so a B fragment is "component 4 kk + kq of
//     the ancestor of particle 16 pb + i" -- one gather per lane, no transposition; the ancestors of a wave's 64 slots come from a
//     64-entry table in LDS;
//   * A fragments ([T R], column-major, zero-padded to MP = 32 / 64 rows) are read straight from the filter's table (L1 / L2);
//   * the noise goes through a 2 KB tile per wave four rows at a time: four normals are exactly one PACKED Philox block
//     (pf_device.h), drawn by the lane that owns the particle;
//   * k-chunks of four are skipped beyond the real m and k, so padding costs rows, not columns;
//   * the D fragments ARE the new component-major state: stored as they are (128-byte segments per component and 16 particles);
//     the signals Z_j x' are partial sums over a lane's 4 MB components and two cross-lane adds.
// Everything that involves more than one particle is pf_canon.h, so the two shapes below give the same bits for a filter:
//   pf_big_wg        one workgroup per filter, particle system in LDS, one wave per row of 64 particles (N <= 512 at MP = 32,
//                    <= 256 at MP = 64), the whole time loop in one launch, 3 workgroup barriers per step;
//   pf_big_row_step  one wave per row, particle system in HBM, one launch per time step, the exchange of pf_row_step.h (N <= 16 384 with the row
//                    prefixes in registers; up to 102 400 -- the reference's front-end stops at 1e5 -- with the prefixes in LDS: round 6).
// fp64 MFMA and fp64 VALU have the same peak on MI355X (78.6 TFLOP/s): the matrix cores buy the operand delivery (no m^2 scalar
// operands per particle) and the registers (64 accumulators per lane at MP = 64 instead of three arrays of 64), not flops.
#pragma once

namespace bssm {

typedef double big_d4 __attribute__((ext_vector_type(4)));
constexpr int BIG_NT_MAX = 512;

// new states of the wave's 64 particles in D-fragment layout: v[rb][pb][q] = x'[16 rb + kq + 4 q][particle 16 pb + i]
template <int MB> struct BigAcc { big_d4 v[MB][4]; };

__device__ __forceinline__ double big_xor_add(double x) {  // sum over the four lanes that share i = lane & 15 (fixed order)
  x += __shfl_xor(x, 16, 64);
  x += __shfl_xor(x, 32, 64);
  return x;
}

// acc += A[:, 4 kk ... 4 kk + 3] . B chunk, every row block; Acol = first of the four columns (column-major, leading dimension MP)
// (LD: leading dimension of the tables = the padded state dimension of the batch; LD < 16 MB -- an eight-dimensional state in one
// 16-row block, the C4 experiment -- leaves the rows beyond LD zero)
template <int MB, int LD>
__device__ __forceinline__ void big_afrag(const double* __restrict__ Acol, double (&afr)[MB], const int lane) {
  const int i = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int rb = 0; rb < MB; rb++)
    afr[rb] = (LD == 16 * MB || 16 * rb + i < LD) ? Acol[(LD == 16 * MB ? 16 * rb + i : min(16 * rb + i, LD - 1)) + LD * kq] : 0.0;
}
// acc += A[:, 4 kk ... 4 kk + 3] . B chunk, every row block
template <int MB>
__device__ __forceinline__ void big_chunk(BigAcc<MB>& acc, const double (&afr)[MB], const double (&bf)[4]) {
#pragma unroll
  for (int rb = 0; rb < MB; rb++)
#pragma unroll
    for (int pb = 0; pb < 4; pb++) acc.v[rb][pb] = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[rb], bf[pb], acc.v[rb][pb], 0, 0, 0);
}

// X' = c0 + A1 . S(ancestors) + A2 . noise for the 64 particles of this wave:
//   c0  [MP]                     the constant (C_t; a1 at t = 0)
//   A1  [MP x MP] column-major   T_t, applied to components [0, ms) of the ancestors' states (ms = 0: no state term)
//   A2  [MP x KP] column-major   R_t (L at t = 0), applied to `ncnt` normals per particle
//   state_at(c, pb)              component c of the ancestor of particle 16 pb + (lane & 15); rows >= ms are never asked for
//   noise4(kk, z)                normals 4 kk ... 4 kk + 3 of THIS lane's particle (lane = particle inside the wave's row)
//   tile                         4 x 64 doubles of LDS, this wave's
// The operands of k-chunk kk + 1 (A fragments from the table, B fragments from the particle system) are requested BEFORE the
// matrix instructions of chunk kk are issued: the loops have run-time bounds (real m, real k), so nothing else would overlap a
// chunk's load round trip with the previous chunk's 8 ... 16 MFMAs (round 5, workgroup shape: psi-APF m = 53 54 -> 28 ms per 1024 filters,
// m = 25 15.6 -> 11.0 ms; bootstrap m = 53 15.1 -> 12.8 ms; profiles/r05_big_states.txt).
// zexit (round 6, psi-APF): A2 is a factor U sqrt(s) with its columns in DESCENDING order of s (conditional_cov; the thin factors and the cut leave the
// columns behind the rank exact zeros): the first k-chunk of A2 whose fragments are all zero ends the noise loop -- the chunks behind it are zero as
// well, and adding their exact-zero products changes nothing.  bsm_ng(period = 52): 1 chunk of 14 (3 noise directions of 53), i.e. 1 instead of 14
// Philox blocks, 2 instead of 27 Box-Muller pairs and 15 instead of 28 chunks of matrix instructions per particle-step.  The stream FORMAT still
// follows the full count (`ncnt`), so the normals that are drawn are the oracle's.
template <int MB, int LD, class StateAt, class Noise4>
__device__ __forceinline__ void big_affine(BigAcc<MB>& acc, const double* __restrict__ c0, const double* __restrict__ A1, const int ms,
                                           const double* __restrict__ A2, const int ncnt, StateAt&& state_at, Noise4&& noise4,
                                           double* tile, const int lane, const bool zexit = false) {
  constexpr int MP = LD;
  const int i = lane & 15, kq = lane >> 4;
  auto c_at = [&](int r) { return (LD == 16 * MB || r < LD) ? c0[LD == 16 * MB ? r : min(r, LD - 1)] : 0.0; };
#pragma unroll
  for (int rb = 0; rb < MB; rb++) {
    const big_d4 cv{c_at(16 * rb + kq), c_at(16 * rb + kq + 4), c_at(16 * rb + kq + 8), c_at(16 * rb + kq + 12)};
#pragma unroll
    for (int pb = 0; pb < 4; pb++) acc.v[rb][pb] = cv;
  }
  const int nks = (ms + 3) >> 2, nkn = (ncnt + 3) >> 2;
  double afr[MB], bf[4], afn[MB], bfn[4];
  auto fetch_state = [&](int kk, double (&a)[MB], double (&bq)[4]) {
    big_afrag<MB, LD>(A1 + (size_t)MP * 4 * kk, a, lane);
    const int c = 4 * kk + kq;
#pragma unroll
    for (int pb = 0; pb < 4; pb++) bq[pb] = c < ms ? state_at(c, pb) : 0.0;
  };
  if (nks > 0) fetch_state(0, afr, bf);
  for (int kk = 0; kk < nks; kk++) {
    if (kk + 1 < nks) fetch_state(kk + 1, afn, bfn);
    else if (nkn > 0) big_afrag<MB, LD>(A2, afn, lane);  // (the first noise chunk's A fragments under the last state chunk)
    big_chunk<MB>(acc, afr, bf);
#pragma unroll
    for (int rb = 0; rb < MB; rb++) afr[rb] = afn[rb];
#pragma unroll
    for (int pb = 0; pb < 4; pb++) bf[pb] = bfn[pb];
  }
  if (nks == 0 && nkn > 0) big_afrag<MB, LD>(A2, afr, lane);
  for (int kk = 0; kk < nkn; kk++) {
    if (zexit) {
      bool nz = false;
#pragma unroll
      for (int rb = 0; rb < MB; rb++) nz = nz || afr[rb] != 0.0;
      if (__builtin_amdgcn_ballot_w64(nz) == 0ull) break;
    }
    double z[4];
    noise4(kk, z);
    if (kk + 1 < nkn) big_afrag<MB, LD>(A2 + (size_t)MP * 4 * (kk + 1), afn, lane);
    lds_wave_sync();  // the previous chunk's reads of the tile are done
#pragma unroll
    for (int c = 0; c < 4; c++) tile[c * 64 + lane] = 4 * kk + c < ncnt ? z[c] : 0.0;
    lds_wave_sync();
#pragma unroll
    for (int pb = 0; pb < 4; pb++) bf[pb] = tile[kq * 64 + 16 * pb + i];
    big_chunk<MB>(acc, afr, bf);
#pragma unroll
    for (int rb = 0; rb < MB; rb++) afr[rb] = afn[rb];
  }
}

// normals 4 kk ... 4 kk + 3 of draw (purpose, t, particle): the stream format follows the number of normals of the whole draw
// (pf_device.h: WIDE up to two, PACKED above -- four normals are one block)
__device__ __forceinline__ void big_normals4(const RngKey key, const uint32_t purpose, const uint32_t t, const uint32_t particle, const int cnt,
                                             const int kk, const u32x4& blk0, double (&z)[4]) {
  z[0] = z[1] = z[2] = z[3] = 0.0;
  if (cnt <= 2) { box_muller(blk0, z[0], z[1]); return; }
  const u32x4 w = kk == 0 ? blk0 : philox4x32_10(particle, t, (uint32_t)kk, purpose, key.seed, key.stream);
  box_muller_packed(w.x, w.y, z[0], z[1]);
  box_muller_packed(w.z, w.w, z[2], z[3]);
}

// log-weight of this lane's particle (lane = particle inside the row) from the D fragments: the signals D_j + Z_j x' of the row's
// p series (src/model_ssm_ung.cpp:447-493, :385-439; src/model_ssm_mng.cpp:357-442), then the densities as log_weight_of
template <int MB, int LD>
__device__ __forceinline__ double big_log_weight(const FilterBatch& b, const double* __restrict__ row, const BigAcc<MB>& acc, const int lane) {
  constexpr int MP = LD;
  if (row[RH_FLAG] == 0.0) return 0.0;  // nothing observed at t: weights are ones (src/model_ssm_ung.cpp:644-647)
  const int kq = lane >> 4;
  double w = 0.0;
  for (int j = 0; j < b.p; j++) {
    const double* __restrict__ ob = row + ROW_HDR + j * (SER_HDR + MP);
    if (ob[OB_FLAG] == 0.0) continue;
    double own = 0.0;
#pragma unroll
    for (int pb = 0; pb < 4; pb++) {
      double part = 0.0;
#pragma unroll
      for (int rb = 0; rb < MB; rb++)
#pragma unroll
        for (int q = 0; q < 4; q++) { if (LD == 16 * MB || 16 * rb + kq + 4 * q < LD) part = __builtin_fma(ob[SER_HDR + (LD == 16 * MB ? 16 * rb + kq + 4 * q : min(16 * rb + kq + 4 * q, LD - 1))], acc.v[rb][pb][q], part); }
      const double s = big_xor_add(part);
      own = kq == pb ? s : own;  // this lane's particle is 16 kq + i
    }
    const double s = ob[OB_D] + own;
    const int dist = (int)ob[OB_DIST];
    w += b.psi ? log_psi_weight(dist, ob, ob[OB_PHI], s) : log_obs_density(dist, ob, ob[OB_PHI], s);
  }
  return b.psi ? w - row[RH_SCALE] : w;
}

// the D fragments as the new component-major particle system: dst[c * stride + particle]
template <int MB, bool NT>
__device__ __forceinline__ void big_store(const BigAcc<MB>& acc, double* __restrict__ dst, const size_t stride, const int rows, const int first,
                                          const int count, const int lane) {
  const int i = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int rb = 0; rb < MB; rb++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int c = 16 * rb + kq + 4 * q;
      if (c >= rows) continue;
#pragma unroll
      for (int pb = 0; pb < 4; pb++) {
        const int prt = 16 * pb + i;
        if (prt < count) {
          if constexpr (NT) __builtin_nontemporal_store(acc.v[rb][pb][q], &dst[(size_t)c * stride + first + prt]);
          else dst[(size_t)c * stride + first + prt] = acc.v[rb][pb][q];
        }
      }
    }
}


// ------------------------------------------------------------------------------------------------------------------
// Round 6: STRUCTURED transitions.  The models the limit of 16 states was lifted for are bsm_ng / bsm_lg with a seasonal component
// (R/models.R:995-1060 builds their T): level' = level + slope, slope' = slope, gamma_1' = -(gamma_1 + ... + gamma_{s-1}),
// gamma_i' = gamma_{i-1} -- a companion matrix of +-1 with ~2 m non-zeros of 64 x 56 padded entries, and three or four noise terms.
// The dense matrix-core transition spends 2 MP (m + k) flops per particle on it and reaches 0.26-0.49 of the matrix peak; what the step
// really is: gather the ancestor's m components, write m components, a handful of additions.  The host recognises such a T (every entry
// +-1, at most two per column, at most eight rows that are not a plain copy: api_internal.h, sparse_descriptor) and the kernels below
// take this path for the bootstrap filter's steps (the initial draw a1 + L u stays on the matrix cores: L is dense):
//   rows with ONE entry and no noise ("direct"):  x'[r] = C[r] +- x[c]
//   the other rows ("slots", <= 8):               x'[r] = C[r] + sum of +-x[c] in ascending c + sum_k R[r][k] eta_k
//   signal Z x' accumulated over the rows with a non-zero Z in the order they are produced (direct rows by source column, then the slots)
// xval(c) hands over component c of this lane's ancestor; store(r, v) takes component r of the new state.  The column loop is unrolled
// (c is static: xval may be a register array; the row family's loads all issue up front), everything read from `sp` / the tables is
// wave-uniform.  Both shapes call it with the same operands: the same bits.
// (SP_SLOTS, SP_HDR, SP_SLOT_WORDS, SP_WORDS: pf_kernels.h -- the host builds the descriptor)
// Descriptor (ints, wave-uniform): [0] r_lo, [1] r_hi: the SHIFT BLOCK, rows r_lo ... r_hi with x'[r] = C[r] + x[r - 1] (the seasonal lags);
// [2], [3]: bit r set = Z is non-zero at row r of the block; then per slot s (a row that is not such a copy: level, slope, the first seasonal
// state, a noise state ...): [row | 0x100 valid | 0x200 Z non-zero | 0x400 has noise], the columns with +1 (two words), the columns with -1.
template <int MP, class XVal, class Store>
__device__ __forceinline__ double sparse_apply(const int* __restrict__ sp, const int cnz, const double* __restrict__ mt, const int KP, const int m,
                                               const int k, const double (&z)[8], const double* __restrict__ zrow /* Z of the row to weigh, or null */,
                                               XVal&& xval, Store&& store) {
  const double* __restrict__ Rm = mt + MP * MP;
  const double* __restrict__ Cv = mt + MP * MP + MP * KP;
  const int r_lo = __builtin_amdgcn_readfirstlane(sp[0]), r_hi = __builtin_amdgcn_readfirstlane(sp[1]);
  const unsigned long long zmask = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(sp[2]) | ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(sp[3]) << 32);
  double zsum = 0.0;
  // ---- the slots first into registers (they read the OLD components, and a slot's row may be a source of the shift block)
  double sv[SP_SLOTS];
  int srow[SP_SLOTS];
#pragma unroll
  for (int s = 0; s < SP_SLOTS; s++) {
    const int* __restrict__ w = sp + SP_HDR + SP_SLOT_WORDS * s;
    srow[s] = __builtin_amdgcn_readfirstlane(w[0]);
    sv[s] = 0.0;
    if (!(srow[s] & 0x100)) continue;
    const unsigned long long P = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(w[1]) | ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(w[2]) << 32);
    const unsigned long long Nn = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(w[3]) | ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(w[4]) << 32);
    const int row = srow[s] & 63;
    double acc = Cv[row];
#pragma unroll
    for (int c = 0; c < MP; c++) {
      const double v = xval(c);
      acc += ((P >> c) & 1ull) ? v : (((Nn >> c) & 1ull) ? -v : 0.0);   // (ascending c; an absent column adds an exact zero)
    }
    if (srow[s] & 0x400) {
#pragma unroll
      for (int kk = 0; kk < 8; kk++) if (kk < k) acc = __builtin_fma(Rm[row + MP * kk], z[kk], acc);
    }
    sv[s] = acc;
  }
  // ---- the shift block
#pragma unroll
  for (int r = 1; r < MP; r++) {
    if (r < r_lo || r > r_hi) continue;
    const double v = cnz ? Cv[r] + xval(r - 1) : xval(r - 1);
    store(r, v);
    if (zrow && ((zmask >> r) & 1ull)) zsum = __builtin_fma(zrow[r], v, zsum);
  }
#pragma unroll
  for (int s = 0; s < SP_SLOTS; s++) {
    if (!(srow[s] & 0x100)) continue;
    const int row = srow[s] & 63;
    store(row, sv[s]);
    if (zrow && (srow[s] & 0x200)) zsum = __builtin_fma(zrow[row], sv[s], zsum);
  }
  (void)m;
  return zsum;
}
#ifndef __HIPCC_RTC__
// BSSM_AMD_SPARSE: 0 never, all = wherever a descriptor exists (experiments), default = where it measured faster
inline int sparse_mode() { return opt().sparse; }
#endif
// bootstrap log-weight of one particle from its signal (one observation series): big_log_weight for p = 1, the signal given
__device__ __forceinline__ double sparse_log_weight(const double* __restrict__ row, const double zsum) {
  if (row[RH_FLAG] == 0.0) return 0.0;
  const double* __restrict__ ob = row + ROW_HDR;
  if (ob[OB_FLAG] == 0.0) return 0.0;
  return log_obs_density((int)ob[OB_DIST], ob, ob[OB_PHI], ob[OB_D] + zsum);
}
// the k <= 8 normals of a step draw (two PACKED blocks at most; the stream format of big_normals4)
__device__ __forceinline__ void sparse_normals(const RngKey key, const uint32_t t, const uint32_t particle, const int k, const u32x4& blk, double (&z)[8]) {
  double z4[4];
  big_normals4(key, RNG_TRANSITION, t, particle, k, 0, blk, z4);
#pragma unroll
  for (int q = 0; q < 4; q++) { z[q] = z4[q]; z[4 + q] = 0.0; }
  if (k > 4) {
    big_normals4(key, RNG_TRANSITION, t, particle, k, 1, blk, z4);
#pragma unroll
    for (int q = 0; q < 4; q++) z[4 + q] = z4[q];
  }
}

// ------------------------------------------------------------------------------------------------------------------
// one workgroup per filter, one wave per row of 64 particles, particle system in LDS
// ------------------------------------------------------------------------------------------------------------------
// (MB = 4 runs 256 threads = one wave per SIMD; budgeted for 512 anyway: with the 470 registers a 256-thread bound allows it measured
// 16.8 ms against 12.8 ms per 1024 bootstrap filters of 256 particles at 256 registers and 67 spilt, profiles/r05_big_states.txt)
template <int MB, bool SPARSE = false>
__global__ void __launch_bounds__(BIG_NT_MAX, 1) pf_big_wg(const FilterBatch b) {
  constexpr int MP = 16 * MB;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  math_tab_init();
  const int NT = (int)blockDim.x, tid = threadIdx.x, lane = tid & 63, nw = NT >> 6, i16 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Ncap = NT, N = b.N, n = b.n, m = b.m, KP = b.KP, f = blockIdx.x;
  double* __restrict__ s_state = smem;                          // [MP][Ncap]
  double* __restrict__ s_cdf = smem + (size_t)MP * Ncap;        // [Ncap] row-local scans (pf_canon.h)
  double* __restrict__ s_max = s_cdf + Ncap;                    // [8]
  double* __restrict__ s_tot = s_max + 8;                       // [8]
  double* __restrict__ s_tile = s_tot + 8 + wave * 256;         // [nw][4 x 64]
  int* __restrict__ s_anc = (int*)(s_tot + 8 + nw * 256);       // [Ncap]
  const RngKey key{b.seed, b.stream0 + (uint32_t)f};
  const double* __restrict__ mats = b.mats + (size_t)f * b.mats_fstride;
  const double* __restrict__ obs = b.obs + (size_t)f * b.obs_fstride;
  const double* __restrict__ ini = b.init + (size_t)f * b.init_fstride;
  const int obs_stride = ROW_HDR + b.p * (SER_HDR + MP);
  const int j = tid;                 // this lane's particle
  const int row0 = wave * CANON_ROW;  // first particle of this wave's row
  const int cnt_row = min(CANON_ROW, N - row0);  // (<= 0: a wave without particles cannot occur: NT = 64 ceil(N / 64))
  const int ncnt_step = b.psi ? m : b.k;  // normals of a step draw (psi-APF: the m-dimensional proposal noise, src/model_ssm_ung.cpp:585-586)
  const int B = (N + CANON_ROW - 1) / CANON_ROW;
  double loglik = 0.0, w;
  BigAcc<MB> acc;

  // ---- initial particles a1 + L u (src/model_ssm_ung.cpp:626; psi: alphahat_0 + V_0 u, :543) and their weights
  {
    const u32x4 blk0 = philox4x32_10((uint32_t)j, 0u, 0u, RNG_INIT, key.seed, key.stream);
    big_affine<MB, MP>(acc, ini, ini, 0, ini + MP, m, [](int, int) { return 0.0; },
                   [&](int kk, double (&z)[4]) { big_normals4(key, RNG_INIT, 0u, (uint32_t)j, m, kk, blk0, z); }, s_tile, lane, b.psi != 0);
    big_store<MB, false>(acc, s_state, (size_t)Ncap, MP, row0, CANON_ROW, lane);
    if (b.alpha_hist) big_store<MB, true>(acc, b.alpha_hist + (size_t)f * (n + 1) * m * N, (size_t)N, m, row0, cnt_row, lane);
    w = n > 0 ? big_log_weight<MB, MP>(b, obs, acc, lane) : 0.0;
    if (j >= N) w = -INFINITY;
  }

  for (int t = 0; t <= n; t++) {
    // ---- normalise the time-t weights, canonical row form (pf_canon.h; src/model_ssm_ung.cpp:675-679)
    const CanonRow rs = canon_row_stats(w);
    if (j < N) s_cdf[j] = rs.sc;
    if (lane == 63) s_tot[wave] = rs.sc;
    if (lane == 0) s_max[wave] = rs.M;
    lds_barrier();  // (1)
    const double pmx[1] = {lane < B ? s_max[lane] : -INFINITY}, psm[1] = {lane < B ? s_tot[lane] : 0.0};
    const CanonComb<1> cc = canon_combine<1>(pmx, psm, B);
    if (b.weights && j < N) b.weights[((size_t)f * (n + 1) + t) * N + j] = rs.e * canon_row_scale(cc, wave);
    if (!(cc.S > 0.0)) {  // :678-682 (uniform across the block)
      loglik = -INFINITY;
      if (tid == 0) b.dead_at[f] = t + 1;
      break;
    }
    const double c_scale = canon_stratum(cc.S, N);
    if (wave == 0) loglik += canon_llinc(cc.M, c_scale);  // :683
    if (t == n) break;

    // ---- stratified resampling (stratified_sample.cpp:15-26): source row, bisection inside it
    const u32x4 blk = rng_step_block(key, (uint32_t)t, (uint32_t)j);
    const uint32_t uword = rng_resample_word_rt(ncnt_step, key, (uint32_t)t, (uint32_t)j, blk);
    const CanonSource src = canon_source_row(canon_target(uword, j, c_scale), j < N, cc.px, lane);
    int anc = 0;
    if (j < N) {
      const int slen = min(CANON_ROW, N - src.e * CANON_ROW);
      anc = src.e * CANON_ROW + min(canon_count_below(s_cdf + src.e * CANON_ROW, slen, src.vp), slen - 1);
      if (b.indices) __builtin_nontemporal_store((uint32_t)anc, &b.indices[((size_t)f * n + t) * N + j]);
    }
    const double* __restrict__ mt = mats + (size_t)t * b.mats_tstride;
    if constexpr (SPARSE) {
      // ---- structured transition (sparse_apply): this lane's particle -- its ancestor's components into registers, the barrier, then the
      // new components straight into the particle system
      double x[MP], z[8];
#pragma unroll
      for (int c = 0; c < MP; c++) x[c] = c < m ? s_state[c * Ncap + anc] : 0.0;
      sparse_normals(key, (uint32_t)t, (uint32_t)j, ncnt_step, blk, z);
      lds_barrier();  // (2) every gather has read the old particle system (and every search the scans)
      double* __restrict__ hist_new = b.alpha_hist ? b.alpha_hist + ((size_t)f * (n + 1) + (t + 1)) * m * N : nullptr;
      const double* __restrict__ orow = obs + (size_t)(t + 1 < n ? t + 1 : 0) * obs_stride;
      const double zsum = sparse_apply<MP>(b.sp, b.sp_cnz, mt, KP, m, ncnt_step, z, t + 1 < n ? orow + ROW_HDR + SER_HDR : nullptr,
                                           [&](int c) { return x[c]; },
                                           [&](int r, double v) { s_state[r * Ncap + j] = v; if (hist_new && j < N) __builtin_nontemporal_store(v, &hist_new[(size_t)r * N + j]); });
      w = t + 1 < n ? sparse_log_weight(orow, zsum) : 0.0;
      if (j >= N) w = -INFINITY;
      continue;
    }
    s_anc[j] = anc;
    lds_barrier();  // (2) every search has read the scans; the ancestor table is visible

    // ---- transition on the matrix cores: B fragments gathered from the component-major particle system
    int ancs[4];
#pragma unroll
    for (int pb = 0; pb < 4; pb++) ancs[pb] = s_anc[row0 + 16 * pb + i16];
    big_affine<MB, MP>(acc, mt + MP * MP + MP * KP, mt, m, mt + MP * MP, ncnt_step,
                   [&](int c, int pb) { return s_state[c * Ncap + ancs[pb]]; },
                   [&](int kk, double (&z)[4]) { big_normals4(key, RNG_TRANSITION, (uint32_t)t, (uint32_t)j, ncnt_step, kk, blk, z); }, s_tile, lane, b.psi != 0);
    lds_barrier();  // (3) every gather has read the old particle system
    big_store<MB, false>(acc, s_state, (size_t)Ncap, MP, row0, CANON_ROW, lane);
    if (b.alpha_hist) big_store<MB, true>(acc, b.alpha_hist + ((size_t)f * (n + 1) + (t + 1)) * m * N, (size_t)N, m, row0, cnt_row, lane);
    w = t + 1 < n ? big_log_weight<MB, MP>(b, obs + (size_t)(t + 1) * obs_stride, acc, lane) : 0.0;
    if (j >= N) w = -INFINITY;
  }
  if (tid == 0 && b.loglik) b.loglik[f] = loglik;
}

// largest N of the workgroup-resident shape: one wave per row, at most BIG_NT_MAX threads, the particle system in 160 KB of LDS
inline int big_wg_capacity(int MP) {
  for (int N = BIG_NT_MAX; N >= 64; N -= 64) {
    const size_t lds = ((size_t)(MP + 1) * N + 16 + (size_t)(N / 64) * 256) * sizeof(double) + (size_t)N * sizeof(int);
    if (lds <= 160 * 1024) return N;
  }
  return 0;
}

template <int MB>
hipError_t launch_big_wg(const FilterBatch& b, hipStream_t stream) {
  constexpr int MP = 16 * MB;
  if (b.MP != MP || b.fpar || b.KP % 4 != 0 || b.KP < 4 || b.KP > MP || b.N > big_wg_capacity(MP)) return hipErrorInvalidValue;
  const int nt = (b.N + 63) / 64 * 64;
  const size_t lds = ((size_t)(MP + 1) * nt + 16 + (size_t)(nt / 64) * 256) * sizeof(double) + (size_t)nt * sizeof(int);
  // (measured, profiles/r06_big_states.txt: the structured transition does NOT pay in the workgroup-resident shape -- 256 particles x 1024 filters of
  // 25 / 53 states 7.9 / 15.1 ms against 5.9 / 12.7 ms on the matrix cores: one wave per SIMD holds 32 / 64 components per lane, 165-268 scalar
  // registers spilt; BSSM_AMD_SPARSE=all selects it for experiments)
  const bool sparse = b.sp != nullptr && b.psi == 0 && b.p == 1 && b.mats_tstride == 0 && b.k <= 8 && sparse_mode() == 2;
  { char nm[96]; std::snprintf(nm, sizeof nm, sparse ? "pf_big_wg<%d,sparse>" : "pf_big_wg<%d>", MB); note_filter_kernel(nm); }
  const void* kern = sparse ? (const void*)pf_big_wg<MB, true> : (const void*)pf_big_wg<MB, false>;
  hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  if ((e = hipMemsetAsync(b.dead_at, 0, sizeof(int) * b.F, stream)) != hipSuccess) return e;
  if (sparse) hipLaunchKernelGGL((pf_big_wg<MB, true>), dim3(b.F), dim3(nt), lds, stream, b);
  else hipLaunchKernelGGL((pf_big_wg<MB, false>), dim3(b.F), dim3(nt), lds, stream, b);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// one wave per row, particle system in HBM, one launch per time step (the exchange of pf_row_step.h)
// ------------------------------------------------------------------------------------------------------------------
// waves per SIMD the row shape is budgeted for: a step is a chain of dependent round trips (partials, source rows, one gather per
// k-chunk), hidden by other waves, not by a deeper pipeline inside one (MB = 2: 196 VGPRs unconstrained, 128 with two spilt)
constexpr int big_row_occupancy(int mb) { return mb <= 2 ? 4 : 2; }
template <int MB, int LD = 16 * MB>
__global__ void __launch_bounds__(ROW_NT) pf_big_row_init(const FilterBatch b, const GridWorkspace ws) {
  constexpr int MP = LD;
  __shared__ double s_tiles[4 * 256];
  math_tab_init();
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int f = ws.f0 + blockIdx.y, g = blockIdx.x * 4 + wave, N = b.N, n = b.n, m = b.m;
  if (g >= ws.B) return;
  const int j = g * CANON_ROW + lane, cnt_row = min(CANON_ROW, N - g * CANON_ROW);
  const RngKey key{b.seed, b.stream0 + (uint32_t)f};
  const double* __restrict__ obs = b.obs + (size_t)f * b.obs_fstride;
  const double* __restrict__ ini = b.init + (size_t)f * b.init_fstride;
  BigAcc<MB> acc;
  const u32x4 blk0 = philox4x32_10((uint32_t)j, 0u, 0u, RNG_INIT, key.seed, key.stream);
  big_affine<MB, LD>(acc, ini, ini, 0, ini + MP, m, [](int, int) { return 0.0; },
                 [&](int kk, double (&z)[4]) { big_normals4(key, RNG_INIT, 0u, (uint32_t)j, m, kk, blk0, z); }, s_tiles + wave * 256, lane, b.psi != 0);
  double* __restrict__ st_new = b.alpha_hist ? b.alpha_hist + (size_t)f * (n + 1) * m * N : ws.state[0] + (size_t)f * m * N;
  big_store<MB, false>(acc, st_new, (size_t)N, m, g * CANON_ROW, cnt_row, lane);
  const double w = n > 0 ? big_log_weight<MB, LD>(b, obs, acc, lane) : 0.0;
  row_publish(b, ws, f, g, j, j < N, 0, w, lane);
}

// Round 6: filters of more than 16 384 particles (the reference's front-end takes up to 1e5 for any model, R/check_arguments.R:407-418;
// src/model_ssm_ung.cpp:611 has no limit).  The canonical arithmetic (pf_canon.h) is defined for any number of rows; what ends at 256 rows is
// the REGISTER form of the row prefixes (ROW_NP slots per lane).  The WIDE form keeps them in LDS -- Q (inclusive scaled prefix), I (1 / scale)
// and the scale of every row, up to BIG_WIDE_ROWS rows = 102 400 particles -- built by wave 0 slot by slot exactly as canon_combine does (one
// wave scan per 64 rows on top of the running total), and every lane finds its source row by a bisection over Q instead of ballots.
constexpr int BIG_WIDE_ROWS = 1600;
// (M, S) of the time-t weights and the three per-row tables from the (M_row, S_row) partials; wave 0 calls it, every lane of it
__device__ __forceinline__ void big_wide_combine(const double* __restrict__ part, const int B, double* sQ, double* sI, double* sSC, double* sMS, const int lane) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  double mx = -INFINITY;
  for (int k0 = 0; k0 < B; k0 += 64) { const int g = k0 + lane; if (g < B) mx = fmax(mx, part[2 * g]); }
  const double M = wave_max_key(mx);
  double run = 0.0;
  for (int k0 = 0; k0 < B; k0 += 64) {
    const int g = k0 + lane;
    double pm = -INFINITY, ps = 0.0;
    if (g < B) { const d2 pp = *(const d2*)(part + 2 * g); pm = pp.x; ps = pp.y; }
    const double sc = fast_exp(pm - M);   // rows beyond B: 0; a poisoned (NaN) partial makes S NaN
    const double Qk = wave_scan_incl(canon_opaque(ps * sc));
    const double Q = k0 == 0 ? Qk : run + Qk;
    if (g < B) { sQ[g] = Q; sI[g] = 1.0 / sc; sSC[g] = sc; }
    run = lane_broadcast(Q, 63);
  }
  if (lane == 0) { sMS[0] = M; sMS[1] = run; }
}
// source row of target v: #{g : Q_g < v}, clamped; the target in that row's own scale (canon_source_row's result for one lane)
__device__ __forceinline__ void big_wide_source(const double v, const int B, const double* sQ, const double* sI, int& e, double& vp) {
  int lo = 0, len = B;   // first index with Q >= v
  while (len > 0) {
    const int half = len >> 1;
    if (sQ[lo + half] < v) { lo += half + 1; len -= half + 1; } else len = half;
  }
  e = min(lo, B - 1);
  const double Pm1 = e > 0 ? sQ[e - 1] : 0.0;
  vp = canon_opaque(v - Pm1) * sI[e];
}

template <int MB, int LD = 16 * MB, bool WIDE = false, bool SPARSE = false>
__global__ void __launch_bounds__(ROW_NT, big_row_occupancy(MB)) pf_big_row_step(const FilterBatch b, const GridWorkspace ws, const int t) {
  constexpr int MP = LD;
  __shared__ double s_win[4 * ROW_WIN];
  __shared__ double s_comb[WIDE ? 3 * BIG_WIDE_ROWS + 2 : 3 * ROW_NP * 64 + 2];  // the combine of wave 0 (pf_canon.h: canon_publish; WIDE: Q | I | scale | M, S)
  __shared__ double s_tiles[4 * 256];
  __shared__ int s_ancs[4 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), i16 = lane & 15;
  const int f = ws.f0 + blockIdx.y, g = blockIdx.x * 4 + wave, B = ws.B, N = b.N, n = b.n, m = b.m, KP = b.KP, tn = t + 1;
  const MathTabRegs mtab = math_tab_load_begin256();
  const RngKey key{b.seed, b.stream0 + (uint32_t)f};
  const int j = g * CANON_ROW + lane;
  const bool has_row = g < B, act = has_row && j < N;
  const int ncnt_step = b.psi ? m : b.k;
  const double* __restrict__ lcdf = ws.lcdf + ((size_t)(t & 1) * b.F + f) * N;
  double* const wQ = s_comb, * const wI = s_comb + (WIDE ? BIG_WIDE_ROWS : 0), * const wSC = s_comb + (WIDE ? 2 * BIG_WIDE_ROWS : 0), * const wMS = s_comb + (WIDE ? 3 * BIG_WIDE_ROWS : 0);
  if (wave == 0) {
    if constexpr (WIDE) big_wide_combine(ws.partials + ((size_t)(t & 1) * b.F + f) * B * 2, B, wQ, wI, wSC, wMS, lane);
    else row_combine_publish(ws.partials + ((size_t)(t & 1) * b.F + f) * B * 2, B, s_comb, lane);
  }
  const u32x4 blk = rng_step_block(key, (uint32_t)t, (uint32_t)j);
  const uint32_t uword = rng_resample_word_rt(ncnt_step, key, (uint32_t)t, (uint32_t)j, blk);
  math_tab_load_end256(mtab);
  lds_barrier();  // (1) the only workgroup barrier of the step
  CanonPrefix<ROW_NP> px;
  double M, S, wscale_g;
  if constexpr (WIDE) { M = wMS[0]; S = wMS[1]; wscale_g = has_row ? wSC[g] : 0.0; px.B = B; }
  else { px = canon_prefix_load<ROW_NP>(s_comb, lane, B); M = canon_published_M<ROW_NP>(s_comb); S = canon_published_S<ROW_NP>(s_comb); wscale_g = canon_published_scale<ROW_NP>(s_comb, g); }
  if (b.weights && act) {  // the stored weights of time t are exp(w - M_row): to the scale of the whole time point (also on a degenerate step, :676-682)
    double* __restrict__ wt = b.weights + ((size_t)f * (n + 1) + t) * N;
    wt[j] = wt[j] * wscale_g;
  }
  if (!(S > 0.0)) {
    if (has_row && lane == 0) {
      double* __restrict__ pn = ws.partials + (((size_t)(tn & 1) * b.F + f) * B + g) * 2;
      pn[0] = __builtin_nan("");
      pn[1] = __builtin_nan("");
      if (g == 0) atomicCAS(&b.dead_at[f], 0, t + 1);
    }
    return;
  }
  if (!has_row) return;
  const double c_scale = canon_stratum(S, N);
  if (g == B - 1 && lane == 63) ws.llinc[(size_t)f * (n + 1) + t] = canon_llinc(M, c_scale);

  CanonSource src;
  if constexpr (WIDE) {
    // the 64 slots of a wave are consecutive, so their targets (and source rows) increase along the lanes: the first and the last active
    // lane bound the rows the wave draws from
    const double v = canon_target(uword, j, c_scale);
    src = CanonSource{0, 0.0, 0, -1};
    const unsigned long long amask = __builtin_amdgcn_ballot_w64(act);
    if (amask != 0ull) {
      big_wide_source(v, B, wQ, wI, src.e, src.vp);
      src.c_lo = __builtin_amdgcn_readlane(src.e, __builtin_ctzll(amask));
      src.c_hi = __builtin_amdgcn_readlane(src.e, 63 - __builtin_clzll(amask));
    }
  } else {
    src = canon_source_row(canon_target(uword, j, c_scale), act, px, lane);
  }
  double* sw = s_win + wave * ROW_WIN;
  int anc = 0;
  for (int r0 = src.c_lo; r0 <= src.c_hi; r0 += 4) {  // wave-uniform bounds
    const int nr = min(4, src.c_hi - r0 + 1);
    double xr[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int k = (r0 + q) * CANON_ROW + lane;
      xr[q] = (q < nr && k < N) ? lcdf[k] : 0.0;
    }
    lds_wave_sync();
#pragma unroll
    for (int q = 0; q < 4; q++) sw[q * CANON_ROW + lane] = xr[q];
    lds_wave_sync();
    if (act && src.e >= r0 && src.e < r0 + 4) {
      const int slen = min(CANON_ROW, N - src.e * CANON_ROW);
      anc = src.e * CANON_ROW + min(canon_count_below(sw + (src.e - r0) * CANON_ROW, slen, src.vp), slen - 1);
    }
  }
  if (b.indices && act) __builtin_nontemporal_store((uint32_t)anc, &b.indices[((size_t)f * n + t) * N + j]);
  const double* __restrict__ st_old = b.alpha_hist ? b.alpha_hist + ((size_t)f * (n + 1) + t) * m * N : ws.state[t & 1] + (size_t)f * m * N;
  double* __restrict__ st_new = b.alpha_hist ? b.alpha_hist + ((size_t)f * (n + 1) + tn) * m * N : ws.state[tn & 1] + (size_t)f * m * N;
  const double* __restrict__ mt = b.mats + (size_t)f * b.mats_fstride + (size_t)t * b.mats_tstride;
  const double* __restrict__ obs = b.obs + (size_t)f * b.obs_fstride;
  const int obs_stride = ROW_HDR + b.p * (SER_HDR + MP);
  if constexpr (SPARSE) {
    // ---- structured transition (sparse_apply): one particle per lane, its ancestor's components straight from the old particle system
    double z[8];
    sparse_normals(key, (uint32_t)t, (uint32_t)j, ncnt_step, blk, z);
    const double* __restrict__ orow = obs + (size_t)(tn < n ? tn : 0) * obs_stride;
    double x[MP];   // the ancestor's components: every load in flight before the first use
#pragma unroll
    for (int c = 0; c < MP; c++) x[c] = c < m ? st_old[(size_t)c * N + anc] : 0.0;
    const double zsum = sparse_apply<MP>(b.sp, b.sp_cnz, mt, KP, m, ncnt_step, z, tn < n ? orow + ROW_HDR + SER_HDR : nullptr,
                                         [&](int c) { return x[c]; },
                                         [&](int r, double v) { if (act) st_new[(size_t)r * N + j] = v; });
    const double ws_ = tn < n ? sparse_log_weight(orow, zsum) : 0.0;
    row_publish(b, ws, f, g, j, act, tn, ws_, lane);
    return;
  }
  int* sa = s_ancs + wave * 64;
  sa[lane] = anc;
  lds_wave_sync();
  int ancs[4];
#pragma unroll
  for (int pb = 0; pb < 4; pb++) ancs[pb] = sa[16 * pb + i16];

  BigAcc<MB> acc;
  big_affine<MB, LD>(acc, mt + MP * MP + MP * KP, mt, m, mt + MP * MP, ncnt_step,
                 [&](int c, int pb) { return st_old[(size_t)c * N + ancs[pb]]; },
                 [&](int kk, double (&z)[4]) { big_normals4(key, RNG_TRANSITION, (uint32_t)t, (uint32_t)j, ncnt_step, kk, blk, z); },
                 s_tiles + wave * 256, lane, b.psi != 0);
  big_store<MB, false>(acc, st_new, (size_t)N, m, g * CANON_ROW, min(CANON_ROW, N - g * CANON_ROW), lane);
  const double w = tn < n ? big_log_weight<MB, LD>(b, obs + (size_t)tn * obs_stride, acc, lane) : 0.0;
  row_publish(b, ws, f, g, j, act, tn, w, lane);
}

template <int MB, int LD = 16 * MB>
hipError_t launch_big_row(const FilterBatch& b, const GridWorkspace& ws, hipStream_t stream, StepTiming* timing) {
  constexpr int MP = LD;
  if (b.MP != MP || b.fpar || b.KP % 4 != 0 || b.KP < 4 || b.KP > MP) return hipErrorInvalidValue;
  if constexpr (LD == 16 * MB) {
    // (measured: the row shape gains 10-12 % at 33 ... 64 states -- 53 states 10.6-11.1 -> 9.6-9.9 ms per 262 144 particles x 200 steps = 0.59 of the
    // HBM roof on 16 m + 16 bytes -- and loses 5 % at 17 ... 32, where the dense tiles are small: on for MB = 4 only; BSSM_AMD_SPARSE=all / 0)
    const bool sparse = b.sp != nullptr && b.psi == 0 && b.p == 1 && b.mats_tstride == 0 && b.k <= 8 && (sparse_mode() == 2 || (sparse_mode() == 1 && MB == 4));
    if (ws.B > ROW_MAX_ROWS) {   // more than 16 384 particles: the row prefixes in LDS (round 6)
      if (ws.B > BIG_WIDE_ROWS) return hipErrorInvalidValue;
      { char nm[96]; std::snprintf(nm, sizeof nm, sparse ? "pf_big_row_step<%d,wide,sparse>" : "pf_big_row_step<%d,wide>", MB); note_filter_kernel(nm); }
      if (sparse) return launch_row_loop(b, ws, stream, timing, pf_big_row_step<MB, LD, true, true>, pf_big_row_init<MB, LD>, BIG_WIDE_ROWS);
      return launch_row_loop(b, ws, stream, timing, pf_big_row_step<MB, LD, true>, pf_big_row_init<MB, LD>, BIG_WIDE_ROWS);
    }
    if (sparse) {
      { char nm[96]; std::snprintf(nm, sizeof nm, "pf_big_row_step<%d,sparse>", MB); note_filter_kernel(nm); }
      return launch_row_loop(b, ws, stream, timing, pf_big_row_step<MB, LD, false, true>, pf_big_row_init<MB, LD>);
    }
  }
  { char nm[96]; if (LD == 16 * MB) std::snprintf(nm, sizeof nm, "pf_big_row_step<%d>", MB); else std::snprintf(nm, sizeof nm, "pf_big_row_step<%d,%d>", MB, LD); note_filter_kernel(nm); }
  return launch_row_loop(b, ws, stream, timing, pf_big_row_step<MB, LD>, pf_big_row_init<MB, LD>);
}

}  // namespace bssm
