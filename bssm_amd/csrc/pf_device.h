// pf_device.h -- device-side building blocks of the particle-filter kernels (gfx950, wave64).
//
// Counter-based random stream, wavefront reductions/scans, observation
// log-densities.  The reference draws from a sequential sitmo engine
// (src/model_ssm_ung.h:67; call sites src/model_ssm_ung.cpp:624,652,666); a
// sequential stream cannot be consumed by 65 536 lanes, so every draw here is
// a pure function of (seed, filter stream, purpose, time, particle, draw#).
#pragma once
// (__HIPCC_RTC__: the in-process compile of a user model, api_filters.cpp bssm_user_model_jit -- hiprtc supplies the HIP runtime
// declarations and the fixed-width integer types itself and has no host headers)
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

#include "pf_math.h"

namespace bssm {

// ---------------------------------------------------------------- layout of the per-filter tables
// obs row (one per time point t in [0, n)): ROW_HDR doubles, then p series blocks of SER_HDR + MP doubles
//   row:    RH_FLAG (any series observed), RH_SCALE (psi: scales_t)
//   series: y, u, D_j + xbeta, psi pseudo-observation, 1 / psi sd, host-precomputed quotient, observed flag,
//           distribution code, phi_j, pad; then Z_t[j, 0..MP)
enum { RH_FLAG = 0, RH_SCALE = 1, ROW_HDR = 2 };
enum { OB_Y = 0, OB_U = 1, OB_D = 2, OB_PSIY = 3, OB_PSIIH = 4, OB_AUX = 5, OB_FLAG = 6, OB_DIST = 7, OB_PHI = 8, SER_HDR = 10 };
// distribution codes = R/bootstrap_filter.R:106-109 ; 5 = gaussian (ssm_ulg path)
enum { DIST_SVM = 0, DIST_POISSON = 1, DIST_BINOMIAL = 2, DIST_NEGBIN = 3, DIST_GAMMA = 4, DIST_GAUSSIAN = 5 };
// RNG purposes (1 is retired: the resampling uniform rides in the step block, see rng_step_block)
enum { RNG_INIT = 0, RNG_RESAMPLE = 1, RNG_TRANSITION = 2, RNG_PICK = 3, RNG_SDE = 4 };

// Read-only model tables (T_t, R_t, C_t, observation rows, initial moments) are wave-uniform: loaded with s_load they cost
// no vector-memory instruction and no VGPR.  The compiler only scalarises a load it can prove unclobbered, and every
// load after the kernel's first global store fails that test, so such pointers are re-typed to the constant address
// space (same addresses; tells the compiler the memory is not written while the kernel runs -- true for these tables).
// Second obstacle: gfx950 has no scalar 64-bit multiply, so `base + (size_t)f * stride` of wave-uniform operands is
// evaluated on the VALU and the (uniform) address stays in VGPRs -- again vector loads.  v_readfirstlane on the two
// halves of the pointer moves it to SGPRs.
typedef __attribute__((address_space(4))) const double cdbl;
__device__ __forceinline__ const cdbl* as_uniform_table(const double* p) {
  const uint64_t a = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return (const cdbl*)(((uint64_t)hi << 32) | lo);
}

// Per-lane element of an array whose base is wave-uniform, with a 32-bit byte offset: selects the
// `global_load/store v, v_offset, s[base:base+1]` addressing form (no 64-bit address arithmetic on the VALU).
template <class T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
  const uint64_t a = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return (T*)(((uint64_t)hi << 32) | lo);
}
// (the pointer is re-typed to the global address space: after the integer round trip of uniform_ptr the compiler would
// otherwise fall back to flat_load / flat_store)
template <class T>
__device__ __forceinline__ T ld32(const T* uniform_base, uint32_t index) {
  typedef __attribute__((address_space(1))) const T gT;
  typedef __attribute__((address_space(1))) const char gC;
  return *(gT*)((gC*)(uint64_t)uniform_base + (uint32_t)(index * (uint32_t)sizeof(T)));
}
template <class T>
__device__ __forceinline__ void st32(T* uniform_base, uint32_t index, T value) {
  typedef __attribute__((address_space(1))) T gT;
  typedef __attribute__((address_space(1))) char gC;
  *(gT*)((gC*)(uint64_t)uniform_base + (uint32_t)(index * (uint32_t)sizeof(T))) = value;
}

template <class T>
__device__ __forceinline__ T ld32_nt(const T* uniform_base, uint32_t index) {
  typedef __attribute__((address_space(1))) const T gT;
  typedef __attribute__((address_space(1))) const char gC;
  return __builtin_nontemporal_load((gT*)((gC*)(uint64_t)uniform_base + (uint32_t)(index * (uint32_t)sizeof(T))));
}
// the same as a non-temporal store (`nt`: streamed, not kept in the caches): outputs nobody on the device reads again
template <class T>
__device__ __forceinline__ void st32_nt(T* uniform_base, uint32_t index, T value) {
  typedef __attribute__((address_space(1))) T gT;
  typedef __attribute__((address_space(1))) char gC;
  __builtin_nontemporal_store(value, (gT*)((gC*)(uint64_t)uniform_base + (uint32_t)(index * (uint32_t)sizeof(T))));
}

// Hides a uniform table pointer from common-subexpression elimination: loads through the result are re-issued (cheap
// s_loads from the scalar cache) instead of being kept live in dozens of SGPRs across the code in between -- the
// m = k = 4 transition needs 72 SGPRs for T, R, C, and holding them across the four particles of a lane spilt 55 SGPRs
// to VGPR lanes (a v_readlane per use).
__device__ __forceinline__ const cdbl* reload_table(const cdbl* p) {
  __asm__ volatile("" : "+s"(p));
  return p;
}

// ---------------------------------------------------------------- Philox4x32-10
struct u32x4 { uint32_t x, y, z, w; };

#ifndef BSSM_PHILOX_ROUNDS
#define BSSM_PHILOX_ROUNDS 10
#endif
__host__ __device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                         uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < BSSM_PHILOX_ROUNDS; r++) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return u32x4{c0, c1, c2, c3};
}

__host__ __device__ __forceinline__ uint64_t bits53(uint32_t a, uint32_t b) {
  return ((uint64_t)(a >> 5) << 26) | (uint64_t)(b >> 6);
}

struct RngKey { uint32_t seed, stream; };

// U[0,1) with 53 random bits (trajectory picks; not on the per-step path)
__device__ __forceinline__ double rng_uniform(RngKey k, uint32_t purpose, uint32_t t, uint32_t i, uint32_t d) {
  const u32x4 w = philox4x32_10(i, t, d, purpose, k.seed, k.stream);
  return u53_to_double(w.x >> 5, w.y >> 6) * 0x1.0p-53;
}

// Two block formats, chosen by the number of normals a draw needs (the oracle's rng_normals / orc_rng_uniform follow
// the same rule, oracle/bssm_oracle.c):
//   WIDE (a draw of <= 2 normals, and every pair drawn on its own -- the Milstein sub-steps): one block per pair,
//     words 0,1 -> 53-bit radius uniform in (0,1];  word 2 -> 32-bit angle;  word 3 -> spare 32-bit uniform.
//     The block (particle i, step t, draw 0, RNG_TRANSITION) is the "step block": its spare word is the stratified
//     resampling uniform of output slot i at step t, its pair the transition normals of the same slot, so a particle-step
//     of a model with <= 2 noise terms costs a single Philox call.
//   PACKED (a draw of > 2 normals by a linear-state filter -- ssm_ung / ssm_mng / ssm_ulg families: bootstrap, psi-APF,
//     simulation smoother; the functor-model filters keep WIDE blocks whatever the count; round 4): two pairs per block -- pair p comes from block (i, t, p >> 1, purpose), words
//     (0,1) for even p and (2,3) for odd p: 40-bit radius uniform in (0,1] (first word, then the top 8 bits of the second;
//     |z| <= 7.45) and 24-bit angle (the low 24 bits of the second word).  No word is left over, so the resampling uniforms
//     of such a filter come from blocks of their own, shared by FOUR consecutive slots: slot i at step t takes word i & 3
//     of block (i >> 2, t, 0, RNG_RESAMPLE).  A particle-step with four noise terms costs 1.25 Philox calls instead of 2,
//     one with eight 2.25 instead of 4 (a lane that holds four consecutive slots for the search draws their uniforms at once).
__device__ __forceinline__ u32x4 rng_step_block(RngKey k, uint32_t t, uint32_t i) {
  return philox4x32_10(i, t, 0u, RNG_TRANSITION, k.seed, k.stream);
}
__device__ __forceinline__ double block_uniform(const u32x4& w) { return (double)w.w * 0x1.0p-32; }  // [0,1)
// the four resampling uniforms of slots 4 g ... 4 g + 3 at step t (filters whose step draw is PACKED)
__device__ __forceinline__ u32x4 rng_resample_block(RngKey k, uint32_t t, uint32_t g) {
  return philox4x32_10(g, t, 0u, RNG_RESAMPLE, k.seed, k.stream);
}
__device__ __forceinline__ uint32_t u32x4_word(const u32x4& w, uint32_t j) {
  const uint32_t lo = (j & 1u) ? w.y : w.x, hi = (j & 1u) ? w.w : w.z;
  return (j & 2u) ? hi : lo;
}
// resampling uniform word of slot i at step t for a filter whose step draw has CNT normals; blk0 = the slot's step block
template <int CNT>
__device__ __forceinline__ uint32_t rng_resample_word(RngKey k, uint32_t t, uint32_t i, const u32x4& blk0) {
  if constexpr (CNT <= 2) return blk0.w;
  else return u32x4_word(rng_resample_block(k, t, i >> 2), i & 3u);
}
__device__ __forceinline__ uint32_t rng_resample_word_rt(int cnt, RngKey k, uint32_t t, uint32_t i, const u32x4& blk0) {
  return cnt <= 2 ? blk0.w : u32x4_word(rng_resample_block(k, t, i >> 2), i & 3u);
}

// sqrt for the Box-Muller radius argument x = -2 log(u) in [0, 74]: v_rsq_f64 seed, one coupled
// Newton step and two residual corrections; no range scaling needed on this interval.  <= 1 ulp.
__device__ __forceinline__ double sqrt_small(double x) {
#if BSSM_FAST_MATH
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
  g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
  return fmax(g, 0.0);  // x = 0 (u = 1): rsq gives inf, g is NaN, and v_max_f64 returns the other operand: 0
#else
  return sqrt(x);
#endif
}

// Box-Muller pair from one WIDE block
__device__ __forceinline__ void box_muller(const u32x4& w, double& z0, double& z1) {
  // (0,1]: (hi27 * 2^26 + lo26 + 1) * 2^-53, every step exact (two conversions and two fused multiply-adds)
  const double u1 = __builtin_fma((double)(w.x >> 5), 0x1.0p-27, __builtin_fma((double)(w.y >> 6), 0x1.0p-53, 0x1.0p-53));
  const double rad = sqrt_small(-2.0 * tl_log<false>(u1));
  double sn, cs;
  tl_sincos_2pi(w.z, sn, cs);  // angle 2*pi*w2*2^-32
  z0 = rad * cs;
  z1 = rad * sn;
}
// Box-Muller pair from one half (two words) of a PACKED block
__device__ __forceinline__ void box_muller_packed(const uint32_t a, const uint32_t b, double& z0, double& z1) {
  // (0,1]: ((a << 8 | b >> 24) + 1) * 2^-40, exact
  const double u1 = __builtin_fma((double)a, 0x1.0p-32, __builtin_fma((double)(b >> 24), 0x1.0p-40, 0x1.0p-40));
  const double rad = sqrt_small(-2.0 * tl_log<false>(u1));
  double sn, cs;
  tl_sincos_2pi(b << 8, sn, cs);  // angle 2*pi*(b & 0xffffff)*2^-24
  z0 = rad * cs;
  z1 = rad * sn;
}
__device__ __forceinline__ void rng_normal_pair(RngKey k, uint32_t purpose, uint32_t t, uint32_t i, uint32_t d,
                                                double& z0, double& z1) {
  box_muller(philox4x32_10(i, t, d, purpose, k.seed, k.stream), z0, z1);
}

// CNT normals for (purpose, t, i); blk0 is block (i, t, 0, purpose), already drawn.  CNT <= 2: the WIDE pair of blk0;
// CNT > 2: PACKED, pair p from half p & 1 of block p >> 1
// PACK = false: one WIDE block per pair whatever the count (the functor-model filters: ssm_nlg, ssm_sde, EKF particle filter)
template <int CNT, bool PACK = true>
__device__ __forceinline__ void rng_normals_blk(const u32x4& blk0, RngKey k, uint32_t purpose, uint32_t t, uint32_t i,
                                                double (&out)[CNT]) {
  if constexpr (CNT <= 2 || !PACK) {
#pragma unroll
    for (int j = 0; j < CNT; j += 2) {
      double z0, z1;
      if (j == 0) box_muller(blk0, z0, z1);
      else rng_normal_pair(k, purpose, t, i, (uint32_t)(j / 2), z0, z1);
      out[j] = z0;
      if (j + 1 < CNT) out[j + 1] = z1;
    }
  } else {
#pragma unroll
    for (int j = 0; j < CNT; j += 4) {
      const u32x4 w = j == 0 ? blk0 : philox4x32_10(i, t, (uint32_t)(j / 4), purpose, k.seed, k.stream);
      double z0, z1;
      box_muller_packed(w.x, w.y, z0, z1);
      out[j] = z0;
      if (j + 1 < CNT) out[j + 1] = z1;
      if (j + 2 < CNT) {
        box_muller_packed(w.z, w.w, z0, z1);
        out[j + 2] = z0;
        if (j + 3 < CNT) out[j + 3] = z1;
      }
    }
  }
}
template <int CNT, bool PACK = true>
__device__ __forceinline__ void rng_normals(RngKey k, uint32_t purpose, uint32_t t, uint32_t i, double (&out)[CNT]) {
  rng_normals_blk<CNT, PACK>(philox4x32_10(i, t, 0u, purpose, k.seed, k.stream), k, purpose, t, i, out);
}
// the same rule for a count known only at run time (out has room for cnt rounded up to even)
__device__ __forceinline__ void rng_normals_rt(RngKey k, uint32_t purpose, uint32_t t, uint32_t i, int cnt, double* out) {
  if (cnt <= 2) { rng_normal_pair(k, purpose, t, i, 0u, out[0], out[1]); return; }
  for (int j = 0; j < cnt; j += 4) {
    const u32x4 w = philox4x32_10(i, t, (uint32_t)(j / 4), purpose, k.seed, k.stream);
    box_muller_packed(w.x, w.y, out[j], out[j + 1]);
    if (j + 2 < cnt) box_muller_packed(w.z, w.w, out[j + 2], out[j + 3]);
  }
}

// ---------------------------------------------------------------- wavefront primitives (64 lanes)
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// Cross-lane moves on the DPP data path (no LDS crossbar, no address VGPR): a double travels as two
// 32-bit halves.  `old` supplies the value for lanes without a valid source (bound_ctrl = 0) and for
// rows masked out by row_mask, so an identity element there makes every step branch-free.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double identity, double v) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(identity), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(identity), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
constexpr int DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;

__device__ __forceinline__ double lane_broadcast(double v, int src_lane) {  // src_lane wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

// row shift with zero fill (bound_ctrl:1): lanes whose source lies outside the row read 0, so -- unlike dpp_move -- no
// identity has to be moved into the destination first (2 instructions per double instead of 4)
template <int CTRL>
__device__ __forceinline__ double dpp_shr_zero(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// inclusive prefix sum over the 64 lanes, fixed shape: 4 steps inside each row of 16, then row 0->1 and
// 2->3 (bcast15), then rows 0-1 -> 2-3 (bcast31).  Same association on every call => reproducible bits.
__device__ __forceinline__ double wave_scan_incl(double v) {
  // (the in-row shifts read 0 outside the row through bound_ctrl: no identity moved into the destination first)
  v += dpp_shr_zero<DPP_ROW_SHR1>(v);
  v += dpp_shr_zero<DPP_ROW_SHR2>(v);
  v += dpp_shr_zero<DPP_ROW_SHR4>(v);
  v += dpp_shr_zero<DPP_ROW_SHR8>(v);
  v += dpp_move<DPP_ROW_BCAST15, 0xa>(0.0, v);
  v += dpp_move<DPP_ROW_BCAST31, 0xc>(0.0, v);
  return v;
}
// maximum over the wave, returned in every lane
__device__ __forceinline__ double wave_max(double v) {
  const double ninf = -__builtin_huge_val();
  v = fmax(v, dpp_move<DPP_ROW_SHR1, 0xf>(ninf, v));
  v = fmax(v, dpp_move<DPP_ROW_SHR2, 0xf>(ninf, v));
  v = fmax(v, dpp_move<DPP_ROW_SHR4, 0xf>(ninf, v));
  v = fmax(v, dpp_move<DPP_ROW_SHR8, 0xf>(ninf, v));
  v = fmax(v, dpp_move<DPP_ROW_BCAST15, 0xa>(ninf, v));
  v = fmax(v, dpp_move<DPP_ROW_BCAST31, 0xc>(ninf, v));
  return lane_broadcast(v, 63);
}
// The same maximum through an order-preserving map to unsigned integers: the 32-bit maximum folds its DPP move into the
// instruction (v_max_u32_dpp: one instruction per step against five for a double -- two v_mov_dpp with an identity each, one
// v_max_f64), so two reductions -- the high words, then the low words of the lanes that hold the largest high word -- cost 18
// instructions where wave_max costs 32, and the result lands in scalar registers.  Exact (the value of one of the lanes); a NaN
// with a clear sign bit orders above +inf, so a NaN weight gives a NaN maximum (the filters then stop as degenerate, which is
// what a NaN weight did before through the sum).  Used by the row statistics of pf_canon.h: one maximum per row of 64 particles.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_umax_step(uint32_t v) {
  const uint32_t s = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true);
  return v > s ? v : s;
}
__device__ __forceinline__ uint32_t wave_umax(uint32_t v) {
  v = dpp_umax_step<DPP_ROW_SHR1, 0xf>(v);
  v = dpp_umax_step<DPP_ROW_SHR2, 0xf>(v);
  v = dpp_umax_step<DPP_ROW_SHR4, 0xf>(v);
  v = dpp_umax_step<DPP_ROW_SHR8, 0xf>(v);
  v = dpp_umax_step<DPP_ROW_BCAST15, 0xa>(v);
  v = dpp_umax_step<DPP_ROW_BCAST31, 0xc>(v);
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ double wave_max_key(double x) {
  const uint32_t hi = (uint32_t)__double2hiint(x), lo = (uint32_t)__double2loint(x);
  const uint32_t sg = (uint32_t)((int)hi >> 31);              // all ones for negative numbers
  const uint32_t kh = hi ^ (sg | 0x80000000u), kl = lo ^ sg;  // unsigned order of (kh, kl) = order of the doubles
  const uint32_t mh = wave_umax(kh);
  const uint32_t ml = wave_umax(kh == mh ? kl : 0u);
  const uint32_t rs = (mh & 0x80000000u) ? 0u : 0xFFFFFFFFu;  // undo the map (wave-uniform: scalar)
  return __hiloint2double((int)(mh ^ (rs | 0x80000000u)), (int)(ml ^ rs));
}
// sum over the wave in every lane (the last lane of the scan holds the total)
__device__ __forceinline__ double wave_sum(double v) { return lane_broadcast(wave_scan_incl(v), 63); }

// Resampling uniforms of a lane's FOUR rows (slots r * NT + tid, r = 0 ... 3; NT a multiple of 4) in a filter whose step
// draw is PACKED: the four lanes of a quad hold four consecutive slots in every row, i.e. they need the same four
// RNG_RESAMPLE blocks (one per row).  Lane q of the quad draws row q's block, and a 4 x 4 transpose inside the quad (DPP
// quad_perm broadcasts + selects: 28 instructions instead of three more Philox blocks) hands every lane word (lane & 3) of
// each row's block.  All four lanes of every quad must be active.
template <int R>
__device__ __forceinline__ uint32_t quad_bcast_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, R * 0x55, 0xf, 0xf, true);
}
__device__ __forceinline__ void rng_resample_words_quad4(RngKey k, uint32_t t, uint32_t NT, uint32_t tid, uint32_t (&uw)[4]) {
  const uint32_t q = tid & 3u;
  const u32x4 b = rng_resample_block(k, t, (q * NT + (tid & ~3u)) >> 2);
  uw[0] = u32x4_word(u32x4{quad_bcast_u32<0>(b.x), quad_bcast_u32<0>(b.y), quad_bcast_u32<0>(b.z), quad_bcast_u32<0>(b.w)}, q);
  uw[1] = u32x4_word(u32x4{quad_bcast_u32<1>(b.x), quad_bcast_u32<1>(b.y), quad_bcast_u32<1>(b.z), quad_bcast_u32<1>(b.w)}, q);
  uw[2] = u32x4_word(u32x4{quad_bcast_u32<2>(b.x), quad_bcast_u32<2>(b.y), quad_bcast_u32<2>(b.z), quad_bcast_u32<2>(b.w)}, q);
  uw[3] = u32x4_word(u32x4{quad_bcast_u32<3>(b.x), quad_bcast_u32<3>(b.y), quad_bcast_u32<3>(b.z), quad_bcast_u32<3>(b.w)}, q);
}

// ---------------------------------------------------------------- observation log-densities
// Unnormalised log g(y_t | signal s), src/model_ssm_ung.cpp:447-493; gaussian:
// src/model_ssm_ulg.cpp:582-586.  `ob` points at the obs-row header; state-
// independent quotients are precomputed on the host into OB_AUX:
//   svm (y/phi)^2, negbin y+phi, gamma y/u, gaussian 1/H^2.
template <class PT>
__device__ __forceinline__ double log_obs_density(int dist, PT ob, double phi, double s) {
  const double y = ob[OB_Y];
  switch (dist) {
    case DIST_SVM: return -0.5 * (s + ob[OB_AUX] * tl_exp(-s));
    case DIST_POISSON: return y * s - ob[OB_U] * tl_exp(s);
    case DIST_BINOMIAL: return y * s - ob[OB_U] * log1p(tl_exp(s));
    case DIST_NEGBIN: return y * s - ob[OB_AUX] * tl_log(phi + ob[OB_U] * tl_exp(s));
    case DIST_GAMMA: return -phi * (s + ob[OB_AUX] * tl_exp(-s));
    case DIST_GAUSSIAN: { const double r = y - s; return -0.5 * r * r * ob[OB_AUX]; }
  }
  return 0.0;
}

// psi-APF weight, src/model_ssm_ung.cpp:385-439 and the `- scales(t)` of :550,590
template <class PT>
__device__ __forceinline__ double log_psi_weight(int dist, PT ob, double phi, double s) {
  const double q = (ob[OB_PSIY] - s) * ob[OB_PSIIH];
  return log_obs_density(dist, ob, phi, s) + 0.5 * q * q;   // the caller subtracts scales_t once per row
}

}  // namespace bssm
