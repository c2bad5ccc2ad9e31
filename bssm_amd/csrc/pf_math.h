// pf_math.h -- fp64 elementary functions sized for the particle-filter kernels.
//
// The stock device-library exp/log/sincos are <1 ulp but spend ~100-200 instructions each on
// double-double arithmetic and on general range reduction.  A particle-step needs 5-9 of them
// (Box-Muller pairs, observation density, weight normalisation), and at one wave per SIMD their
// dependent chains ARE the kernel time.  These versions keep ~1 ulp accuracy with 20-35
// instructions by exploiting what the kernels know about their arguments:
//   * sincos only ever sees 2*pi*u with u a 32-bit fixed-point fraction, so the octant and the
//     reduced argument come straight from the integer bits (no Payne-Hanek, no Cody-Waite);
//   * log uses the hardware frexp instructions and the fdlibm minimax polynomial;
//   * exp uses a two-constant ln2 reduction and a degree-13 polynomial + v_ldexp_f64.
// Coefficients: fdlibm (e_log.c, k_sin.c, k_cos.c), Sun Microsystems, freely redistributable.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#else
typedef unsigned int uint32_t;
typedef int int32_t;
typedef unsigned long long uint64_t;
typedef long long int64_t;
#ifndef INFINITY
#define INFINITY (__builtin_huge_val())
#endif
#endif

#include "pf_math_tables.h"

namespace bssm {

#ifndef BSSM_FAST_MATH
#define BSSM_FAST_MATH 1
#endif

// 53-bit integer (hi 27 bits, lo 26 bits) -> double, exact
__device__ __forceinline__ double u53_to_double(uint32_t hi27, uint32_t lo26) {
  return __builtin_fma((double)hi27, 67108864.0, (double)lo26);
}

// f / d for d in [1.5, 2.5] and |f| < 1: v_rcp_f64 seed, two Newton steps on the reciprocal and one residual
// correction of the quotient (<= 1 ulp; the IEEE division sequence with its scaling and fix-up steps is not needed
// on this range and costs twice as many instructions)
__device__ __forceinline__ double div_near2(double f, double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  const double q = f * r;
  return __builtin_fma(__builtin_fma(-d, q, f), r, q);
}

// f / d for operands whose quotient, divisor and reciprocal are normal numbers (no scaling / fix-up steps): v_rcp_f64
// seed, two Newton steps, one residual correction -- <= 1 ulp at a third of the IEEE sequence's instructions.  Infinite or
// zero divisors give NaN / Inf like the division they replace, only denormal ranges lose accuracy.
__device__ __forceinline__ double fdiv(double f, double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  const double q = f * r;
  return __builtin_fma(__builtin_fma(-d, q, f), r, q);
}

// natural log; CHECKED = false skips the range test for arguments known to be normal positive numbers
// (the Box-Muller radius argument lies in [2^-53, 1])
template <bool CHECKED = true>
__device__ __forceinline__ double fast_log(double x) {
#if BSSM_FAST_MATH
  if constexpr (CHECKED) {
    if (__builtin_expect(!(x >= 2.2250738585072014e-308 && x <= 1.7976931348623157e308), 0)) return log(x);
  }
  double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
  int k = __builtin_amdgcn_frexp_exp(x);
  const bool lowhalf = m < 0.70710678118654752440;
  m = lowhalf ? m + m : m;  // [sqrt(1/2), sqrt(2))
  k = lowhalf ? k - 1 : k;
  const double f = m - 1.0;
  const double s = div_near2(f, 2.0 + f);
  const double dk = (double)k;
  const double z = s * s, w = z * z;
  const double t1 = w * __builtin_fma(w, __builtin_fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01),
                                      3.999999999940941908e-01);
  const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, 1.479819860511658591e-01,
                                                                        1.818357216161805012e-01),
                                                       2.857142874366239149e-01),
                                      6.666666666666735130e-01);
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  return dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
#else
  return log(x);
#endif
}

// e^x for any double (overflow -> inf, underflow -> 0, NaN -> NaN)
__device__ __forceinline__ double fast_exp(double x) {
#if BSSM_FAST_MATH
  const double xc = fmin(fmax(x, -800.0), 800.0);  // keeps k inside the ldexp range
  const double kd = __builtin_rint(xc * 1.44269504088896338700e+00);
  double r = __builtin_fma(-kd, 6.93147180369123816490e-01, xc);  // ln2_hi has 32 trailing zero bits: exact
  r = __builtin_fma(-kd, 1.90821492927058770002e-10, r);
  // degree-13 Taylor polynomial on |r| <= ln2/2 (truncation 4e-18), Horner, coefficients 1/k!
  double p = 1.6059043836821613e-10;
  p = __builtin_fma(p, r, 2.08767569878681e-09);
  p = __builtin_fma(p, r, 2.505210838544172e-08);
  p = __builtin_fma(p, r, 2.755731922398589e-07);
  p = __builtin_fma(p, r, 2.7557319223985893e-06);
  p = __builtin_fma(p, r, 2.48015873015873e-05);
  p = __builtin_fma(p, r, 1.984126984126984e-04);
  p = __builtin_fma(p, r, 1.388888888888889e-03);
  p = __builtin_fma(p, r, 8.333333333333333e-03);
  p = __builtin_fma(p, r, 4.1666666666666664e-02);
  p = __builtin_fma(p, r, 1.6666666666666666e-01);
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  const double res = __builtin_amdgcn_ldexp(p, (int)kd);
  return x != x ? x : res;
#else
  return exp(x);
#endif
}

// sin and cos of 2*pi*u for u = a * 2^-32 given as its integer bits
__device__ __forceinline__ void fast_sincos_2pi(uint32_t a, double& sn, double& cs) {
#if BSSM_FAST_MATH
  const uint32_t q = a >> 29;                       // octant 0..7
  uint32_t fr = a & 0x1FFFFFFFu;                    // 29-bit fraction inside the octant
  if (q & 1u) fr = 0x20000000u - fr;                // reflect (exact in integers)
  // theta = 2*pi * frac * 2^-32  in [0, pi/4]
  const double th = (double)fr * (6.283185307179586476925 * 0x1.0p-32);
  const double z = th * th;
  // fdlibm k_sin.c / k_cos.c on [-pi/4, pi/4]
  const double rs = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z,
                    1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06),
                    -1.98412698298579493134e-04), 8.33333333332248946124e-03), -1.66666666666666324348e-01);
  const double s0 = __builtin_fma(z * th, rs, th);
  const double rc = z * __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z,
                    -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07),
                    2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
  const double hz = 0.5 * z, wv = 1.0 - hz;
  const double c0 = wv + (((1.0 - wv) - hz) + z * rc);
  const bool swap = ((q + 1u) & 2u) != 0u;          // octants 1,2,5,6: sin <-> cos
  const double sa = swap ? c0 : s0, ca = swap ? s0 : c0;
  sn = (q >= 4u) ? -sa : sa;
  cs = ((q + 2u) & 4u) ? -ca : ca;                  // octants 2,3,4,5: cos negative
#else
  sincos(2.0 * 3.14159265358979323846 * ((double)a * 0x1.0p-32), &sn, &cs);
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// Table-driven versions (round 2).  The polynomial forms above spend 25-45 fp64 instructions per call and a quarter-rate
// v_rcp_f64; measured issue costs on gfx950 (scripts/ubench/valu_rates.hip): every fp64 / VOP3 instruction 4.4-4.7
// cycles per wave, plain 32-bit VOP2 2.6, v_rcp/v_rsq/v_sqrt_f64 16.4.  With a 6.5 KB table set in LDS (one
// ds_read_b128 / b64 per call, on the LDS pipe, not the VALU) the reduced argument is small enough for a degree 4-7
// polynomial: log 19, exp 20, sincos 21 instructions, no reciprocal.  <= 1.5 ulp (tests/test_gpu_math.py).
//
// The tables live in LDS: a kernel calls math_tab_init() once (all threads, before the first use; it ends with a
// workgroup barrier); device code then reaches them through math_tab().
static __device__ const double g_math_tab[kMathTabDoubles] __attribute__((aligned(16))) = {BSSM_MATH_TABLE_VALUES};

__device__ __forceinline__ double* math_tab() {
  __shared__ __attribute__((aligned(16))) double s_math_tab[kMathTabDoubles];
  return s_math_tab;
}
__device__ __forceinline__ void math_tab_init() {
  typedef double d2 __attribute__((ext_vector_type(2)));
  d2* dst = (d2*)math_tab();
  const d2* src = (const d2*)g_math_tab;
  for (int q = threadIdx.x; q < kMathTabDoubles / 2; q += blockDim.x) dst[q] = src[q];
  __asm__ volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// split form for 256-thread kernels: issue the global loads first (begin), store them to LDS later (end), so that the
// round trip overlaps whatever the kernel loads next; the caller's next workgroup barrier publishes the table
struct MathTabRegs { double v[4]; };
__device__ __forceinline__ MathTabRegs math_tab_load_begin256() {
  static_assert(kMathTabDoubles <= 4 * 256, "table does not fit two 16-byte loads per thread");
  typedef double d2 __attribute__((ext_vector_type(2)));
  const d2* src = (const d2*)g_math_tab;
  const int q0 = threadIdx.x, q1 = threadIdx.x + 256;
  const d2 a = src[q0];
  const d2 b = src[q1 < kMathTabDoubles / 2 ? q1 : q0];
  return MathTabRegs{{a.x, a.y, b.x, b.y}};
}
__device__ __forceinline__ void math_tab_load_end256(const MathTabRegs& r) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  d2* dst = (d2*)math_tab();
  dst[threadIdx.x] = d2{r.v[0], r.v[1]};
  if (threadIdx.x + 256 < kMathTabDoubles / 2) dst[threadIdx.x + 256] = d2{r.v[2], r.v[3]};
}

__device__ __forceinline__ uint32_t dbl_hi(double x) { return (uint32_t)__double2hiint(x); }
__device__ __forceinline__ uint32_t dbl_lo(double x) { return (uint32_t)__double2loint(x); }

// natural log.  x = 2^k z with z in [0.6865, 1.373); interval i of the table (uniform in the bits of z, 7 index bits)
// has centre c_i, invc ~ 1/c_i, logc = -log(invc); r = z invc - 1 (one fma), |r| < 2^-7.9;
// log x = k ln2 + logc + log1p(r), log1p by its Taylor polynomial to degree 7 (Estrin).  z = 1 is the centre of its
// interval with invc = 1, logc = 0 exactly, so results near 0 keep full relative accuracy.
template <bool CHECKED = true>
__device__ __forceinline__ double tl_log(double x) {
  const uint32_t hx = dbl_hi(x);
  if constexpr (CHECKED) {  // zero, negative, subnormal, inf, nan: the library routine
    if (__builtin_expect(hx - 0x00100000u >= 0x7FE00000u, 0)) return log(x);
  }
  const uint32_t tmp = hx - 0x3FE5F000u;
  const int k = (int)tmp >> 20;
  const uint32_t i2 = (tmp >> 9) & 0x7F0u;  // byte offset of entry i = (tmp >> 13) & 127 (16 bytes each)
  const double z = __hiloint2double((int)(hx - (tmp & 0xFFF00000u)), (int)dbl_lo(x));
  typedef double d2 __attribute__((ext_vector_type(2)));
  const d2 e = *(const d2*)((const char*)(math_tab() + kMathTabLog) + i2);
  const double r = __builtin_fma(z, e.x, -1.0);
  const double kd = (double)k;
  const double w = __builtin_fma(kd, 6.93147180369123816490e-01, e.y);
  const double r2 = r * r;
  const double a = __builtin_fma(r, 3.33333333333333333e-01, -0.5);
  const double b = __builtin_fma(r, 0.2, -0.25);
  const double c = __builtin_fma(r, 1.42857142857142857e-01, -1.66666666666666667e-01);
  const double q = __builtin_fma(r2, __builtin_fma(r2, c, b), a);
  double t = __builtin_fma(kd, 1.90821492927058770002e-10, r);
  t = __builtin_fma(r2, q, t);
  return w + t;
}

// e^x for any double (overflow -> inf, underflow -> 0, NaN -> NaN): x = (64 e + j) ln2/64 + r, |r| <= ln2/128,
// e^x = 2^e * T[j] * (1 + expm1(r)), expm1 by its Taylor polynomial to degree 5
__device__ __forceinline__ double tl_exp(double x) {
  const double xc = fmin(fmax(x, -800.0), 800.0);
  const double kd = __builtin_rint(xc * 9.23324826168936567e+01);                // 64 / ln2
  double r = __builtin_fma(-kd, 1.08304246932675596327e-02, xc);                 // ln2_hi / 64: 32 significant bits, exact product
  r = __builtin_fma(-kd, 2.98158582698529328128e-12, r);                         // ln2_lo / 64
  const int ki = (int)kd;
  const double T = math_tab()[kMathTabExp + (ki & 63)];
  const double r2 = r * r;
  double q = __builtin_fma(r, 8.33333333333333333e-03, 4.16666666666666667e-02);
  q = __builtin_fma(r, q, 1.66666666666666667e-01);
  q = __builtin_fma(r, q, 0.5);
  const double p = __builtin_fma(r2, q, r);
  const double res = __builtin_amdgcn_ldexp(__builtin_fma(T, p, T), ki >> 6);
  return x != x ? x : res;
}

// sin and cos of 2*pi*a*2^-32: the top 8 bits select one of 256 tabulated angles (interval centres), the low 24 bits
// are the signed offset d from it, |d| <= 2*pi/512; sin/cos(centre + d) by the addition formulas with short Taylor
// polynomials for sin d and cos d - 1
__device__ __forceinline__ void tl_sincos_2pi(uint32_t a, double& sn, double& cs) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  const d2 e = *(const d2*)((const char*)(math_tab() + kMathTabSin) + ((a >> 20) & 0xFF0u));
  const int d256 = (int)((a << 8) ^ 0x80000000u);                                // 256 * (low 24 bits - 2^23), exact
  const double d = (double)d256 * (6.283185307179586476925 * 0x1.0p-40);
  const double z = d * d;
  const double sd = __builtin_fma(d * z, __builtin_fma(z, __builtin_fma(z, -1.98412698412698413e-04, 8.33333333333333333e-03),
                                                       -1.66666666666666667e-01), d);
  const double cm = z * __builtin_fma(z, __builtin_fma(z, -1.38888888888888889e-03, 4.16666666666666667e-02), -0.5);
  sn = e.x + __builtin_fma(e.y, sd, e.x * cm);
  cs = e.y + __builtin_fma(-e.x, sd, e.y * cm);
}

}  // namespace bssm
