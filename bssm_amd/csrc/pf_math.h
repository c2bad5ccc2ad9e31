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
#include <hip/hip_runtime.h>
#include <stdint.h>

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

}  // namespace bssm
