// Host emulation of the table-driven log / exp / sincos of pf_math.h (same operation sequence, fma = one rounding)
// against long-double libm: max error in ulp.  g++ -O2 -mfma math_check.cpp -o math_check
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include "../../bssm_amd/csrc/pf_math_tables.h"
using namespace bssm;
static const double tab[kMathTabDoubles] = {BSSM_MATH_TABLE_VALUES};
static uint32_t hi(double x) { uint64_t u; memcpy(&u, &x, 8); return u >> 32; }
static uint32_t lo(double x) { uint64_t u; memcpy(&u, &x, 8); return (uint32_t)u; }
static double mk(uint32_t h, uint32_t l) { uint64_t u = ((uint64_t)h << 32) | l; double x; memcpy(&x, &u, 8); return x; }
static double tl_log(double x) {
  const uint32_t hx = hi(x);
  if (hx - 0x00100000u >= 0x7FE00000u) return log(x);
  const uint32_t tmp = hx - 0x3FE5F000u;
  const int k = (int)tmp >> 20;
  const uint32_t i = (tmp >> 13) & 127u;
  const double z = mk(hx - (tmp & 0xFFF00000u), lo(x));
  const double invc = tab[2 * i], logc = tab[2 * i + 1];
  const double r = fma(z, invc, -1.0), kd = k;
  const double w = fma(kd, 6.93147180369123816490e-01, logc);
  const double r2 = r * r;
  const double a = fma(r, 3.33333333333333333e-01, -0.5), b = fma(r, 0.2, -0.25), c = fma(r, 1.42857142857142857e-01, -1.66666666666666667e-01);
  const double q = fma(r2, fma(r2, c, b), a);
  double t = fma(kd, 1.90821492927058770002e-10, r);
  t = fma(r2, q, t);
  return w + t;
}
static double tl_exp(double x) {
  const double xc = fmin(fmax(x, -800.0), 800.0);
  const double kd = rint(xc * 9.23324826168936567e+01);
  double r = fma(-kd, 1.08304246932675596327e-02, xc);
  r = fma(-kd, 2.98158582698529328128e-12, r);
  const int ki = (int)kd;
  const double T = tab[kMathTabExp + (ki & 63)];
  const double r2 = r * r;
  double q = fma(r, 8.33333333333333333e-03, 4.16666666666666667e-02);
  q = fma(r, q, 1.66666666666666667e-01);
  q = fma(r, q, 0.5);
  const double p = fma(r2, q, r);
  return ldexp(fma(T, p, T), ki >> 6);
}
static void tl_sincos(uint32_t a, double& sn, double& cs) {
  const uint32_t j = a >> 24;
  const double s0 = tab[kMathTabSin + 2 * j], c0 = tab[kMathTabSin + 2 * j + 1];
  const int d256 = (int)((a << 8) ^ 0x80000000u);
  const double d = (double)d256 * (6.283185307179586476925 * 0x1.0p-40);
  const double z = d * d;
  const double sd = fma(d * z, fma(z, fma(z, -1.98412698412698413e-04, 8.33333333333333333e-03), -1.66666666666666667e-01), d);
  const double cm = z * fma(z, fma(z, -1.38888888888888889e-03, 4.16666666666666667e-02), -0.5);
  sn = s0 + fma(c0, sd, s0 * cm);
  cs = c0 + fma(-s0, sd, c0 * cm);
}
static double ulp_err(double got, long double ref) {
  if (ref == 0) return fabs(got) > 0 ? 1e9 : 0;
  int e; frexpl(ref, &e);
  long double ulp = ldexpl(1.0L, e - 53);
  if (fabsl(ref) < 2.2250738585072014e-308L) ulp = 4.9406564584124654e-324L;
  return (double)(fabsl((long double)got - ref) / ulp);
}
int main() {
  std::mt19937_64 g(1);
  double me = 0, wx = 0;
  for (int it = 0; it < 20000000; it++) {
    double x;
    switch (it & 3) {
      case 0: x = ldexp((double)(g() >> 11) + 1.0, -53); break;                       // Box-Muller argument
      case 1: x = exp(((double)(g() >> 11) * 0x1p-53 - 0.5) * 1400.0); break;          // whole range
      case 2: x = 1.0 + ((double)(g() >> 11) * 0x1p-53 - 0.5) * 0.05; break;           // near 1
      default: x = (double)(g() >> 11) * 0x1p-53 * 100.0 + 1e-3; break;
    }
    const double e = ulp_err(tl_log(x), logl((long double)x));
    if (e > me) { me = e; wx = x; }
  }
  printf("log     max err %.3f ulp at x = %.17g\n", me, wx);
  me = 0;
  for (int it = 0; it < 20000000; it++) {
    double x = (it & 1) ? ((double)(g() >> 11) * 0x1p-53 - 0.5) * 1400.0 : ((double)(g() >> 11) * 0x1p-53 - 0.5) * 40.0;
    const double e = ulp_err(tl_exp(x), expl((long double)x));
    if (e > me) { me = e; wx = x; }
  }
  printf("exp     max err %.3f ulp at x = %.17g   exp(-inf)=%g exp(800)=%g exp(-745.2)=%g\n", me, wx, tl_exp(-INFINITY), tl_exp(800), tl_exp(-745.2));
  me = 0; double mc = 0, mabs = 0;
  const long double twopi = 6.283185307179586476925286766559L;
  for (int it = 0; it < 20000000; it++) {
    const uint32_t a = (uint32_t)g();
    double s, c; tl_sincos(a, s, c);
    const long double th = twopi * (long double)a * 0x1p-32L;
    const long double rs = sinl(th), rc = cosl(th);
    mabs = fmax(mabs, fmax((double)fabsl(s - rs), (double)fabsl(c - rc)));
    if (fabsl(rs) > 1e-3) me = fmax(me, ulp_err(s, rs));
    if (fabsl(rc) > 1e-3) mc = fmax(mc, ulp_err(c, rc));
  }
  printf("sincos  max err sin %.3f cos %.3f ulp (|value| > 1e-3), max abs err %.3g\n", me, mc, mabs);
  return 0;
}
