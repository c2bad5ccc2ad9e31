// Standalone check of lg_smoother_wave.h (LgWave<M>) and lg_factors_rows.h (LgFactorRows<M>) at M = 32 / 64 against the host routines of
// lg_smoother.h on random padded models.  Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -I bssm_amd/csrc scripts/lgwave_check.hip -o /tmp/lgwave_check   (-DLGW_STAMPS: phase clocks of the 64-state smoother)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define LGF_STATS 1
#include "lg_smoother.h"
#include "lg_smoother_wave.h"
#include "lg_factors_rows.h"

using namespace bssm;

struct Dev {
  int F, n, mr;
  const double *Z, *T, *RR, *D, *C, *y, *H, *a1, *P1;
  double *at, *Pt, *cc, *K, *v, *Fs, *Vf, *Cf;
  int* ok;
  int* flags;   // [F][n + 1]: which (draw, time point) the full factor kernel still has to do (written by the thin one)
};

template <int M>
__global__ void __launch_bounds__(LgWave<M>::kThreads, (M <= 32 ? 2 : 1)) smooth_kernel(const Dev b) {
  constexpr int G = 64 / M;
  __shared__ double lds[LgWave<M>::kLdsDoubles];
  const int d = min((int)(blockIdx.x * G + (threadIdx.x & 63) / M), b.F - 1);
  LgModel g;
  g.n = b.n;
  g.Z = b.Z + d; g.Zs = 0; g.T = b.T + d; g.Ts = 0; g.RR = b.RR + d; g.Rs = 0; g.D = b.D + d; g.Ds = 0; g.C = b.C + d; g.Cs = 0;
  g.y = b.y + d; g.H = b.H + d; g.Hs = 1; g.a1 = b.a1 + d; g.P1 = b.P1 + d;
  LgWave<M> w(lds);
  w.mr = b.mr;
  const bool ok = w.run(g, b.at + d, b.Pt + d, b.cc + d, b.K + d, b.v + d, b.Fs + d, (size_t)b.F);
  b.ok[d] = ok ? 1 : 0;
}
template <int M>
__global__ void __launch_bounds__(LgFactorRows<M>::kThreads, (M >= 32 ? 2 : 1)) factors_kernel(const Dev b) {
  constexpr int G = 64 / M;
  __shared__ double lds[LgFactorRows<M>::kLdsDoubles];
  const int d = min((int)(blockIdx.x * G + (threadIdx.x & 63) / M), b.F - 1);
  LgFactorRows<M> rows(lds);
  rows.skip = b.ok[d] == 0;
  rows.mr = b.mr;
  const bool fine = rows.cond_step((int)blockIdx.y, b.Pt + d, b.cc + d, b.Vf + d, b.Cf + d, (size_t)b.F, 0, blockIdx.y > 0 ? b.flags + (size_t)d * (b.n + 1) + blockIdx.y : nullptr);
  if (!fine && !rows.skip && rows.r == 0 && rows.w == 0) b.ok[d] = 0;
}
// the thin path of time point blockIdx.y + 1 (as ung_factors_thin_kernel of pf_kernels.hip)
template <int M>
__global__ void __launch_bounds__(LgFactorRows<M>::kThreads, (M >= 32 ? 2 : 1)) factors_thin_kernel(const Dev b) {
  constexpr int G = 64 / M;
  __shared__ double lds[LgFactorRows<M>::kLdsDoubles];
  const int d = min((int)(blockIdx.x * G + (threadIdx.x & 63) / M), b.F - 1);
  const int t = (int)blockIdx.y + 1;
  LgFactorRows<M> rows(lds);
  rows.skip = b.ok[d] == 0;
  rows.mr = b.mr;
  const bool fine = rows.cond_step_thin(t, b.Pt + d, b.cc + d, b.Vf + d, b.Cf + d, (size_t)b.F, 0, b.flags + (size_t)d * (b.n + 1) + t);
  if (!fine && !rows.skip && rows.r == 0 && rows.w == 0) b.ok[d] = 0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(1); } } while (0)

template <int M>
int check(int mr, int n, int F, unsigned seed) {
  constexpr int MM = M * M;
  std::mt19937_64 rng(seed);
  std::normal_distribution<double> nd;
  std::uniform_real_distribution<double> ud(0.0, 1.0);
  // per draw model (time-invariant Z, T, RR, D, C), padded with zeros beyond mr
  std::vector<double> Z((size_t)F * M, 0.0), T((size_t)F * MM, 0.0), RR((size_t)F * MM, 0.0), D(F, 0.0), C((size_t)F * M, 0.0), y((size_t)F * n), H((size_t)F * n),
      a1((size_t)F * M, 0.0), P1((size_t)F * MM, 0.0);
  for (int d = 0; d < F; d++) {
    // a seasonal-like structure: level + slope + dummy seasonal, plus a small random part
    double* Td = &T[(size_t)d * MM];
    Td[0 + M * 0] = 1.0; if (mr > 1) { Td[0 + M * 1] = 1.0; Td[1 + M * 1] = 1.0; }
    if (mr > 2) {
      for (int j = 2; j < mr; j++) Td[2 + M * j] = -1.0;
      for (int i = 3; i < mr; i++) Td[i + M * (i - 1)] = 1.0;
    }
    for (int j = 0; j < mr; j++) for (int i = 0; i < mr; i++) Td[i + M * j] += 0.01 * nd(rng);
    const int k = std::min(mr, 3);
    std::vector<double> R((size_t)mr * k, 0.0);
    for (int c = 0; c < k; c++) R[c + (size_t)mr * c] = 0.05 + 0.2 * ud(rng);
    for (int j = 0; j < mr; j++) for (int i = 0; i < mr; i++) { double s = 0; for (int c = 0; c < k; c++) s += R[i + (size_t)mr * c] * R[j + (size_t)mr * c]; RR[(size_t)d * MM + i + M * j] = s; }
    Z[(size_t)d * M] = 1.0; if (mr > 2) Z[(size_t)d * M + 2] = 1.0;
    for (int i = 0; i < mr; i++) { Z[(size_t)d * M + i] += 0.01 * nd(rng); C[(size_t)d * M + i] = 0.01 * nd(rng); a1[(size_t)d * M + i] = 0.3 * nd(rng); P1[(size_t)d * MM + i + M * i] = 0.2 + ud(rng); }
    D[d] = 0.1 * nd(rng);
    for (int t = 0; t < n; t++) { y[(size_t)d * n + t] = nd(rng) * 2.0; H[(size_t)d * n + t] = 0.3 + ud(rng); }
    y[(size_t)d * n + 3] = NAN;
  }
  // host reference, draw by draw
  const size_t n1 = (size_t)n + 1;
  std::vector<double> h_at((size_t)F * n1 * M), h_Pt((size_t)F * n1 * MM), h_cc((size_t)F * n1 * MM), h_Vf((size_t)F * n1 * MM), h_Cf((size_t)F * n1 * MM, 0.0);
  std::vector<int> h_ok(F);
  const auto t0 = std::chrono::steady_clock::now();
  for (int d = 0; d < F; d++) {
    LgModel g;
    g.n = n;
    g.Z = &Z[(size_t)d * M]; g.Zs = 0; g.T = &T[(size_t)d * MM]; g.Ts = 0; g.RR = &RR[(size_t)d * MM]; g.Rs = 0; g.D = &D[d]; g.Ds = 0; g.C = &C[(size_t)d * M]; g.Cs = 0;
    g.y = &y[(size_t)d * n]; g.H = &H[(size_t)d * n]; g.Hs = 1; g.a1 = &a1[(size_t)d * M]; g.P1 = &P1[(size_t)d * MM];
    std::vector<double> K((size_t)n * M), v(n), Fs(n);
    bool ok = LgSmoother<M>::run(g, &h_at[(size_t)d * n1 * M], &h_Pt[(size_t)d * n1 * MM], &h_cc[(size_t)d * n1 * MM], K.data(), v.data(), Fs.data(), 1, false);
    for (int t = 0; t <= n && ok; t++) ok = LgSmoother<M>::cond_step(t, &h_Pt[(size_t)d * n1 * MM], &h_cc[(size_t)d * n1 * MM], &h_Vf[(size_t)d * n1 * MM], &h_Cf[(size_t)d * n1 * MM], 1);
    h_ok[d] = ok;
  }
  const double host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  // device, draw-interleaved
  auto inter = [&](const std::vector<double>& src, size_t per) {
    std::vector<double> o(src.size());
    for (int d = 0; d < F; d++) for (size_t q = 0; q < per; q++) o[q * F + d] = src[(size_t)d * per + q];
    return o;
  };
  auto up = [&](const std::vector<double>& hsrc) { double* p; CK(hipMalloc(&p, hsrc.size() * 8)); CK(hipMemcpy(p, hsrc.data(), hsrc.size() * 8, hipMemcpyHostToDevice)); return p; };
  auto dalloc = [&](size_t cnt) { double* p; CK(hipMalloc(&p, cnt * 8)); CK(hipMemset(p, 0, cnt * 8)); return p; };
  Dev b{};
  b.F = F; b.n = n; b.mr = mr;
  b.Z = up(inter(Z, M)); b.T = up(inter(T, MM)); b.RR = up(inter(RR, MM)); b.D = up(D); b.C = up(inter(C, M)); b.y = up(inter(y, n)); b.H = up(inter(H, n));
  b.a1 = up(inter(a1, M)); b.P1 = up(inter(P1, MM));
  b.at = dalloc((size_t)F * n1 * M); b.Pt = dalloc((size_t)F * n1 * MM); b.cc = dalloc((size_t)F * n1 * MM); b.K = dalloc((size_t)F * n * M); b.v = dalloc((size_t)F * n); b.Fs = dalloc((size_t)F * n);
  b.Vf = dalloc((size_t)F * n1 * MM); b.Cf = dalloc((size_t)F * n1 * MM);
  CK(hipMalloc(&b.ok, F * sizeof(int)));
  CK(hipMalloc(&b.flags, (size_t)F * n1 * sizeof(int)));
  CK(hipMemset(b.flags, 0, (size_t)F * n1 * sizeof(int)));
  constexpr int G = 64 / M;
  hipEvent_t e0, e1, e2;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((smooth_kernel<M>), dim3((F + G - 1) / G), dim3(LgWave<M>::kThreads), 0, 0, b);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  std::vector<double> g_at((size_t)F * n1 * M), g_Pt((size_t)F * n1 * MM), g_cc((size_t)F * n1 * MM), g_Vf((size_t)F * n1 * MM), g_Cf((size_t)F * n1 * MM);
  CK(hipMemcpy(g_at.data(), b.at, g_at.size() * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(g_Pt.data(), b.Pt, g_Pt.size() * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(g_cc.data(), b.cc, g_cc.size() * 8, hipMemcpyDeviceToHost));
  hipEvent_t e3;
  CK(hipEventCreate(&e3));
  CK(hipEventRecord(e1));
  hipLaunchKernelGGL((factors_thin_kernel<M>), dim3((F + G - 1) / G, n), dim3(LgFactorRows<M>::kThreads), 0, 0, b);
  CK(hipEventRecord(e3));
  hipLaunchKernelGGL((factors_kernel<M>), dim3((F + G - 1) / G, n + 1), dim3(LgFactorRows<M>::kThreads), 0, 0, b);
  CK(hipEventRecord(e2));
  CK(hipDeviceSynchronize());
  float ms_f = 0, ms_thin = 0;
  CK(hipEventElapsedTime(&ms_f, e1, e2));
  CK(hipEventElapsedTime(&ms_thin, e1, e3));
  {
    std::vector<int> fl((size_t)F * n1);
    CK(hipMemcpy(fl.data(), b.flags, fl.size() * sizeof(int), hipMemcpyDeviceToHost));
    size_t full = 0;
    for (int q : fl) full += q != 0;
    std::printf("   thin kernel %.3f ms; %zu of %zu (draw, time point >= 1) left to the full kernel\n", ms_thin, full, (size_t)F * n);
  }
  {
    unsigned long long st[8] = {0}, zero[8] = {0};
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(bssm::lgf_stats), sizeof(st)));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(bssm::lgf_stats), zero, sizeof(zero)));
    std::printf("   factors: %llu decompositions, %.2f sweeps each, %.0f clocks (100 MHz) per decomposition; %llu steps behind t = 0 at %.0f clocks each\n", st[0], (double)st[1] / (double)st[0],
                (double)st[2] / (double)st[0], st[4], (double)st[3] / (double)(st[4] ? st[4] : 1));
  }
  CK(hipMemcpy(g_Vf.data(), b.Vf, g_Vf.size() * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(g_Cf.data(), b.Cf, g_Cf.size() * 8, hipMemcpyDeviceToHost));
  std::vector<int> g_ok(F);
  CK(hipMemcpy(g_ok.data(), b.ok, F * sizeof(int), hipMemcpyDeviceToHost));
  // timing of the smoother alone
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((smooth_kernel<M>), dim3((F + G - 1) / G), dim3(LgWave<M>::kThreads), 0, 0, b);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms_s = 0;
  CK(hipEventElapsedTime(&ms_s, e0, e1));
  auto cmp = [&](const char* name, const std::vector<double>& hh, const std::vector<double>& gg, size_t per) {
    double worst = 0.0, scale = 0.0;
    for (int d = 0; d < F; d++) for (size_t q = 0; q < per; q++) {
      const double a = hh[(size_t)d * per + q], g2 = gg[q * F + d];
      scale = std::max(scale, std::fabs(a));
      const double df = std::fabs(a - g2);
      if (!(df <= worst)) worst = df;   // (NaN sticks)
    }
    std::printf("   %-4s max |host - device| %.3e (largest |host| %.3e)\n", name, worst, scale);
    return worst;
  };
#ifdef LGW_STAMPS
  {
    unsigned long long st[32] = {0}, zero[32] = {0};
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(bssm::lgw_stamps), sizeof(st)));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(bssm::lgw_stamps), zero, sizeof(zero)));
    std::printf("   smoother phases (100 MHz clocks per time point, two runs of thread 0 / workgroup 0):");
    for (int k = 0; k < 21; k++) std::printf(" %d:%.0f", k, (double)st[k] / (2.0 * n));
    std::printf("\n");
  }
#endif
  std::printf("M = %d (m = %d), n = %d, F = %d: host %.1f ms; device smoother %.3f ms, factors %.3f ms; ok host/device:", M, mr, n, F, host_ms, ms_s, ms_f);
  int okdiff = 0;
  for (int d = 0; d < F && d < 8; d++) std::printf(" %d/%d", h_ok[d], g_ok[d]);
  for (int d = 0; d < F; d++) okdiff += h_ok[d] != g_ok[d];
  std::printf("\n");
  double w = 0;
  w = std::max(w, cmp("at", h_at, g_at, n1 * M));
  w = std::max(w, cmp("Pt", h_Pt, g_Pt, n1 * MM));
  w = std::max(w, cmp("cc", h_cc, g_cc, n1 * MM));
  w = std::max(w, cmp("Vf", h_Vf, g_Vf, n1 * MM));
  w = std::max(w, cmp("Cf", h_Cf, g_Cf, n1 * MM));
  for (const double* p : {b.Z, b.T, b.RR, b.D, b.C, b.y, b.H, b.a1, b.P1}) CK(hipFree((void*)p));
  for (double* p : {b.at, b.Pt, b.cc, b.K, b.v, b.Fs, b.Vf, b.Cf}) CK(hipFree(p));
  CK(hipFree(b.ok));
  CK(hipFree(b.flags));
  return (w < 1e-8 && okdiff == 0) ? 0 : 1;
}

int main(int argc, char** argv) {
  int rc = 0;
  rc |= check<16>(13, 30, 7, 9);
  rc |= check<16>(16, 12, 4, 10);
  rc |= check<32>(25, 40, 5, 11);
  rc |= check<32>(32, 20, 4, 12);
  rc |= check<64>(53, 30, 3, 13);
  if (argc > 1) {  // timing at a batch size
    rc |= check<16>(13, 72, 2048, 16);
    rc |= check<32>(25, 100, 512, 14);
    rc |= check<64>(53, 100, 256, 15);
  }
  std::printf(rc ? "FAILED\n" : "all fine\n");
  return rc;
}
