// Issue cost of the VALU instructions the particle-step kernel is made of (gfx950): cycles per wave64 instruction per SIMD.
// hipcc --offload-arch=gfx950 -O2 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define REP8(x) x x x x x x x x
#define ITER 2000

#define KERNEL_F64(name, asmline)                                                        \
  __global__ void __launch_bounds__(256) name(double* out, double seed) {                \
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, \
           a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;                                       \
    double b = seed * 0.5, c = seed * 0.25;                                              \
    for (int i = 0; i < ITER; i++) {                                                     \
      asm volatile(asmline(0) asmline(1) asmline(2) asmline(3) asmline(4) asmline(5)     \
                   asmline(6) asmline(7)                                                 \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5),         \
                     "+v"(a6), "+v"(a7)                                                  \
                   : "v"(b), "v"(c));                                                    \
    }                                                                                    \
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;        \
  }
#define A_FMA(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define A_ADD(i) "v_add_f64 %" #i ", %" #i ", %8\n"
#define A_MUL(i) "v_mul_f64 %" #i ", %" #i ", %8\n"
#define A_RCP(i) "v_rcp_f64 %" #i ", %" #i "\n"
#define A_RSQ(i) "v_rsq_f64 %" #i ", %" #i "\n"
#define A_SQRT(i) "v_sqrt_f64 %" #i ", %" #i "\n"
#define A_RNDNE(i) "v_rndne_f64 %" #i ", %" #i "\n"
#define A_FREXPM(i) "v_frexp_mant_f64 %" #i ", %" #i "\n"
#define A_MAX(i) "v_max_f64 %" #i ", %" #i ", %8\n"
#define A_MOV64(i) "v_mov_b64 %" #i ", %8\n"
#define A_FMAC(i) "v_fmac_f64 %" #i ", %8, %9\n"
#define A_PKFMA32(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
KERNEL_F64(k_fma, A_FMA)
KERNEL_F64(k_add, A_ADD)
KERNEL_F64(k_mul, A_MUL)
KERNEL_F64(k_rcp, A_RCP)
KERNEL_F64(k_rsq, A_RSQ)
KERNEL_F64(k_sqrt, A_SQRT)
KERNEL_F64(k_rndne, A_RNDNE)
KERNEL_F64(k_frexpm, A_FREXPM)
KERNEL_F64(k_max, A_MAX)
KERNEL_F64(k_mov64, A_MOV64)
KERNEL_F64(k_fmac, A_FMAC)
KERNEL_F64(k_pkfma32, A_PKFMA32)

#define KERNEL_U32(name, asmline)                                                        \
  __global__ void __launch_bounds__(256) name(double* out, double seed) {                \
    uint32_t a0 = (uint32_t)seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3,  \
             a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;                        \
    uint32_t b = (uint32_t)(seed * 3) | 1u, c = (uint32_t)(seed * 7);                    \
    for (int i = 0; i < ITER; i++) {                                                     \
      asm volatile(asmline(0) asmline(1) asmline(2) asmline(3) asmline(4) asmline(5)     \
                   asmline(6) asmline(7)                                                 \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5),         \
                     "+v"(a6), "+v"(a7)                                                  \
                   : "v"(b), "v"(c));                                                    \
    }                                                                                    \
    out[blockIdx.x * 256 + threadIdx.x] = (double)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7); \
  }
#define U_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define U_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define U_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define U_MULHI(i) "v_mul_hi_u32 %" #i ", %" #i ", %8\n"
#define U_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define U_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define U_ALIGN(i) "v_alignbit_b32 %" #i ", %" #i ", %" #i ", 13\n"
#define U_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define U_DPP(i) "v_mov_b32_dpp %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define U_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define U_XAD(i) "v_xad_u32 %" #i ", %" #i ", %8, %9\n"
#define U_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 3, 9\n"
#define U_CVTF64(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define U_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define U_READLANE(i) "v_readlane_b32 s20, %" #i ", 5\n"
KERNEL_U32(k_xor, U_XOR)
KERNEL_U32(k_addu, U_ADD)
KERNEL_U32(k_mullo, U_MULLO)
KERNEL_U32(k_mulhi, U_MULHI)
KERNEL_U32(k_mul24, U_MUL24)
KERNEL_U32(k_mad24, U_MAD24)
KERNEL_U32(k_align, U_ALIGN)
KERNEL_U32(k_cndmask, U_CNDMASK)
KERNEL_U32(k_dpp, U_DPP)
KERNEL_U32(k_add3, U_ADD3)
KERNEL_U32(k_xad, U_XAD)
KERNEL_U32(k_bfe, U_BFE)
KERNEL_U32(k_cvtf32, U_CVTF64)
KERNEL_U32(k_perm, U_PERM)

// 64-bit results: v_mad_u64_u32, v_cvt_f64_u32, v_ldexp_f64, v_lshl_add_u64
__global__ void __launch_bounds__(256) k_mad64(double* out, double seed) {
  uint64_t a[8];
  for (int j = 0; j < 8; j++) a[j] = (uint64_t)seed + threadIdx.x + j;
  uint32_t b = (uint32_t)(seed * 3) | 1u;
  for (int i = 0; i < ITER; i++) {
#pragma unroll
    for (int j = 0; j < 8; j++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n" : "+v"(a[j]) : "v"((uint32_t)j + b), "v"(b) : "vcc");
  }
  uint64_t s = 0;
  for (int j = 0; j < 8; j++) s += a[j];
  out[blockIdx.x * 256 + threadIdx.x] = (double)s;
}
__global__ void __launch_bounds__(256) k_cvt_f64_u32(double* out, double seed) {
  double a[8];
  uint32_t b = (uint32_t)(seed * 3) + threadIdx.x;
  for (int i = 0; i < ITER; i++) {
#pragma unroll
    for (int j = 0; j < 8; j++) asm volatile("v_cvt_f64_u32 %0, %1\n" : "=v"(a[j]) : "v"(b));
  }
  double s = 0;
  for (int j = 0; j < 8; j++) s += a[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_ldexp(double* out, double seed) {
  double a[8];
  for (int j = 0; j < 8; j++) a[j] = seed + threadIdx.x + j;
  int b = (int)seed & 1;
  for (int i = 0; i < ITER; i++) {
#pragma unroll
    for (int j = 0; j < 8; j++) asm volatile("v_ldexp_f64 %0, %0, %1\n" : "+v"(a[j]) : "v"(b));
  }
  double s = 0;
  for (int j = 0; j < 8; j++) s += a[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_cmp64(double* out, double seed) {
  double a[8];
  for (int j = 0; j < 8; j++) a[j] = seed + threadIdx.x + j;
  double b = seed;
  for (int i = 0; i < ITER; i++) {
#pragma unroll
    for (int j = 0; j < 8; j++) asm volatile("v_cmp_lt_f64 vcc, %0, %1\n" : : "v"(a[j]), "v"(b) : "vcc");
  }
  double s = 0;
  for (int j = 0; j < 8; j++) s += a[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_readlane(double* out, double seed) {
  uint32_t a[8];
  for (int j = 0; j < 8; j++) a[j] = (uint32_t)seed + threadIdx.x + j;
  uint32_t acc = 0;
  for (int i = 0; i < ITER; i++) {
#pragma unroll
    for (int j = 0; j < 8; j++) { uint32_t s; asm volatile("v_readlane_b32 %0, %1, 5\n" : "=s"(s) : "v"(a[j])); acc += s; }
  }
  out[blockIdx.x * 256 + threadIdx.x] = (double)acc;
}
__global__ void __launch_bounds__(256) k_ldsread(double* out, double seed) {
  __shared__ double tab[2048];
  for (int j = threadIdx.x; j < 2048; j += 256) tab[j] = seed + j;
  __syncthreads();
  double acc = 0;
  uint32_t idx = (threadIdx.x * 37u) & 2047u;
  for (int i = 0; i < ITER; i++) {
#pragma unroll
    for (int j = 0; j < 8; j++) { acc += tab[(idx + j * 257u + i) & 2047u]; }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
// dependent chain latency of fma f64 (one wave per SIMD equivalent: a single chain)
__global__ void __launch_bounds__(256) k_fma_chain(double* out, double seed) {
  double a = seed + threadIdx.x, b = seed * 0.5, c = seed * 0.25;
  for (int i = 0; i < ITER; i++) {
    asm volatile(REP8("v_fma_f64 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c));
  }
  out[blockIdx.x * 256 + threadIdx.x] = a;
}

template <typename K>
static void run(const char* name, K kern, double* out, int blocks, double clock_ghz, int wavesPerSimd) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.5);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.5);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_wave = 8.0 * ITER;
  const double cyc = ms * 1e-3 * clock_ghz * 1e9 / (instr_per_wave * wavesPerSimd);
  printf("%-14s blocks %5d  %8.3f ms  %6.2f cycles per wave-instruction per SIMD (at %.2f GHz)\n", name, blocks, ms, cyc, clock_ghz);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const double ghz = p.clockRate * 1e-6;
  const int cus = p.multiProcessorCount;
  printf("%s CUs %d clock %.3f GHz\n", p.gcnArchName, cus, ghz);
  double* out;
  hipMalloc(&out, sizeof(double) * 256 * cus * 8);
  for (int wps : {1, 2, 4}) {  // waves per SIMD: blocks per CU = wps (a 256-thread block = one wave on each SIMD)
    const int blocks = cus * wps;
    printf("--- %d wave(s) per SIMD\n", wps);
#define RUN(k) run(#k, k, out, blocks, ghz, wps)
    RUN(k_fma); RUN(k_fmac); RUN(k_add); RUN(k_mul); RUN(k_max); RUN(k_mov64); RUN(k_rcp); RUN(k_rsq); RUN(k_sqrt); RUN(k_rndne);
    RUN(k_frexpm); RUN(k_ldexp); RUN(k_cmp64); RUN(k_cvt_f64_u32); RUN(k_pkfma32);
    RUN(k_xor); RUN(k_addu); RUN(k_add3); RUN(k_xad); RUN(k_bfe); RUN(k_mullo); RUN(k_mulhi); RUN(k_mul24); RUN(k_mad24); RUN(k_mad64);
    RUN(k_align); RUN(k_perm); RUN(k_cndmask); RUN(k_dpp); RUN(k_cvtf32); RUN(k_readlane); RUN(k_ldsread); RUN(k_fma_chain);
  }
  return 0;
}
