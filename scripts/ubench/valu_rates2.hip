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
#define U_CND32(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define U_CND64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n"
#define U_CNDNEW(i) "v_cndmask_b32 %" #i ", %8, %9, vcc\n"
#define U_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define U_CMPCND(i) "v_cmp_lt_u32 vcc, %" #i ", %8\nv_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define U_CMPONLY(i) "v_cmp_lt_u32 vcc, %" #i ", %8\n"
#define U_CMPS(i) "v_cmp_lt_u32 s[20:21], %" #i ", %8\nv_cndmask_b32_e64 %" #i ", %" #i ", %9, s[20:21]\n"
#define U_MAXU(i) "v_max_u32 %" #i ", %" #i ", %8\n"
#define U_MINI(i) "v_min_i32 %" #i ", %" #i ", %8\n"
#define U_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 3, %8\n"
#define U_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define U_LSHR(i) "v_lshrrev_b32 %" #i ", 3, %" #i "\n"
#define U_SUB(i) "v_sub_u32 %" #i ", %" #i ", %8\n"
#define U_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
KERNEL_U32(k_cnd32, U_CND32)
KERNEL_U32(k_cnd64, U_CND64)
KERNEL_U32(k_cndnew, U_CNDNEW)
KERNEL_U32(k_mov32, U_MOV)
KERNEL_U32(k_cmpcnd, U_CMPCND)
KERNEL_U32(k_cmponly, U_CMPONLY)
KERNEL_U32(k_cmps, U_CMPS)
KERNEL_U32(k_maxu, U_MAXU)
KERNEL_U32(k_mini, U_MINI)
KERNEL_U32(k_lshladd, U_LSHLADD)
KERNEL_U32(k_andor, U_ANDOR)
KERNEL_U32(k_lshr, U_LSHR)
KERNEL_U32(k_sub, U_SUB)
KERNEL_U32(k_and, U_AND)
#define A_CND64PAIR(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
__global__ void __launch_bounds__(256) k_lds128(double* out, double seed) {
  __shared__ __attribute__((aligned(16))) double tab[2048];
  for (int j = threadIdx.x; j < 2048; j += 256) tab[j] = seed + j;
  __syncthreads();
  double acc = 0;
  uint32_t idx = (threadIdx.x * 37u) & 1023u;
  typedef double d2 __attribute__((ext_vector_type(2)));
  for (int i = 0; i < ITER; i++) {
#pragma unroll
    for (int j = 0; j < 8; j++) { d2 v = *(const d2*)&tab[2 * ((idx + j * 257u + i * 13u) & 1023u)]; acc += v.x + v.y; }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) k_lds64seq(double* out, double seed) {
  __shared__ double tab[2048];
  for (int j = threadIdx.x; j < 2048; j += 256) tab[j] = seed + j;
  __syncthreads();
  double acc = 0;
  for (int i = 0; i < ITER; i++) {
#pragma unroll
    for (int j = 0; j < 8; j++) { acc += tab[(threadIdx.x + j * 256u + i) & 2047u]; }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
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
    RUN(k_cnd32); RUN(k_cnd64); RUN(k_cndnew); RUN(k_mov32); RUN(k_cmpcnd); RUN(k_cmponly); RUN(k_cmps); RUN(k_maxu); RUN(k_mini);
    RUN(k_lshladd); RUN(k_andor); RUN(k_lshr); RUN(k_sub); RUN(k_and); RUN(k_lds128); RUN(k_lds64seq);
  }
  return 0;
}
