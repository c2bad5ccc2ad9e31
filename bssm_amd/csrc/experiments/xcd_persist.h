// experiments/xcd_persist.h -- the XCD-resident persistent step kernel of round 4: an EXPERIMENT that was built, is bit-identical to
// the launch-per-step kernels, was measured at 0.35-0.57 x their rate (profiles/r04_xcd_persistent.txt) and is NOT part of the
// product.  Included by pf_grid_v2.h only with -DBSSM_XCD_PERSIST (`make -C bssm_amd/csrc persist` -> ab_persist.so,
// scripts/xcd_persist_probe.py); BSSM_AMD_XCD_PERSIST=1 then selects it where a kernel exists.
#pragma once
namespace bssm {
// ------------------------------------------------------------------------------------------------------------------
// XCD-RESIDENT PERSISTENT FILTER (round 4; an EXPERIMENT that was measured and NOT kept -- profiles/r04_xcd_persistent.txt:
// correct and bit-identical to the step kernels, but 0.57x for a single C4 filter and 0.35x at 64 filters in flight; compiled
// only with -DBSSM_XCD_PERSIST, `make persist` -> ab_persist.so, scripts/xcd_persist_probe.py): the whole time loop of a
// filter in ONE launch, its B workgroups confined to
// one XCD and synchronised by a counter in that XCD's L2 instead of by kernel boundaries.
//
// Why an XCD: the per-XCD L2s are not coherent with each other, so a grid-wide barrier needs an agent-scope release +
// acquire per step (4-7 us, MI355X_MICROARCH.md price list) -- more than the 1.5-1.9 us kernel boundary it would replace.
// Inside one XCD the L2 IS the point of coherence: a producer's plain stores are in it once acknowledged (s_waitcnt
// vmcnt(0)), and a consumer on another CU of the same XCD sees them through sc1 loads, which bypass only the CU's own vector
// L1 (v2_step_body<..., COH = true>).  No fences, no L2 write-back, no L1 invalidation per step.
//
// Placement is MEASURED, not assumed: every workgroup reads the id of the XCD it actually runs on (HW_REG_XCC_ID) and takes a
// slot on THAT XCD; B consecutive slots form a team, a team is one filter's B particle blocks.  Teams pull filters from a
// queue until it is empty.  A team whose slots are not all taken once every workgroup of the grid has registered (an uneven
// dispatch) simply leaves; its share of the filters is done by the others.  The host sizes the grid to the occupancy of the
// kernel, so every workgroup is resident from the start; every wait is bounded by a wall-clock timeout that raises the error
// word instead of hanging the device (the host then falls back to one launch per step).
struct PersistCtl { unsigned* w; int total, F, f_base; unsigned long long timeout_ticks; int dbg; /* measurement: bit 1 no step body, bit 2 no waiting at the barrier */ };

__device__ __forceinline__ unsigned pc_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned pc_add(unsigned* p, unsigned v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void pc_store(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// all B workgroups of a team: everything each of them stored before is visible to the sc1 loads the others issue after
__device__ __forceinline__ bool v2_team_barrier(const PersistCtl& ctl, unsigned* bar, const unsigned target, unsigned* s_flag) {
  __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave: its stores have been acknowledged by the L2
  __syncthreads();
  if (threadIdx.x == 0) {
    pc_add(bar, 1u);
    unsigned ok = 1u;
    const unsigned long long t0 = wall_clock64();
    while (!(ctl.dbg & 4) && (int)(pc_load(bar) - target) < 0) {
      __builtin_amdgcn_s_sleep(6);
      if (wall_clock64() - t0 > ctl.timeout_ticks) { pc_store(ctl.w + PC_ERROR, 1u); ok = 0u; break; }
    }
    if (pc_load(ctl.w + PC_ERROR) != 0u) ok = 0u;   // another team gave up: leave too
    *s_flag = ok;
  }
  __syncthreads();
  return *s_flag != 0u;
}

// a parameter struct read from device memory through scalar loads (the pointer is wave-uniform; the words land in SGPRs)
template <class T>
__device__ __forceinline__ T v2_load_params(const T* p) {
  typedef __attribute__((address_space(4))) const char cc;
  T out;
  __builtin_memcpy(&out, (cc*)(uint64_t)p, sizeof(T));
  return out;
}

// One step as a REAL function call (noinline): inlined into the time loop the body's loop invariants -- base addresses, Philox
// keys, ~90 scalar words of descriptors -- are hoisted in front of the loop and overflow the register files (145 spilt SGPRs,
// 60 spilt VGPRs however the inputs are disguised); as a function the step has the register allocation of the launch-per-step
// kernel (2 + 2 spills), and the call costs a few hundred cycles per 10-20 us step.  The descriptors come from device memory
// through scalar loads, the way a kernel reads its arguments.
template <int MP, int KP, int PPT, bool MS, bool WX, bool EXACT>
__device__ __attribute__((noinline)) bool v2_step_call(const FilterBatch* bp, const GridWorkspace* wp, int t, int pb, int f) {
  bp = uniform_ptr(bp);
  wp = uniform_ptr(wp);
  t = __builtin_amdgcn_readfirstlane(t); pb = __builtin_amdgcn_readfirstlane(pb); f = __builtin_amdgcn_readfirstlane(f);
  const FilterBatch b = v2_load_params(bp);
  const GridWorkspace ws = v2_load_params(wp);
  return v2_step_body<MP, KP, PPT, MS, WX, EXACT, true>(b, ws, t, pb, f, false);
}

// bp / wp: the batch and workspace descriptors in DEVICE memory (behind the control block).  As by-value kernel arguments the
// compiler loads all of their ~90 scalar words in front of the time loop and keeps them there (they are loop invariants):
// 145 spilt SGPRs.  Read afresh at the top of every step through a pointer the compiler cannot see through, they live exactly
// as long as in the launch-per-step kernel.
template <int MP, int KP, int PPT, bool MS, bool WX, bool EXACT>
__global__ void __launch_bounds__(GRID_NT, v2_occupancy(MP, PPT, MS)) pf_xcd_persist_v2(const FilterBatch* bp, const GridWorkspace* wp, const PersistCtl ctl, const int B, const int n_steps) {
  __shared__ unsigned s_ctl[8];
  const int tid = threadIdx.x;
  if (tid == 0) {
    unsigned xcc;
    __asm__ volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    const unsigned slot = pc_add(ctl.w + PC_REG + xcc, 1u);
    pc_add(ctl.w + PC_REGISTERED, 1u);
    const unsigned team = slot / (unsigned)B, member = slot % (unsigned)B;
    unsigned ok = 0u;
    if (team < (unsigned)kPersistMaxTeams) {
      const unsigned long long t0 = wall_clock64();
      for (;;) {  // is my team complete?
        if (pc_load(ctl.w + PC_REG + xcc) >= (team + 1u) * (unsigned)B) { ok = 1u; break; }
        if (pc_load(ctl.w + PC_REGISTERED) >= (unsigned)ctl.total) { ok = pc_load(ctl.w + PC_REG + xcc) >= (team + 1u) * (unsigned)B ? 1u : 0u; break; }
        if (wall_clock64() - t0 > ctl.timeout_ticks) break;   // (part of the grid never became resident)
        __builtin_amdgcn_s_sleep(4);
      }
    }
    s_ctl[0] = xcc; s_ctl[1] = team; s_ctl[2] = member; s_ctl[3] = ok;
  }
  __syncthreads();
  if (s_ctl[3] == 0u) return;
  math_tab_init();   // once per launch (the step kernels publish the tables through their first barrier, every launch)
  const unsigned xcc = __builtin_amdgcn_readfirstlane(s_ctl[0]), team = __builtin_amdgcn_readfirstlane(s_ctl[1]);
  const int pb = __builtin_amdgcn_readfirstlane((int)s_ctl[2]);
  unsigned* bar = ctl.w + PC_TEAM + (xcc * kPersistMaxTeams + team) * kPersistLine;
  unsigned* team_f = bar + 16;
  unsigned gen = 0u;
  for (;;) {
    if (pb == 0 && tid == 0) pc_store(team_f, pc_add(ctl.w + PC_HEAD, 1u));   // the team's leader takes the next filter
    if (!v2_team_barrier(ctl, bar, ++gen * (unsigned)B, &s_ctl[4])) return;
    if (tid == 0) s_ctl[5] = pc_load(team_f);
    __syncthreads();
    const unsigned fq = __builtin_amdgcn_readfirstlane(s_ctl[5]);
    if (fq >= (unsigned)ctl.F) return;
    const int f = ctl.f_base + (int)fq;
    for (int t = 0; t < n_steps; t++) {
      const bool alive = (ctl.dbg & 2) ? true : v2_step_call<MP, KP, PPT, MS, WX, EXACT>(bp, wp, t, pb, f);
      if (!v2_team_barrier(ctl, bar, ++gen * (unsigned)B, &s_ctl[4])) return;
      if (!alive) break;   // degenerate weights: every workgroup of the filter takes this exit at the same step
    }
    if (pb == 0 && tid == 0) pc_add(ctl.w + PC_DONE, 1u);
  }
}

template <int MP, int KP, int PPT, bool MS>
inline auto v2_persist_kernel(bool wx, bool exact) -> void (*)(const FilterBatch*, const GridWorkspace*, const PersistCtl, const int, const int) {
  // instantiated where it can pay: complete blocks (EXACT), the 4- and 8-dimensional shapes of the BASELINE configurations
  if constexpr ((MP == 8 && KP == 8 && MS && PPT <= 2) || (MP == 4 && KP == 4 && !MS)) {
    if (exact) return wx ? pf_xcd_persist_v2<MP, KP, PPT, MS, true, true> : pf_xcd_persist_v2<MP, KP, PPT, MS, false, true>;
  }
  return nullptr;
}


// host side: try the persistent form for this batch.  Returns true when the call is finished (status in *status), false when
// the ordinary one-launch-per-step path should run (the persistent kernel is not applicable, or did not complete every filter).
template <class KP_, class KI_>
static bool xcd_persist_try(const FilterBatch& b, const GridWorkspace& ws, const GridWorkspace& ws0, hipStream_t stream, StepTiming* timing,
                            KP_ kern_persist, KI_ kern_init, const int fw, const char* name, hipError_t* status) {
  const dim3 block(GRID_NT);
  hipError_t e;
  auto fail = [&](hipError_t err) { *status = err; return true; };
  if (!(kern_persist && kern_init && opt().xcd_persist >= 1 && (!timing || timing->every <= 0) && ws.persist)) return false;
  int occ = 0, dev = 0;
  hipDeviceProp_t prop;
  if (!(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)kern_persist, GRID_NT, 0) == hipSuccess && occ > 0 &&
        hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount >= 8)) return false;
  const int per_xcd = prop.multiProcessorCount / 8 * occ;
  const int teams_xcd_max = std::min(per_xcd / ws.B, kPersistMaxTeams);
  if (teams_xcd_max < 1) return false;
  const int teams_xcd = std::min(teams_xcd_max, (b.F + 7) / 8);
  const int total = 8 * teams_xcd * ws.B;
  GridWorkspace w0 = ws0;
  w0.f0 = 0;
  const PersistCtl ctl{ws.persist, total, b.F, 0, 20000000ull /* 0.2 s of the 100 MHz wall clock per wait */, opt().xcd_persist_dbg};
  if ((e = hipMemsetAsync(ws.persist, 0, sizeof(unsigned) * kPersistWords, stream)) != hipSuccess) return fail(e);
  // the descriptors the persistent kernel re-reads every step: device memory behind the control words
  static_assert(sizeof(FilterBatch) % 8 == 0 && sizeof(GridWorkspace) % 8 == 0, "descriptors are copied and read as 64-bit words");
  char* dparams = (char*)(ws.persist + kPersistWords);
  if ((e = hipMemcpyAsync(dparams, &b, sizeof(FilterBatch), hipMemcpyHostToDevice, stream)) != hipSuccess) return fail(e);
  if ((e = hipMemcpyAsync(dparams + sizeof(FilterBatch), &w0, sizeof(GridWorkspace), hipMemcpyHostToDevice, stream)) != hipSuccess) return fail(e);
  hipLaunchKernelGGL(kern_init, dim3(ws.B, b.F), block, 0, stream, b, w0);
  hipLaunchKernelGGL(kern_persist, dim3(total), block, 0, stream, (const FilterBatch*)dparams, (const GridWorkspace*)(dparams + sizeof(FilterBatch)), ctl, ws.B, b.n);
  unsigned tail[4] = {0, 0, 0, 0};
  if ((e = hipMemcpyAsync(tail, ws.persist + PC_REGISTERED, sizeof tail, hipMemcpyDeviceToHost, stream)) != hipSuccess) return fail(e);
  if ((e = hipStreamSynchronize(stream)) != hipSuccess) return fail(e);
  const bool complete = tail[PC_ERROR - PC_REGISTERED] == 0u && tail[PC_DONE - PC_REGISTERED] == (unsigned)b.F;
  if (timing) { timing->concurrent = 1; timing->persistent = complete ? teams_xcd * 8 : -1; }
  if (complete) {
    note_filter_kernel(name);
    hipLaunchKernelGGL(loglik_finalize_kernel, dim3(b.F), dim3(64), 0, stream, b.loglik, ws.llinc, b.dead_at, b.n, b.F);
    return fail(hipGetLastError());
  }
  // not every filter was completed: start over, one launch per step (the outputs are rewritten from t = -1)
  if ((e = hipMemsetAsync(b.dead_at, 0, sizeof(int) * b.F, stream)) != hipSuccess) return fail(e);
  if ((e = hipMemsetAsync(ws.llinc, 0, sizeof(double) * b.F * (b.n + 1), stream)) != hipSuccess) return fail(e);
  if (fw && (e = hipMemsetAsync(ws.wscale, 0, sizeof(double) * b.F * (b.n + 1) * ws.B, stream)) != hipSuccess) return fail(e);
  return false;
}

}  // namespace bssm
