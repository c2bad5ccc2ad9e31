// pf_kernels.h -- launch interface between the host library and the HIP kernels.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#endif

namespace bssm {

// layout of FilterBatch::sp, the descriptor of a structured transition (pf_big.h: sparse_apply; api_internal.h: sparse_descriptor)
enum { SP_SLOTS = 8, SP_HDR = 4, SP_SLOT_WORDS = 6, SP_WORDS = SP_HDR + SP_SLOTS * SP_SLOT_WORDS };

// Device-resident description of a batch of F independent filters that share
// (n, m, k, N, distribution) but may differ in every matrix / observation row
// (one per retained theta-draw in post-correction, src/approx_mcmc.cpp:370-417).
// All strides are in doubles.  A stride of 0 shares one table among filters.
struct FilterBatch {
  int n;         // time points
  int m, k;      // real state / disturbance dimensions
  int p;         // observation series per time point (1 = univariate)
  int MP, KP;    // padded dimensions the kernels are instantiated for
  int N;         // particles per filter
  int F;         // filters in the batch
  int psi;       // 0 bootstrap weights, 1 psi-APF weights
  uint32_t seed;
  uint32_t stream0;  // filter f uses Philox key (seed, stream0 + f)
  // [f][tt][MP*MP + MP*KP + MP] : T_tt, R_tt, C_tt (column-major, zero padded)
  const double* mats; size_t mats_fstride; int mats_tstride;
  // [f][t][ROW_HDR + p * (SER_HDR + MP)] for t in [0, n)
  const double* obs; size_t obs_fstride;
  // [f][MP + MP*MP] : a1 then L (initial mean and lower factor)
  const double* init; size_t init_fstride;
  // built-in non-linear / SDE functor parameters (pf_functors.h), [f][FP_SIZE]; null for the linear-state models
  const double* fpar; size_t fpar_fstride;
  // psi-APF of ssm_nlg: approximating T_t (MP x MP), C_t (MP) per time point, [f][n][MP*MP + MP]
  const double* amats; size_t amats_fstride;
  // outputs (any may be null)
  double* alpha_hist;   // [f][n+1][m][N]   particle-minor
  uint32_t* indices;    // [f][n][N]
  double* weights;      // [f][n+1][N]      exp(w - max_t), ones where unobserved
  double* loglik;       // [f]              sum_t (max_t + log mean exp), no constants
  int* dead_at;         // [f] 0 = alive, else (time whose weights were degenerate) + 1; zeroed by the launcher
  int user_slot;        // 0, or the registry slot of a run-time compiled user model whose library holds the kernels
  int fn_id;            // functor models: the model id of the parameter blocks (all filters of a batch share it)
  int lean;             // host only: the workgroup-resident kernel in its 96-register form where one exists (a last round with a tail beside it)
  // Round 6: structured transition of the bootstrap filter with 17 ... 64 states (pf_big.h: sparse_apply).  Null = the dense matrix-core
  // transition.  Otherwise a descriptor of T's pattern (all entries +-1), shared by the filters of the batch: the block of rows that copy
  // the component in front of them, and <= 8 other rows as column masks (pf_big.h: SP_WORDS ints).  The VALUES of R, C, Z still come
  // from every filter's own tables.
  const int* sp;
  int sp_cnz;           // some C_t entry is non-zero (direct rows then add theirs)
};

// Largest N the workgroup-resident kernel can hold for a given padded state size
#ifndef __HIPCC_RTC__
int wg_filter_capacity(int MP);
// One workgroup per filter, whole time loop inside the kernel.
hipError_t launch_wg_filter(const FilterBatch& b, hipStream_t stream);

#endif
// Multi-workgroup filter for large N: workspace + per-time-step launches.
// control block of the XCD-resident persistent step kernel (pf_grid_v2.h): word offsets
constexpr int kPersistMaxTeams = 64;   // teams per XCD
constexpr int kPersistLine = 32;       // words per 128-byte line: every team's barrier counter and mailbox sit in a line of their own
enum { PC_REG = 0 /* [8] workgroups registered per XCD */, PC_REGISTERED = 8, PC_HEAD = 9 /* next filter */, PC_ERROR = 10,
       PC_DONE = 11 /* filters completed */, PC_TEAM = 32 /* [8][kPersistMaxTeams] lines: word 0 barrier counter, word 16 the team's filter */,
       kPersistWords = 32 + 8 * kPersistMaxTeams * kPersistLine };
struct GridWorkspace {
  double* state[2];    // [F][m][N] ping-pong when no history is kept
  double* lcdf;        // [2][F][N] block-local inclusive scans of exp(w - max_b)
  double* coarse;      // [2][F][B][chunk/16] last entry of every row of 16 of lcdf
  double* partials;    // [2][F][B][2]  (max_b, sum_b)
  double* logw;        // [F][N] raw log-weights when weights output is wanted
  double* maxhist;     // [F][n+1] spare (was: global max per time point for a separate weights finalisation pass)
  double* llinc;       // [F][n+1] log-likelihood increment per time point (summed in time order at the end)
  long long* dbg;      // optional [n][16] cycle stamps of block 0 / thread 0 (debug builds of the step kernel)
  int B;               // particle blocks per filter
  int chunk;           // particles per block
  int f0;              // first filter of this launch (a batch may be stepped as two interleaved launch sequences)
  int xcd_remap;       // 1: workgroup ids are remapped so that every XCD owns a contiguous range of (filter, block) pairs
  int wx;              // 1: the exchange array `lcdf` carries block-scaled weights, not their scan (round-2 step kernel, full output)
  // Factored weights output (round 3, opt-in: BSSM_PLAN_FACTORED_WEIGHTS).  fw = 1 on entry asks for it; the launcher
  // grants it where the weights-exchange step kernel runs (and reports it in StepTiming::factored).  The weights output
  // then IS the exchange array: column t of `weights` holds exp(w_i - max of i's particle block), written once by the launch
  // that computes the weights, read by the next launch's search windows, never rewritten; the factor that takes a block to
  // the reference's scale, exp(max_block - max_t), is stored per (filter, time point, block) in `wscale`
  //   reference weight (t, i) = weights[f][t][i] * wscale[f][t][i / chunk]        (t < n; column n is final as stored)
  // -- 16 of the 104 bytes a particle-step moved disappear (the owner's read of its exchanged weights and the second write).
  int fw;
  double* wscale;      // [F][n+1][B]
  unsigned* persist;   // [kPersistWords] control block of the XCD-resident persistent kernel (zeroed before every launch)
#ifdef __HIPCC_RTC__
  void* side;          // (device view of the struct: same size and alignment)
#else
  hipStream_t side;    // the library's side stream for the second launch sequence (one per host thread); null = none
#endif
};
#ifndef __HIPCC_RTC__   // everything below is the host's launch interface
size_t grid_workspace_bytes(const FilterBatch& b, int* B_out, int* chunk_out);
void grid_workspace_carve(const FilterBatch& b, void* base, GridWorkspace* ws);
// The ROW family (pf_row_step.h): a filter of at most 8192 particles spread over its rows of 64, one launch per time step,
// bit-identical to the workgroup-resident kernel (pf_canon.h).  row_filter_supported: the batch's shape has such kernels (no
// weights output; functor models, and linear-state models of at most four states).  The workspace is a GridWorkspace with
// chunk = 64; launch_grid_filter runs it.
bool row_filter_supported(const FilterBatch& b);
// whether launch_wg_filter has a 96-register form (pf_wg_kernel_lean, FilterBatch::lean) for this batch: one- and two-dimensional states at four
// particles per lane of a 1024-thread workgroup (2048 < N <= 4096); not for user models compiled in process
bool wg_lean_form_exists(const FilterBatch& b);
// States of more than 16 dimensions (pf_big.h): both shapes give the same bits, so the choice is throughput alone -- a batch that fills
// the chip with rows runs faster in the row shape (two to four waves per SIMD) than workgroup-resident (one 131 KB workgroup per CU)
bool prefer_step_shape(const FilterBatch& b, bool use_grid);
size_t row_workspace_bytes(const FilterBatch& b);
void row_workspace_carve(const FilterBatch& b, void* base, GridWorkspace* ws);
// Optional per-launch timing of the step kernel: every `every`-th step launch is bracketed by HIP
// events on `stream`; the caller reads total_ms / count after the stream has drained.
// `progress_every` > 0 (single-sequence launches only): an event is recorded on the launch stream after every that many
// time steps (progress_steps[k] = time points complete when progress_events[k] fires), so that the caller can stream
// finished time slabs of the outputs to the host while the filter is still stepping.
struct StepTiming {
  int every = 0; int count = 0; double total_ms = 0.0; int concurrent = 1;  // concurrent: launch sequences in flight
  int factored = 0;  // 1: the weights output of this run is in factored form (GridWorkspace::fw was granted)
  int persistent = 0;  // > 0: teams of the XCD-resident persistent kernel that ran the batch; -1: it was tried and left for the step kernels
  int progress_every = 0, progress_count = 0;
  static constexpr int kMaxProgress = 64;
  hipEvent_t progress_events[kMaxProgress] = {};
  int progress_steps[kMaxProgress] = {};
};
hipError_t launch_grid_filter(const FilterBatch& b, const GridWorkspace& ws, hipStream_t stream,
                              StepTiming* timing = nullptr);
// Name (with template arguments) of the filter kernel variant the calling thread launched last, as the launcher picked
// it -- benchmarks report this instead of a format string of their own.
void family_launches(unsigned long long (&out)[3]);  // filter batches launched so far by family: workgroup-resident, step kernels, row family (process-wide)
void note_filter_kernel(const char* name);
const char* last_filter_kernel();
// launchers a run-time compiled user-model library exports (user_model_tu.hip); register returns the slot (>= 1)
typedef hipError_t (*user_wg_launcher_t)(const FilterBatch*, hipStream_t);
typedef hipError_t (*user_grid_launcher_t)(const FilterBatch*, const GridWorkspace*, hipStream_t, StepTiming*);
int register_user_launchers(user_wg_launcher_t wg, user_grid_launcher_t grid);
// a slot whose only step shape is the row family (models compiled in process: api_filters.cpp bssm_user_model_jit)
void mark_user_slot_rows_only(int slot);
bool user_slot_rows_only(int slot);
// log-likelihoods of a row-family run from its per-step increments (for launch loops outside pf_row_step.h)
hipError_t launch_loglik_finalize(const FilterBatch& b, const GridWorkspace& ws, hipStream_t stream);

// Ancestor back-trace (src/filter_smoother.cpp:5-18) on the particle-minor history, in place is not
// possible in parallel, so it writes `out` [f][n+1][m][N].
hipError_t launch_backtrace(const double* hist, const uint32_t* indices, double* out, int F, int n, int m, int N,
                            hipStream_t stream);
// summary() (src/summary.cpp:4-17): mean [f][n+1][m] and cov [f][n+1][m][m] over particles
hipError_t launch_summary(const double* hist, double* mean, double* cov, int F, int n1, int m, int N,
                          hipStream_t stream);
// one back-traced trajectory per filter (particle picks[f]): out [f][n+1][m]
hipError_t launch_pick_trajectory(const double* hist, const uint32_t* indices, const uint32_t* picks, double* out, int F,
                                  int n, int m, int N, hipStream_t stream);
// filter_smoother + summary fused (m <= 8): mean [f][n+1][m], cov [f][n+1][m][m]; bcur [F][N] scratch
hipError_t launch_smooth_summary(const double* hist, const uint32_t* indices, uint32_t* bcur, double* mean, double* cov, int F,
                                 int n, int m, int N, hipStream_t stream);
// filter_summary() (src/summary.cpp:42-79): at [f][n+1][m], att [f][n][m], Pt [f][n+1][m*m], Ptt [f][n][m*m]
hipError_t launch_filter_summary(const double* hist, const double* weights, double* at, double* att, double* Pt,
                                 double* Ptt, int F, int n, int m, int N, hipStream_t stream);
// [f][t][c][i] -> Armadillo cube m x (n+1) x N (index c + m*(t + (n+1)*i)) per filter
hipError_t launch_to_cube(const double* hist, double* cube, int F, int n1, int m, int N, hipStream_t stream);
// [f][t][i] u32 -> N x n column-major is already [t][i]; weights likewise. (no kernel needed)

// zero the output slots a degenerate filter never wrote (the reference hands back zero-filled buffers,
// src/R_bootstrap_filter.cpp:136-138); a no-op for healthy filters, so outputs need no memset per run
hipError_t launch_zero_dead_tail(const FilterBatch& b, bool weights_too, hipStream_t stream, double* wscale = nullptr, int B = 0);  // wscale: the factors of a factored weights output [F][n+1][B]
// exact simulation smoother of a linear-Gaussian model; mats [n][2 m m + m] unpadded, init [m + m m], hist [n+1][m][N]
hipError_t launch_sim_smoother(const double* mats, const double* init, double* hist, int n, int m, int N, uint32_t seed,
                               hipStream_t stream);
// stratified_sample (src/stratified_sample.cpp:9-28): p [len] is overwritten by its cumulative sum, out [N]
hipError_t launch_stratified_sample(double* p, const double* r, uint32_t* out, int len, int N, hipStream_t stream);
// psi-APF set-up smoother of F draws as one batch, one lane per draw (lg_smoother.h).  Every array is
// DRAW-INTERLEAVED: element k of draw d at base[k * F + d] (a wave's 64 lanes then touch consecutive doubles).
// `in` holds the F packed models (LgPacked element order, in_stride elements per draw); outputs alphahat [(n+1) m][F],
// Vt / Ct [(n+1) m m][F]; scratch K [n m][F], v / Fs [n][F]; ok [F] = 1 when every result is finite.
// launch_transpose converts between per-draw contiguous [rows][cols] and interleaved [cols][rows].
struct LgBatchDev {
  int F, n, m, sub;   // sub: series per time point treated one at a time (LgModel::sub), 1 = univariate observations
  int Zs, Ts, Rs, Ds, Cs, Hs;
  size_t oZ, oT, oR, oD, oC, oy, oH, oa, oP, in_stride;
  const double* in;
  double *at, *Pt, *cc, *K, *v, *Fs;
  int* ok;
};
hipError_t launch_lg_setup(const LgBatchDev& b, hipStream_t stream);
// psi-APF set-up of ssm_nlg draws entirely on the device: linearisation at the stored mode
// (approximate_for_is, src/model_ssm_nlg.cpp:146-176), Kalman / covariance smoother + conditional factors
// (lg_smoother.h), and the filter's tables written straight into the batch layout -- one lane per draw.
struct FnSetupDev {
  int F, n, m;
  const double* fpar;   // [F][FP_SIZE]
  const double* ext;    // [F][m + m*m]: a1, P1 of the draw
  const double* modes;  // [F][m*n], m x n column-major per draw
  const double* y;      // [n], shared by the draws
  double* scratch;      // work arrays in blocks of 64 draws (FnWork), fn_setup_scratch_doubles(n, m) * ceil64(F) doubles
  double *mats, *obs, *init, *amats;  // per-draw contiguous outputs: [F][n][2 m^2 + m], [F][n][ROW_HDR + SER_HDR + m], [F][m + m^2], [F][n][m^2 + m]
  int* ok;              // [F]: 0 = the smoother failed (tables are then neutral and the caller reports -Inf)
};
size_t fn_setup_scratch_doubles(int n, int m);
hipError_t launch_fn_setup(const FnSetupDev& b, hipStream_t stream);
// psi-APF set-up of ssm_ung draws (univariate observations) entirely on the device: Laplace step at the stored mode and
// scales (approximate_for_is, src/model_ssm_ung.cpp:143-161, 241-292, 302-342), smoother + conditional factors, and the
// filter's tables (what build_psi_tables / fill_obs_rows write on the host) -- one lane per draw.  `in` holds the packed
// draws (ung_pack element order) DRAW-INTERLEAVED: element k of draw d at in[k * F + d].
struct UngSetupDev {
  int F, n, m;
  int p = 1;                 // series per time point (ssm_mng: mng_setup_kernel, mng_pack element order; obs rows ROW_HDR + p (SER_HDR + MP))
  int mr = 0;                // the states the draws really have where m is a padded dimension (16 / 32 / 64; 0: m) -- the rest is exact zeros
  int dm = 0;                // round 6, m = 32 / 64: the MATRIX work arrays (at, Pt, cc, K, Vf, Cf) are draw-major (a draw's slice contiguous) instead of draw-interleaved
  int Zs, Ts, Rs, Ds, Cs;
  size_t oZ, oT, oR, oD, oC, oy, oa, oP, ou, ox, om, oh;
  const double* in;
  double* scratch;           // draw-interleaved work arrays, ung_setup_scratch_doubles(n, m) * F doubles
  double *mats, *obs, *init; // per-draw contiguous outputs, padded to MP = pad(m): [F][n][2 MP^2 + MP], [F][n][ROW_HDR + SER_HDR + MP], [F][MP + MP^2]
  int* ok;
  int* flags;                // round 6, 16 and more padded states: [F][n + 1] which (draw, time point) the full factor kernel has to do (written by the thin one)
  double* cond = nullptr;    // (optional) [F][n + 1]: smallest / first pivot of the thin path's factor of V_{t-1} per (draw, time point >= 1) -- psi_setup of one model checks it
  int phases = 15;           // which parts launch_ung_setup runs: 1 Laplace step at the stored mode, 2 smoother, 4 conditional factors, 8 tables -- the psi set-up of ONE
                             // model (psi_setup: host mode search, then smoother + factors here on the caller's approximating y / H in the work arrays) asks for 2 | 4
  int t_off = 0;             // the full factor kernels: time point of grid row 0 (the t = 0 decomposition runs as a launch of its own beside the thin kernel)
};
size_t ung_setup_scratch_doubles(int n, int m);
hipError_t launch_ung_setup(const UngSetupDev& b, hipStream_t stream);
hipError_t debug_thin_counts(unsigned long long out[2]);
size_t mng_setup_scratch_doubles(int n, int m, int p);
hipError_t launch_mng_setup(const UngSetupDev& b, hipStream_t stream);
hipError_t launch_transpose(const double* in, double* out, int rows, int cols, hipStream_t stream);
hipError_t launch_calib_copy(const double* in, double* out, size_t n, hipStream_t stream);
hipError_t launch_debug_math(int fn, const double* in, double* out, int n, hipStream_t stream);

#endif  // !__HIPCC_RTC__

}  // namespace bssm
