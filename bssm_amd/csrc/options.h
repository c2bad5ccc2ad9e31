// options.h -- the library's environment switches, read ONCE (at load, and again only on bssm_amd_reload_env()).
//
// Three kinds:
//   configuration   BSSM_AMD_CACHE_GB, BSSM_AMD_COPY_THREADS, BSSM_AMD_USER_CACHE, BSSM_AMD_IS_CHUNK, BSSM_AMD_GATHER,
//                   BSSM_AMD_TRACE, BSSM_AMD_TAIL_SPLIT (=0: off; default ON since round 5 -- a sparsely filled last round of a
//                   post-correction chunk through the row family, bit-identical to the workgroup kernel), BSSM_AMD_TAIL_CONCURRENT (=0:
//                   that tail behind the last full round instead of beside it), BSSM_AMD_TAIL_MAX_EIGHTHS, BSSM_AMD_USER_BUILD_JOBS
//   test / tuning   kernel-shape and set-up choices the parity tests and the tuning scripts flip between calls:
//                   BSSM_AMD_FORCE_PATH, BSSM_AMD_GRID_PPT, BSSM_AMD_GRID_V2_WX, BSSM_AMD_GRID_V2_EXACT,
//                   BSSM_AMD_GRID_STREAMS, BSSM_AMD_XCD_REMAP, BSSM_AMD_WG_WIDE, BSSM_AMD_DEVICE_SETUP, BSSM_AMD_GENERIC_SETUP,
//                   BSSM_AMD_MV_SEQUENTIAL, BSSM_AMD_MNG_LANE_PER_DRAW, BSSM_AMD_DEFER_SUMMARY, BSSM_AMD_SPARSE (0 | all), BSSM_AMD_NO_SPARSE,
//                   BSSM_AMD_SETUP_INTERLEAVED, BSSM_AMD_FACTORS_FULL, BSSM_AMD_SETUP_ROWS16 (round 6: comparison switches of the big-state set-up),
//                   BSSM_AMD_HOST_FACTOR_THREADS, BSSM_AMD_SINGLE_DEVICE_SETUP (=1: the psi set-up of ONE model with 9 ... 64 states does its smoother and conditional factors on the device)
//   probes          measurement-only switches that change what a production entry point does (skip an output, bypass the output
//                   pipe, ...): compiled in only with -DBSSM_PROBES (`make probes` -> ../ab_probes.so):
//                   BSSM_AMD_PROBE_NO_WEIGHTS, BSSM_AMD_NO_OUTPIPE, BSSM_AMD_IS_BIG_CHUNK, BSSM_AMD_SIDE_PRIO, BSSM_AMD_SS_GLOBAL,
//                   BSSM_AMD_WG_NO_EXACT
// A process that changes one of them after the library was loaded calls bssm_amd_reload_env() (the Python mirror does so by
// itself when it sees a changed BSSM_AMD_* variable: bssm_amd/_lib.py).
#pragma once
#include <string>

namespace bssm {

struct Options {
  // configuration
  double cache_gb = -1.0;       // < 0: default
  int copy_threads = 0;         // 0: default
  std::string user_cache;       // empty: default
  int is_chunk = 0;             // 0: default
  int gather = 0;               // 0 auto (RCCL -> device copies -> host), 1 "copy", 2 "host"
  int trace = 0;                // BSSM_AMD_TRACE: 1 = timing lines on stderr, 2 = also synchronise after each set-up kernel to time it alone
  bool tail_split = true;       // a sparsely filled last round of a post-correction chunk through the row family (bit-identical: pf_canon.h)
  // test / tuning
  int tail_max_eighths = 4;     // the last round goes through the row family when it fills at most this many eighths of the CUs (BSSM_AMD_TAIL_MAX_EIGHTHS)
  bool prewarm_streams = true;  // the library's streams are brought into being in a fixed order at the first entry-point call (options.cpp: prewarm_streams; BSSM_AMD_PREWARM_STREAMS=0: on demand)
  bool tail_concurrent = true;  // that tail BESIDE the last full round instead of behind it (BSSM_AMD_TAIL_CONCURRENT=0: behind)
  int force_path = 0;           // 0 none, 1 "wg", 2 "grid", 3 "row" (the row family where the batch has such kernels; test hook)
  int grid_ppt = 0;             // 0: automatic
  int grid_v2_wx = -1, grid_v2_exact = -1;   // -1: automatic; 0 / 1
  int grid_streams = 0;         // 0: automatic
  int xcd_remap = -1;           // -1: default
  int wg_wide = -1;             // -1: automatic
  int device_setup = -1;        // -1: default; 0 host, 1 smoother on the device, 2 everything on the device
  bool generic_setup = false, mv_sequential = false, mng_lane_per_draw = false, defer_summary = false;
  int sparse = 1;               // structured transitions (pf_big.h): 0 never, 1 where they measured faster, 2 wherever a descriptor exists (BSSM_AMD_SPARSE=0|all; BSSM_AMD_NO_SPARSE=1 = 0)
  bool setup_interleaved = false;   // set-up of 32 / 64 padded states: draw-interleaved matrix work arrays (the layout before round 6) instead of draw-major
  int host_factor_threads = 0;      // host threads of the conditional factors' time points (host_model.cpp: conditional_cov): 0 automatic (up to 32), 1 the serial loop (BSSM_AMD_HOST_FACTOR_THREADS)
  bool single_device_setup = false; // psi set-up of one model (single psi calls, pm / da chains): smoother + factors of 9 ... 64 states on the device (opt-in: api_postcorrect.cpp: device_factors_one)
  bool setup_rows16 = false;        // set-up of 9 ... 16 states: the smoother with its rows in registers (lg_smoother_rows.h, before late round 6) instead of the matrix-core form
  bool factors_full = false;        // conditional factors of 16 and more padded states: the full (Jacobi) decomposition at EVERY time point -- the thin kernel is not launched (comparison)
  int xcd_persist = -1;         // -1: automatic; 0 never, 1 whenever eligible (the XCD-resident persistent step kernel)
  int xcd_persist_dbg = 0;      // measurement bits of the persistent kernel (BSSM_AMD_XCD_PERSIST_DBG; -DBSSM_PROBES builds)
  // probes (constant defaults unless built with -DBSSM_PROBES)
  bool probe_no_weights = false, no_outpipe = false, side_prio = false, ss_global = false, wg_no_exact = false;
  int is_big_chunk = 0;
};

#ifdef BSSM_USER_MODEL
// a run-time compiled user-model library (user_model_tu.hip) is self-contained: it reads the main library's snapshot through a
// getter the main library hands over when it loads the library (bssm_user_set_options); defaults until then
typedef const Options& (*options_getter_t)();
inline options_getter_t& user_options_getter() { static options_getter_t g = nullptr; return g; }
inline const Options& opt() {
  static const Options dflt;
  const options_getter_t g = user_options_getter();
  return g ? g() : dflt;
}
#else
const Options& opt();     // the current snapshot (lock-free read)
void reload_options();    // re-read the environment
#endif

}  // namespace bssm
