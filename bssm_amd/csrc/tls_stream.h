// tls_stream.h -- one cached HIP stream per (host thread, purpose), re-made when the thread's current device changes
// and handed back when the thread exits.  The library keeps a few long-lived streams per host thread (side launch
// sequence, output-copy stream, set-up smoother); a plain `static thread_local hipStream_t` leaked one stream per
// short-lived worker thread and handed a stream of the wrong device to a thread that had switched devices.
// Round 5: a post-correction call runs its filters on a worker thread of its own, and creating / destroying that thread's
// streams cost 0.4 ms of a 7.6 ms call (hipStreamDestroy at thread exit) -- streams now return to a process-wide pool
// (options.cpp) keyed by (device, priority) and are taken from there first; nothing is destroyed before the process ends.
#pragma once
#include <hip/hip_runtime.h>
#include <sys/syscall.h>
#include <unistd.h>

namespace bssm {

#ifndef BSSM_USER_MODEL
hipStream_t stream_pool_take(int dev, int kind);  // nullptr when the pool has none; kind: 0 plain, 1 highest priority, 2 plain, reserved (created early)
void stream_pool_give(int dev, int kind, hipStream_t s);
bool tail_overlap_available();          // false when no stream of the current device was MEASURED to run beside its null stream (options.cpp)
int tail_overlap_state(int* candidates_tried);  // 0 not calibrated, 1 measured yes, 2 measured no
void prewarm_streams();  // once per DEVICE: at the first entry-point call with that device current, and in every shard thread (options.cpp)
#endif

struct TlsStream {
  hipStream_t s = nullptr;
  int dev = -1;
  int hi = 0;   // kind: 0 plain, 1 highest priority, 2 plain, reserved (created early)
  // highest_priority: for small latency-critical kernels that compete with grids filling every CU
  // reserved: a stream of the kind the library creates once, early, for work that must run BESIDE the null stream's (the tail of a
  // post-correction chunk beside its last round; prewarm_streams() in options.cpp says why) -- a plain stream from a pool of its own
  hipStream_t get(bool highest_priority = false, bool reserved = false) {
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) return nullptr;
    if (s && cur == dev) return s;
    drop();
    hi = highest_priority ? 1 : (reserved ? 2 : 0);
#ifndef BSSM_USER_MODEL
    if ((s = stream_pool_take(cur, hi)) != nullptr) { dev = cur; return s; }
#endif
    hipError_t e;
    if (highest_priority) {
      int lo_p = 0, hi_p = 0;
      (void)hipDeviceGetStreamPriorityRange(&lo_p, &hi_p);
      e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi_p);
    } else {
      e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    }
    if (e != hipSuccess) { s = nullptr; (void)hipGetLastError(); return nullptr; }
    dev = cur;
    return s;
  }
  void drop() {
    if (!s) return;
#ifndef BSSM_USER_MODEL
    stream_pool_give(dev, hi, s);
    s = nullptr;
    return;
#endif
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (cur != dev) (void)hipSetDevice(dev);
    (void)hipStreamDestroy(s);
    if (cur != dev) (void)hipSetDevice(cur);
    s = nullptr;
  }
  ~TlsStream() {
    // the main thread's thread-locals die during process exit, possibly after the HIP runtime: leave its streams alone
    if ((long)syscall(SYS_gettid) == (long)getpid()) return;
    drop();
  }
};

}  // namespace bssm
