// tls_stream.h -- one cached HIP stream per (host thread, purpose), re-made when the thread's current device changes
// and destroyed when the thread exits.  The library keeps a few long-lived streams per host thread (side launch
// sequence, output-copy stream, set-up smoother); a plain `static thread_local hipStream_t` leaked one stream per
// short-lived worker thread and handed a stream of the wrong device to a thread that had switched devices.
#pragma once
#include <hip/hip_runtime.h>
#include <sys/syscall.h>
#include <unistd.h>

namespace bssm {

struct TlsStream {
  hipStream_t s = nullptr;
  int dev = -1;
  // highest_priority: for small latency-critical kernels that compete with grids filling every CU
  hipStream_t get(bool highest_priority = false) {
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) return nullptr;
    if (s && cur == dev) return s;
    drop();
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
