// rccl_gather.cpp -- see rccl_gather.h.  RCCL is reached through dlopen + dlsym only.
#include "rccl_gather.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace bssm {
namespace {

struct Api {
  void* handle = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string why;
  bool ok = false;
};

Api& api() {
  static Api* a = [] {
    Api* x = new Api;
    // The RCCL that belongs to the HIP runtime this library is bound to: a process may hold two ROCm stacks (PyTorch
    // ships its own libamdhip64 / libhsa-runtime64 / librccl next to /opt/rocm's; whichever libamdhip64 was loaded first
    // serves everybody by soname).  An RCCL from the other stack opens its own, uninitialised HSA runtime and reports
    // "no ROCm-capable device".  So: look where the loaded libamdhip64 lives and take the librccl beside it; DEEPBIND keeps
    // a second RCCL's internal calls inside itself when the other stack's copy is already in the global namespace.
    std::vector<std::string> names;
    Dl_info di;
    if (dladdr((void*)&hipGetDeviceCount, &di) && di.dli_fname) {
      std::string dir(di.dli_fname);
      const size_t slash = dir.rfind('/');
      if (slash != std::string::npos) {
        dir.resize(slash + 1);
        names.push_back(dir + "librccl.so.1");
        names.push_back(dir + "librccl.so");
      }
    }
    names.push_back("/opt/rocm/lib/librccl.so.1");
    names.push_back("librccl.so.1");
    for (const std::string& nm : names) {
      x->handle = dlopen(nm.c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
      if (x->handle) break;
    }
    if (!x->handle) {
      const char* e = dlerror();
      x->why = std::string("librccl.so could not be opened: ") + (e ? e : "unknown");
      return x;
    }
    auto sym = [&](const char* nm) -> void* {
      void* p = dlsym(x->handle, nm);
      if (!p && x->why.empty()) x->why = std::string("librccl: missing symbol ") + nm;
      return p;
    };
    x->CommInitAll = (decltype(x->CommInitAll))sym("ncclCommInitAll");
    x->CommDestroy = (decltype(x->CommDestroy))sym("ncclCommDestroy");
    x->AllGather = (decltype(x->AllGather))sym("ncclAllGather");
    x->GroupStart = (decltype(x->GroupStart))sym("ncclGroupStart");
    x->GroupEnd = (decltype(x->GroupEnd))sym("ncclGroupEnd");
    x->GetErrorString = (decltype(x->GetErrorString))sym("ncclGetErrorString");
    x->ok = x->why.empty();
    return x;
  }();
  return *a;
}

std::mutex g_mu;
std::map<std::vector<int>, std::vector<ncclComm_t>>& comm_cache() {
  static auto* m = new std::map<std::vector<int>, std::vector<ncclComm_t>>;
  return *m;
}
// Communicators are destroyed by an atexit handler registered when the first one is made, i.e. after librccl was
// loaded, so it runs BEFORE librccl's own exit-time destructors (leaving live communicators to them ended the process
// with "double free or corruption").
void destroy_comms_at_exit() {
  Api& a = api();
  if (!a.ok) return;
  for (auto& kv : comm_cache())
    for (ncclComm_t c : kv.second) if (c) (void)a.CommDestroy(c);
  comm_cache().clear();
}

}  // namespace

bool rccl_available(std::string* why) {
  Api& a = api();
  if (!a.ok && why) *why = a.why;
  return a.ok;
}

int rccl_all_gather(const std::vector<int>& devices, const std::vector<const double*>& send, const std::vector<double*>& recv,
                    size_t count, const std::vector<hipStream_t>& streams, std::string* err) {
  Api& a = api();
  if (!a.ok) { *err = a.why; return -1; }
  const int nd = (int)devices.size();
  if (nd == 0 || (int)send.size() != nd || (int)recv.size() != nd || (int)streams.size() != nd) { *err = "rccl_all_gather: argument sizes"; return -1; }
  int dev0 = 0;
  (void)hipGetDevice(&dev0);
  std::lock_guard<std::mutex> lk(g_mu);  // one collective at a time per process: the communicators are shared
  auto& cache = comm_cache();
  auto it = cache.find(devices);
  // RCCL checks hipGetLastError() after its own launches: an error code an earlier, unrelated call left behind on one of
  // these devices would be reported as "unhandled cuda error"
  for (int d = 0; d < nd; d++) { (void)hipSetDevice(devices[d]); (void)hipGetLastError(); }
  if (it == cache.end()) {
    static bool registered = false;
    if (!registered) { std::atexit(destroy_comms_at_exit); registered = true; }
    std::vector<ncclComm_t> comms(nd, nullptr);
    const ncclResult_t r = a.CommInitAll(comms.data(), nd, devices.data());
    if (r != ncclSuccess) { *err = std::string("ncclCommInitAll: ") + a.GetErrorString(r); (void)hipSetDevice(dev0); return -1; }
    it = cache.emplace(devices, std::move(comms)).first;
  }
  const std::vector<ncclComm_t>& comms = it->second;
  ncclResult_t r = a.GroupStart();
  for (int d = 0; d < nd && r == ncclSuccess; d++) {
    (void)hipSetDevice(devices[d]);
    r = a.AllGather(send[d], recv[d], count, ncclDouble, comms[d], streams[d]);
  }
  const ncclResult_t r2 = a.GroupEnd();
  (void)hipSetDevice(dev0);
  if (r == ncclSuccess) r = r2;
  if (r != ncclSuccess) { *err = std::string("ncclAllGather: ") + a.GetErrorString(r); return -1; }
  return 0;
}

}  // namespace bssm
