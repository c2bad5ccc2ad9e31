// rccl_gather.h -- the one collective of the sharded IS post-correction inside a single process:
// an all-gather of equal blocks of doubles over the GPUs of the node (RCCL over xGMI).
//
// The reference runs post_correct inside ONE R process and spreads the retained draws over OpenMP threads
// (src/approx_mcmc.cpp:364-369); here one call spreads them over the GPUs of the node, one host thread per device,
// and the per-draw weights / log-likelihoods come back through ncclAllGather on communicators made by
// ncclCommInitAll (cached per device list for the life of the process: creating them costs ~1 s).
// librccl.so is 570 MB and is only needed when a call really uses more than one GPU, so it is opened lazily with
// dlopen and never linked.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

namespace bssm {

// true when librccl could be opened and every symbol used here resolved; *why gets the reason otherwise
bool rccl_available(std::string* why);

// devices: distinct HIP ordinals.  send[d]: `count` doubles in the memory of devices[d]; recv[d]: count * devices.size()
// doubles there; streams[d]: a stream of devices[d] (the gather is enqueued on it, the caller synchronises).
// Returns 0 or -1 with *err set.  The calling thread's current device is restored.
int rccl_all_gather(const std::vector<int>& devices, const std::vector<const double*>& send, const std::vector<double*>& recv,
                    size_t count, const std::vector<hipStream_t>& streams, std::string* err);

}  // namespace bssm
