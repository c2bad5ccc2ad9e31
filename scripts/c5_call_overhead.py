"""Where the wall time of one post-correction call of a 293-draw C5 shard goes: the Python mirror's argument packing against the
library call itself (and, inside the library, BSSM_AMD_TRACE=1: set-up kernel and filters)."""
import os, sys, time, ctypes as C
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import bench
from bssm_amd import _lib, filters as F, post_correct as PC
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 293
model, theta, counts, n = bench._c5_job(ns)
mode = F.psi_setup(model)["mode"].reshape(-1, order="F")
modes = np.repeat(mode[:, None], ns, axis=1)
for it in range(5):
    t0 = time.perf_counter()
    PC._run_local_block_nlg(model, theta, modes, counts, 0, ns, 4096, 1, 2, 3, 8)
    t1 = time.perf_counter()
    print("whole mirror call %.3f ms" % ((t1 - t0) * 1e3))
# the library call alone
theta = np.ascontiguousarray(theta, dtype=np.float64)
s0, keep = _lib.pack_fn_model({**model, "theta": theta[0]})
models = (_lib.FnModel * ns)()
sz = C.sizeof(_lib.FnModel)
raw = np.frombuffer(models, dtype=np.uint8).reshape(ns, sz)
raw[:] = np.frombuffer(bytes(s0), dtype=np.uint8)
off = _lib.FnModel.theta.offset
raw[:, off:off + 8].view(np.uint64)[:, 0] = theta.ctypes.data + np.arange(ns, dtype=np.uint64) * np.uint64(theta.strides[0])
modes_c = np.asfortranarray(modes, dtype=np.float64)
counts_c = np.ascontiguousarray(counts, dtype=np.uint32)
weights, loglik = np.zeros(ns), np.zeros(ns)
head = (models, C.c_int32(0), C.c_int32(ns), _lib.dptr(modes_c), _lib.dptr(None), counts_c.ctypes.data_as(_lib.c_u32_p),
        C.c_int32(1), C.c_int32(2), C.c_int32(3), C.c_uint32(4096), C.c_uint32(1), C.c_int32(8))
tail = (_lib.dptr(weights), _lib.dptr(loglik), _lib.dptr(None), _lib.dptr(None), _lib.dptr(None))
for it in range(5):
    t0 = time.perf_counter()
    _lib.check(_lib.lib().bssm_is_correction_nlg(*head, *tail))
    t1 = time.perf_counter()
    print("library call alone %.3f ms" % ((t1 - t0) * 1e3))
