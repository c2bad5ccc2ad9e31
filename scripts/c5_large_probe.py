import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import bench
from bssm_amd import filters as F, post_correct as PC
ns = 8 * 2340
model, theta, counts, n = bench._c5_job(ns)
mode = F.psi_setup(model)["mode"].reshape(-1, order="F")
modes = np.repeat(mode[:, None], ns, axis=1)
for it in range(4):
    t0 = time.perf_counter()
    loc = PC._run_local_block_nlg(model, theta, modes, counts, 0, ns, 4096, 1, 2, 3, 32)
    print("large: %.4f s %s" % (time.perf_counter() - t0, loc["timing_ms"]), flush=True)
