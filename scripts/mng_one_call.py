import os, sys, time, json
import numpy as np
root = "/root/repo"
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from bssm_amd import models as M, post_correct as PC, filters as F
from models_zoo import mng_c4
os.environ["BSSM_AMD_DEVICE_SETUP"] = "2"
ns, n, N = 512, 200, 256
base = mng_c4(n)
m, k = int(base.m), np.shape(base["R"])[1]
R0 = np.array(base["R"], dtype=float).reshape(m, k)
model = M.BssmModel(base.classes, **{**base, "update_fn": (lambda theta: dict(R=R0 * np.exp(theta[0]))), "theta": np.zeros(1)})
rng = np.random.default_rng(5)
theta = rng.normal(0, 0.1, (ns, 1)); counts = np.ones(ns, dtype=np.uint32)
mode = F.psi_setup(model)["mode"].reshape(-1, order="F")
modes = np.repeat(mode[:, None], ns, axis=1)
arrs = PC.expand_models(model, theta)
for rep in range(3):
    t0 = time.perf_counter(); r = PC._run_local_block(arrs, modes, counts, 0, ns, N, 1, 2, 3, 8); print("call", time.perf_counter() - t0)
