"""psi-APF post-correction of ssm_mng draws: where the per-draw set-up runs (BSSM_AMD_DEVICE_SETUP = 0 host, 1 device
smoother + host Laplace step / tables, 2 everything on the device: mng_setup_kernel) x host threads.
python scripts/mng_setup_probe.py [draws] [n] [particles] [shape: c4 | m4]"""
import json
import os
import sys
import time

import numpy as np

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from bssm_amd import models as M
from bssm_amd import post_correct as PC
from bssm_amd import filters as F
from models_zoo import mng_c4, mng_zoo

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
N = int(sys.argv[3]) if len(sys.argv) > 3 else 256
shape = sys.argv[4] if len(sys.argv) > 4 else "c4"
if shape == "c4":
    base = mng_c4(n)
else:  # m = 4, p = 3 (mixed distributions), n time points
    rng = np.random.default_rng(8)
    p, m = 3, 4
    Z = rng.normal(0, 0.5, (p, m)); T = 0.9 * np.linalg.qr(rng.normal(size=(m, m)))[0]; R = 0.2 * np.eye(m)[:, :3]
    y = np.column_stack([rng.poisson(3, n), rng.negative_binomial(5, 0.5, n), rng.normal(0, 1, n)]).astype(float)
    base = M.ssm_mng(y, Z, T, R, a1=np.zeros(m), P1=0.3 * np.eye(m), distribution=["poisson", "negative binomial", "gaussian"],
                     phi=[1, 5, 0.7])
m, k = int(base.m), np.shape(base["R"])[1]
R0 = np.array(base["R"], dtype=float).reshape(m, k)


def update_fn(theta):
    return dict(R=R0 * np.exp(theta[0]))


model = M.BssmModel(base.classes, **{**base, "update_fn": update_fn, "theta": np.zeros(1)})
rng = np.random.default_rng(5)
theta = rng.normal(0, 0.1, (ns, 1))
counts = np.ones(ns, dtype=np.uint32)
mode = F.psi_setup(model)["mode"].reshape(-1, order="F")
modes = np.repeat(mode[:, None], ns, axis=1)
arrs = PC.expand_models(model, theta)
out = {"workload": f"ssm_mng {shape} m={m} p={int(base.p)} T={n}, {ns} draws, psi-APF N={N}, is2, output 3; the C call bssm_is_correction alone"}
ref = None
for dev in ("0", "1", "2"):
    os.environ["BSSM_AMD_DEVICE_SETUP"] = dev
    for th in (1, 8, 32):
        best = 1e9
        for rep in range(2):
            t0 = time.perf_counter()
            r = PC._run_local_block(arrs, modes, counts, 0, ns, N, 1, 2, 3, th)
            best = min(best, time.perf_counter() - t0)
        if ref is None: ref = r["loglik"]
        out[f"setup{dev}_threads{th}"] = {"s": round(best, 4), "max_abs_dloglik_vs_host": float(np.max(np.abs(r["loglik"] - ref)))}
print(json.dumps(out, indent=1))
