"""psi-APF post-correction of ssm_ung draws: where the per-draw set-up runs (BSSM_AMD_DEVICE_SETUP = 0 host, 1 device
smoother + host Laplace step / tables, 2 everything on the device = the default) x host threads.
python scripts/ung_setup_probe.py [draws] [n] [particles]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bssm_amd import models as M
from bssm_amd import post_correct as PC
from bssm_amd import filters as F

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
N = int(sys.argv[3]) if len(sys.argv) > 3 else 256
rng = np.random.default_rng(5)
level = np.log(5.0) + np.cumsum(rng.normal(0, 0.05, n))
y = rng.negative_binomial(5, 5 / (5 + np.exp(level))).astype(float)
model = M.bsm_ng(y, sd_level=M.halfnormal(0.1, 1), sd_slope=M.halfnormal(0.01, 1), distribution="negative binomial",
                 phi=M.gamma(5.0, 2, 0.5), P1=np.diag([1.0, 0.1]), a1=np.array([np.log(5.0), 0.0]))
theta = np.column_stack([rng.uniform(0.05, 0.2, ns), rng.uniform(0.005, 0.02, ns), rng.uniform(3, 8, ns)])
counts = np.ones(ns, dtype=np.uint32)
mode = F.psi_setup(model)["mode"].reshape(-1)
modes = np.repeat(mode[:, None], ns, axis=1)
t0 = time.perf_counter()
arrs = PC.expand_models(model, theta)  # the Python mirror's per-draw update_model loop: not part of the library call
out = {"workload": f"bsm_ng negbin T={n} m=2, {ns} draws, psi-APF N={N}, is2, output 3; the C call bssm_is_correction alone",
       "python_expand_models_s": round(time.perf_counter() - t0, 4)}
ref = None
for dev in ("0", "1", "2"):
    os.environ["BSSM_AMD_DEVICE_SETUP"] = dev
    for th in (1, 8, 32):
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            r = PC._run_local_block(arrs, modes, counts, 0, ns, N, 1, 2, 3, th)
            best = min(best, time.perf_counter() - t0)
        dt = best
        if ref is None: ref = r["loglik"]
        out[f"setup{dev}_threads{th}"] = {"s": round(dt, 4), "max_abs_dloglik_vs_host": float(np.max(np.abs(r["loglik"] - ref)))}
print(json.dumps(out, indent=1))
