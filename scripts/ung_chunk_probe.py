"""psi-APF post-correction of ssm_ung draws (bsm_ng negbin, T = 300, m = 2) by pipeline chunk size (BSSM_AMD_IS_CHUNK) and
where the set-up runs: python scripts/ung_chunk_probe.py [draws] [particles]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bssm_amd import models as M, post_correct as PC, filters as F
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n = 300
rng = np.random.default_rng(5)
level = np.log(5.0) + np.cumsum(rng.normal(0, 0.05, n))
y = rng.negative_binomial(5, 5 / (5 + np.exp(level))).astype(float)
model = M.bsm_ng(y, sd_level=M.halfnormal(0.1, 1), sd_slope=M.halfnormal(0.01, 1), distribution="negative binomial",
                 phi=M.gamma(5.0, 2, 0.5), P1=np.diag([1.0, 0.1]), a1=np.array([np.log(5.0), 0.0]))
theta = np.column_stack([rng.uniform(0.05, 0.2, ns), rng.uniform(0.005, 0.02, ns), rng.uniform(3, 8, ns)])
counts = np.ones(ns, dtype=np.uint32)
mode = F.psi_setup(model)["mode"].reshape(-1)
modes = np.repeat(mode[:, None], ns, axis=1)
arrs = PC.expand_models(model, theta)
for dev, th in (("2", 32), ("2", 1), ("0", 32)):
    os.environ["BSSM_AMD_DEVICE_SETUP"] = dev
    row = []
    for ch in ("", "256", "512", "1024", "2048", "4096"):
        if ch: os.environ["BSSM_AMD_IS_CHUNK"] = ch
        else: os.environ.pop("BSSM_AMD_IS_CHUNK", None)
        best = 1e9
        for rep in range(4):
            t0 = time.perf_counter()
            PC._run_local_block(arrs, modes, counts, 0, ns, N, 1, 2, 3, th)
            best = min(best, time.perf_counter() - t0)
        row.append("%s %.1f" % (ch or "default", best * 1e3))
    os.environ.pop("BSSM_AMD_IS_CHUNK", None)
    print("draws %d N %d setup=%s threads=%d ms by chunk: " % (ns, N, dev, th) + " | ".join(row), flush=True)
