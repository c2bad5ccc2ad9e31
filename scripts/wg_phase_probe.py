"""Phase timeline of the workgroup-resident filter on the C5 shape (ssm_nlg growth, psi-APF N = 4096, one workgroup of
1024 threads per draw): needs a library built with -DBSSM_WG_STAMPS (prints shader-clock totals per phase for workgroup 0).
python scripts/wg_phase_probe.py [draws] [particles]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bssm_amd import models as M, filters as F, post_correct as PC
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
rng = np.random.default_rng(105)
n, K, dT = 300, 500.0, 0.1
r = np.zeros(n); pp = np.zeros(n); pp[0] = 50.0
for t in range(1, n):
    r[t] = r[t - 1] + rng.normal(0, 0.05)
    rr = np.exp(r[t - 1]) / (1 + np.exp(r[t - 1]))
    pp[t] = K * pp[t - 1] * np.exp(rr * dT) / (K + pp[t - 1] * (np.exp(rr * dT) - 1)) + rng.normal(0, 1)
y = pp + rng.normal(0, 1, n)
th0 = np.array([0.0, np.log(0.05), 0.0])
model = M.ssm_nlg(y, "nlg_growth", theta=th0, known_params=[dT, K, 0.0, 50.0, 1.0, 100.0])
theta = th0 + rng.normal(0, 0.05, (ns, 3))
counts = np.ones(ns, dtype=np.uint32)
mode = F.psi_setup(model)["mode"].reshape(-1, order="F")
modes = np.tile(mode[:, None], (1, ns))
import time
for rep in range(2):
    t0 = time.perf_counter()
    out = PC._run_local_block_nlg(model, theta, modes, counts, 0, ns, N, 9, 2, 3, 4)
    print("pass %d: %.1f ms, timing %s" % (rep, (time.perf_counter() - t0) * 1e3, out.get("timing_ms")), flush=True)
