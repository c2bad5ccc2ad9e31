"""Wall time of consecutive post_correct calls on the C5 job by (output type, threads): looks for first-call costs.
python scripts/c5_outputs_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bssm_amd import models as M, filters as F, post_correct as PC
rng = np.random.default_rng(105)
n, K, dT = 300, 500.0, 0.1
r = np.zeros(n); pp = np.zeros(n); pp[0] = 50.0
for t in range(1, n):
    r[t] = r[t - 1] + rng.normal(0, 0.05)
    rr = np.exp(r[t - 1]) / (1 + np.exp(r[t - 1]))
    pp[t] = K * pp[t - 1] * np.exp(rr * dT) / (K + pp[t - 1] * (np.exp(rr * dT) - 1)) + rng.normal(0, 1)
yg = pp + rng.normal(0, 1, n)
th0 = np.array([0.0, np.log(0.05), 0.0])
growth = M.ssm_nlg(yg, "nlg_growth", theta=th0, known_params=[dT, K, 0.0, 50.0, 1.0, 100.0])
ns = 2340
theta = th0 + rng.normal(0, 0.05, (ns, 3))
counts = rng.geometric(0.234, ns).astype(np.uint32)
mode = F.psi_setup(growth)["mode"].reshape(-1, order="F")
modes = np.tile(mode[:, None], (1, ns))
seq = [(3, 1), (3, 8), (2, 1), (2, 1), (2, 8), (2, 8), (1, 1), (1, 8), (1, 8), (1, 32), (2, 1), (1, 8)]
for ot, thr in seq:
    mo = dict(theta=theta, counts=counts, modes=modes.reshape(2, n, ns, order="F"), output_type=ot)
    t0 = time.perf_counter()
    out = PC.post_correct(growth, mo, 4096, threads=thr, is_type="is2", seed=1)
    e = time.perf_counter() - t0
    print("output %d threads %2d: %.3f s  %s" % (ot, thr, e, out["timing_ms"]), flush=True)
