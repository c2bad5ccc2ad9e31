"""psi post-correction of bsm_ng draws with a weekly seasonal (level + slope + 6 seasonal states: m = 8): the device set-up with the
row-distributed smoother (default) against the one-lane-per-draw form (BSSM_AMD_MNG_LANE_PER_DRAW=1) and the host set-up.
python scripts/ung_m8_probe.py [draws] [n]"""
import os, sys, time, json
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from bssm_amd import models as M, post_correct as PC, filters as F
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.default_rng(3)
y = rng.poisson(6.0, n).astype(float)
model = M.bsm_ng(y, sd_level=M.halfnormal(0.1, 1), sd_slope=M.halfnormal(0.02, 1), sd_seasonal=M.halfnormal(0.05, 1), period=7,
                 distribution="poisson", P1=np.diag([1.0, 0.1] + [0.5] * 6), a1=np.array([1.7] + [0.0] * 7))
theta = np.column_stack([rng.uniform(0.05, 0.2, ns), rng.uniform(0.01, 0.03, ns), rng.uniform(0.02, 0.08, ns)])
counts = np.ones(ns, dtype=np.uint32)
mode = np.asarray(F.psi_setup(model)["mode"]).reshape(-1)
modes = np.repeat(mode[:, None], ns, axis=1)
arrs = PC.expand_models(model, theta)
out = {"workload": f"bsm_ng poisson, period 7 (m = 8), T = {n}, {ns} draws, psi-APF N = 256, is2, output 3, 16 host threads"}
ref = None
for name, env in (("device_rows", {"BSSM_AMD_DEVICE_SETUP": "2"}), ("device_one_lane", {"BSSM_AMD_DEVICE_SETUP": "2", "BSSM_AMD_MNG_LANE_PER_DRAW": "1"}),
                  ("host", {"BSSM_AMD_DEVICE_SETUP": "0"})):
    for k in ("BSSM_AMD_DEVICE_SETUP", "BSSM_AMD_MNG_LANE_PER_DRAW"): os.environ.pop(k, None)
    os.environ.update(env)
    best = 1e9
    for rep in range(2):
        t0 = time.perf_counter(); r = PC._run_local_block(arrs, modes, counts, 0, ns, 256, 1, 2, 3, 16); best = min(best, time.perf_counter() - t0)
    if ref is None: ref = r["loglik"]
    dl = np.abs(r["loglik"] - ref)
    # (a set-up that differs in the last places moves an ancestor now and then; that draw's estimate is then a different, equally valid one)
    out[name] = {"s": round(best, 4), "max_abs_dloglik_vs_rows": float(dl.max()), "median_abs_dloglik_vs_rows": float(np.median(dl)),
                 "draws_beyond_1e-6": int(np.sum(dl > 1e-6))}
print(json.dumps(out, indent=1))
