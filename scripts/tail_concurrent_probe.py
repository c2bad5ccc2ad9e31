"""The tail of a post-correction chunk BESIDE its last full round (round 5, api_postcorrect.cpp; BSSM_AMD_TAIL_CONCURRENT=0: behind it) on shapes
other than C5: bsm_ng local linear trend (m = 2, N = 4096) and a four-state ssm_ung (N = 3000), psi-APF, 293 / 549 draws; wall time of the call and
bit-identity of the outputs -> profiles/r05_tail_concurrent.txt"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
from bssm_amd import models as M, post_correct as PC, filters as F
from bssm_amd.plan import FilterPlan

rng = np.random.default_rng(7)
n = 300
y = rng.poisson(np.exp(1.0 + np.cumsum(rng.normal(0, 0.05, n)))).astype(float)
llt = M.bsm_ng(y, sd_level=M.halfnormal(0.1, 1), sd_slope=M.halfnormal(0.01, 0.1), distribution="poisson", P1=np.diag([1.0, 0.1]), a1=np.array([1.0, 0.0]))
Z4 = np.array([1.0, 0.5, 0.2, 0.1]); T4 = np.diag([0.9, 0.8, 0.7, 0.6]); R4 = 0.2 * np.eye(4)
ung4 = M.ssm_ung(y, Z=Z4, T=T4, R=R4, a1=np.zeros(4), P1=np.eye(4), distribution="poisson")
for name, model, N, theta_of in (("bsm_ng m = 2, N = 4096", llt, 4096, lambda k: np.column_stack([rng.uniform(0.05, 0.2, k), rng.uniform(0.005, 0.02, k)])),
                                 ("ssm_ung m = 4, N = 3000", ung4, 3000, None)):
    for ns in (293, 549):
        if theta_of is None:
            theta = np.zeros((ns, 0))
        else:
            theta = theta_of(ns)
        mode = np.asarray(F.psi_setup(model)["mode"]).reshape(-1, 1)
        mo = dict(theta=theta, counts=np.ones(ns, dtype=np.uint32), modes=np.repeat(mode, ns, axis=1), output_type=3)
        res = {}
        row = []
        for c in ("0", "1"):
            os.environ["BSSM_AMD_TAIL_CONCURRENT"] = c
            PC.post_correct(model, mo, N, threads=8, is_type="is2", seed=6)
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter(); r = PC.post_correct(model, mo, N, threads=8, is_type="is2", seed=6); best = min(best, time.perf_counter() - t0)
            res[c] = r
            row.append("%s %.2f ms" % ("behind" if c == "0" else "beside", best * 1e3))
        same = np.array_equal(res["0"]["weights"], res["1"]["weights"]) and np.array_equal(res["0"]["loglik"], res["1"]["loglik"])
        print("%s, T = %d, %d draws | tail %s | outputs bit-identical: %s  [%s]" % (name, n, ns, " | ".join(row), same, FilterPlan.last_kernel()), flush=True)
