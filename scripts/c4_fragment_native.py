"""C4 experiment of round 5 (review item 3): the m = 8 transition in MFMA-fragment-native layout.  pf_big_row_step<1,8> (pf_big.h: B
fragments gathered straight from the component-major particle history, the noise four normals at a time through a 2 KB tile, D
fragments stored as the new state, one workgroup barrier, one wave per row of 64 particles) against the production step kernel
pf_grid_step_v2<8,8,2,true,true,true> (two LDS transposes per transition) on BASELINE configs[3]: ssm_mng m = 8, p = 4, T = 2000,
N = 16 384, full output.  BSSM_AMD_FORCE_PATH=row selects the experiment -> profiles/r05_c4_fragment_native.txt"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
from bssm_amd import filters as F, models as M
from bssm_amd.plan import FilterPlan
from models_zoo import mng_c4
from oracle import oracle as O
# parity of the experiment against the oracle first
os.environ["BSSM_AMD_FORCE_PATH"] = "row"
small = mng_c4(40)
out = F.bootstrap_filter(small, 2048, seed=4, return_indices=True)
ref = O.mng_bsf_filter(M.model_arrays(small), 2048, seed=4)
print("parity of %s: ancestors identical %s, |dlogLik| %.3g, weights max rel %.3g" % (
    FilterPlan.last_kernel(), np.array_equal(out["indices"], ref["indices"]), abs(out["logLik"] - ref["logLik"]),
    np.max(np.abs(out["weights"] - ref["weights"]) / np.maximum(ref["weights"], 1e-300))), flush=True)
T, N = 2000, 16384
for R in (1, 16, 32, 64):
    row = []
    for path in ("grid", "row"):
        os.environ["BSSM_AMD_FORCE_PATH"] = path
        plan = FilterPlan(mng_c4(T), N, method="bsf", keep_history=True, n_rep=R, factored_weights=(path == "grid"))
        plan.run(1)
        ll = plan.loglik()
        t0 = time.perf_counter()
        for s in range(2):
            plan.run(2 + s)
        plan.loglik()
        e = (time.perf_counter() - t0) / 2
        ps = float(N) * R * T / e
        row.append("%s %.2f ms %.3g p-steps/s = %.3f of 8 TB/s [%s]" % (path, e * 1e3, ps, ps * 152 / 8e12, FilterPlan.last_kernel()))
        plan.close()
    print("C4 x %2d filters | " % R + " | ".join(row), flush=True)
