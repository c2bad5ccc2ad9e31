"""Functor models larger than one workgroup on pf_grid_step_v2_fn (round 5; the round-1 step kernel it was compared with in
profiles/r05_functor_v2.txt -- BSSM_AMD_GRID_V2=0 -- is gone since): particles per lane (BSSM_AMD_GRID_PPT) default / 1 / 2 / 4.  The growth model of the reference's psi-APF study (T = 300, m = 2; the study's
"truth" run uses N = 1e5 bootstrap particles, vignettes/psi_pf_experiments/gm_truth.rds): bootstrap particle_smoother and psi-APF
log-likelihood, single calls and batches -> profiles/r05_functor_v2.txt"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import bench
from bssm_amd import filters as F
from bssm_amd.plan import FilterPlan
model, theta, counts, n = bench._c5_job(4)
def timed(fn, reps=3):
    fn(); best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t0)
    return best
for N in (20000, 50000, 100000):
    for label, fn in (("particle_smoother(bsf)", lambda: F.particle_smoother(model, N, method="bsf", seed=3)),
                      ("logLik(bsf)", lambda: F.logLik(model, N, method="bsf", seed=3)),
                      ("logLik(psi)", lambda: F.logLik(model, N, method="psi", seed=3)),
                      ("logLik_batch(bsf) x 16", lambda: F.logLik_batch(model, N, 16, method="bsf", seed=3))):
        row = []
        for v in ("0", "1", "2", "4"):
            if v == "1" and N > 65536: continue
            os.environ["BSSM_AMD_GRID_PPT"] = v
            e = timed(fn)
            row.append("%s %.2f ms [%s]" % ("default" if v == "0" else v + " per lane", e * 1e3, FilterPlan.last_kernel()))
        os.environ.pop("BSSM_AMD_GRID_PPT")
        print("N = %6d %-24s | " % (N, label) + " | ".join(row), flush=True)
