"""Where does the per-step latency of ONE small workgroup-resident filter go?  Device time of plan.run for the PM-size
model (bsm_ng Poisson m=2, T=100) by particle count and outputs kept."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bssm_amd import models as M
from bssm_amd.plan import FilterPlan
y = np.loadtxt(os.path.join(ROOT, "tests", "golden", "poisson_series.txt"))[:100]
model = M.bsm_ng(y, sd_level=M.halfnormal(0.1, 1), sd_slope=M.halfnormal(0.01, 0.1), distribution="poisson", P1=np.diag([10.0, 0.1]))
for method in ("bsf", "psi"):
    for N in (10, 64, 128, 200, 256, 512, 1024):
        for keep in (True, False):
            for R in (1, 64):
                p = FilterPlan(model, N, method=method, keep_history=keep, n_rep=R)
                p.run(1); p.loglik()
                ms = p.time(3, 50)
                p.close()
                print("%s N=%4d keep=%d R=%2d: %.3f ms per pass, %.2f us per step" % (method, N, keep, R, ms, ms * 1e3 / 100), flush=True)
