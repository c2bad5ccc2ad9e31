"""A/B probe of the workgroup-resident kernels across shapes (scripts/ab_script.sh "scripts/wg_ab.py" 8): regime B, the small
filters of a pseudo-marginal chain, wide kernels at m = 1 / 2, psi-APF batches of a univariate model"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bssm_amd import models as M
from bssm_amd.plan import FilterPlan
from models_zoo import negbin_m4, negbin_ar1
rng = np.random.default_rng(1)
y = rng.poisson(4.0, 100).astype(float)
bsm = M.bsm_ng(y, sd_level=0.2, sd_slope=0.05, distribution="poisson", P1=np.diag([1.0, 0.1]), a1=[1.0, 0.0])
cases = [("regB m4 N2048 x512 bsf", negbin_m4(1000, seed=103), 2048, 512, "bsf", False),
         ("m1 N8192 x256 bsf", negbin_ar1(1000, seed=102), 8192, 256, "bsf", False),
         ("m1 N4096 x512 bsf hist", negbin_ar1(300, seed=102), 4096, 512, "bsf", True),
         ("bsm m2 N64 x1 bsf (pm latency)", bsm, 64, 1, "bsf", False),
         ("bsm m2 N256 x1 psi", bsm, 256, 1, "psi", False),
         ("bsm m2 N4096 x512 psi", bsm, 4096, 512, "psi", False),
         ("bsm m2 N1024 x1024 bsf", bsm, 1024, 1024, "bsf", False)]
for name, mdl, N, R, meth, keep in cases:
    p = FilterPlan(mdl, N, method=meth, n_rep=R, keep_history=keep, force="wg")
    p.run(5); p.loglik()
    ms = min(p.time(6, 3) for _ in range(3))
    print("%-32s %8.3f ms  %s" % (name, ms, FilterPlan.last_kernel()), flush=True)
    p.close()
