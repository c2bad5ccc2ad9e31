"""A/B probe of the workgroup-resident bootstrap filter (regime B of bench.py: 512 filters x N = 2048, m = 4, T = 1000,
log-likelihood only): scripts/ab_script.sh "scripts/regb_ab.py" 1"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bssm_amd.plan import FilterPlan
from models_zoo import negbin_m4
pb = FilterPlan(negbin_m4(1000, seed=103), 2048, method="bsf", n_rep=512, keep_history=False, force="wg")
pb.run(5); pb.loglik()
ms = min(pb.time(6, 2) for _ in range(3))
print("regime B %.2f ms  %.3e p-steps/s  %s" % (ms, 512 * 2048 * 1000 / ms * 1e3, FilterPlan.last_kernel()))
