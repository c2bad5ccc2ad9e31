import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from bssm_amd.plan import FilterPlan
from models_zoo import negbin_m4
mdl = negbin_m4(1000)
for R in (1, 4, 8):
    p = FilterPlan(mdl, 65536, method="bsf", force="grid", keep_history=True, n_rep=R)
    p.run(1); p.loglik()
    ms = min(p.time(9, 3) for _ in range(3))
    print("R=%d %.3e p-steps/s %.2f us/step %s" % (R, R * 65536 * 1000 / (ms * 1e-3), ms, FilterPlan.last_kernel()))
    p.close()
