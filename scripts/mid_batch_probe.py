import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from bssm_amd.plan import FilterPlan
from models_zoo import negbin_m4
mdl = negbin_m4(1000)
N = 65536
for R in (12, 16, 20):
    row = []
    for v2 in ("1", "0"):
        os.environ["BSSM_AMD_GRID_V2"] = v2
        p = FilterPlan(mdl, N, method="bsf", force="grid", keep_history=True, n_rep=R)
        p.run(1); p.loglik()
        ms = p.time(9, 3)
        row.append("v2=%s %.3e (%s)" % (v2, R * N * 1000 / (ms * 1e-3), FilterPlan.last_kernel()))
        p.close()
    print("R=%d: " % R + " | ".join(row), flush=True)
