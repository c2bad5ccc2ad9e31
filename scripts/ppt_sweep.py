"""Particle-steps/s of the metric filter (m = 4 and m = 1, N = 65 536, full output) by filters in flight and particles per
lane of the round-2 step kernel (BSSM_AMD_GRID_PPT): python scripts/ppt_sweep.py [filters in flight ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bssm_amd.plan import FilterPlan
from models_zoo import negbin_m4, negbin_ar1
NP = int(os.environ.get("NP", "65536"))  # particles per filter
for name, mk in (("m4", negbin_m4), ("m1", negbin_ar1)):
    mdl = mk(1000)
    for R in ([int(a) for a in sys.argv[1:]] or (3, 4, 6, 8, 12, 16, 20, 24, 32)):
        row = []
        for ppt in ("1", "2", "4"):
            os.environ["BSSM_AMD_GRID_PPT"] = ppt
            p = FilterPlan(mdl, NP, method="bsf", force="grid", keep_history=True, n_rep=R)
            p.run(1); p.loglik()
            ms = min(p.time(9, 3) for _ in range(2))
            row.append("ppt %s %.3e" % (ppt, R * NP * 1000 / (ms * 1e-3)))
            p.close()
        print("%s R=%d: " % (name, R) + " | ".join(row), flush=True)
