"""Mid-size batches of the metric filter (8 ... 24 filters of 65 536 particles, m = 4, full output): the default kernel
choice against the round-2 kernel at four particles per lane (BSSM_AMD_GRID_PPT=4)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bssm_amd.plan import FilterPlan
from models_zoo import negbin_m4, negbin_ar1
for name, mk in (("m4", negbin_m4), ("m1", negbin_ar1)):
    mdl = mk(1000)
    N = 65536
    for R in (6, 8, 12, 16, 20, 24):
        row = []
        for ppt in ("", "4"):
            if ppt: os.environ["BSSM_AMD_GRID_PPT"] = ppt
            else: os.environ.pop("BSSM_AMD_GRID_PPT", None)
            p = FilterPlan(mdl, N, method="bsf", force="grid", keep_history=True, n_rep=R)
            p.run(1); p.loglik()
            ms = p.time(9, 3)
            row.append("%s %.3e (%s)" % (ppt or "default", R * N * 1000 / (ms * 1e-3), FilterPlan.last_kernel()))
            p.close()
        print("%s R=%d: " % (name, R) + " | ".join(row), flush=True)
