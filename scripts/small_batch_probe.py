"""Small batches of the metric filter (1, 2, 3, 4, 6, 8 filters of 65 536 particles, m = 4, full output) by particles per
lane: which kernel runs and the particle-steps/s (BSSM_AMD_GRID_PPT hook)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bssm_amd.plan import FilterPlan
from models_zoo import negbin_m4
mdl = negbin_m4(1000)
N = 65536
for R in (1, 2, 3, 4, 6, 8):
    row = []
    for ppt in ("", "1", "2", "4"):
        if ppt: os.environ["BSSM_AMD_GRID_PPT"] = ppt
        else: os.environ.pop("BSSM_AMD_GRID_PPT", None)
        p = FilterPlan(mdl, N, method="bsf", force="grid", keep_history=True, n_rep=R)
        p.run(1); p.loglik()
        ms = p.time(9, 3)
        row.append("%s %.3e (%s, %.2f us/step)" % (ppt or "default", R * N * 1000 / (ms * 1e-3), FilterPlan.last_kernel(), ms))
        p.close()
    print("R=%d: " % R + " | ".join(row), flush=True)
