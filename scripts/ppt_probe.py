"""Particles per thread of the step kernel (BSSM_AMD_GRID_PPT) by batch size: particle-steps/s of plan.run (events)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bssm_amd.plan import FilterPlan
from models_zoo import negbin_m4, negbin_ar1
def rate(mdl, N, T, R, keep, reps=3):
    p = FilterPlan(mdl, N, method="bsf", force="grid", keep_history=keep, n_rep=R)
    p.run(1); p.loglik()
    ms = p.time(9, reps)
    p.close()
    return R * N * T / (ms * 1e-3)
for name, mk in (("m4", negbin_m4), ("m1", negbin_ar1)):
    mdl = mk(1000)
    for N in (65536, 16384):
        for R, keep in ((4, True), (8, True), (12, True), (16, True), (24, True), (32, False), (64, False)):
            if N == 65536 and R == 64: continue
            row = []
            for ppt in ("1", "2", "4"):
                os.environ["BSSM_AMD_GRID_PPT"] = ppt
                row.append(rate(mdl, N, 1000, R, keep))
            print("%s N=%5d R=%2d %s ppt1 %.3e ppt2 %.3e ppt4 %.3e  workgroups@ppt1 %d" % (name, N, R, "full" if keep else "ll  ", row[0], row[1], row[2], R * N // 256), flush=True)
