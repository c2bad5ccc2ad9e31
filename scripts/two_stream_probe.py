"""One vs two interleaved launch sequences (BSSM_AMD_GRID_STREAMS) for batches of the metric model and of C2 / C4:
particle-steps/s of plan.run by wall clock and by the plan's own event timing, in both orders."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from bssm_amd.plan import FilterPlan
from models_zoo import negbin_m4, negbin_ar1, mng_c4
def rate(mdl, N, T, R, keep, reps=4):
    p = FilterPlan(mdl, N, method="bsf", force="grid", keep_history=keep, n_rep=R)
    p.run(1); p.loglik()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(reps): p.run(2 + k)
    p.loglik(); e = (time.perf_counter() - t0) / reps
    ms = p.time(9, reps)
    p.close()
    return R * N * T / e, R * N * T / (ms * 1e-3)
for name, mk, N, T in (("m1", negbin_ar1, 65536, 1000), ("m4", negbin_m4, 65536, 1000), ("c4", mng_c4, 16384, 2000)):
    mdl = mk(T)
    for R, keep in ((2, True), (4, True), (8, True), (12, True), (24, True), (32, False)):
        for order in (("1", "2"),):
            row = {}
            for s in order:
                os.environ["BSSM_AMD_GRID_STREAMS"] = s
                row[s] = rate(mdl, N, T, R, keep)
            print("%s R=%2d %s order %s: 1 seq wall %.3e ev %.3e | 2 seq wall %.3e ev %.3e" % (
                name, R, "full" if keep else "ll  ", "".join(order), row["1"][0], row["1"][1], row["2"][0], row["2"][1]), flush=True)
