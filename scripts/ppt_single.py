"""Single filter calls (and two in flight) of the metric model by particles per lane: N = 40 000 / 65 536, full output and log-likelihood only"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bssm_amd.plan import FilterPlan
from models_zoo import negbin_m4, negbin_ar1
for name, mk in (("m4", negbin_m4), ("m1", negbin_ar1)):
    mdl = mk(1000)
    for N in ([int(a) for a in sys.argv[1:]] or (40000, 65536)):
        for keep in (True, False):
            for R in (1, 2):
                row = []
                for ppt in os.environ.get("PPTS", "1,2").split(","):
                    os.environ["BSSM_AMD_GRID_PPT"] = ppt
                    p = FilterPlan(mdl, N, method="bsf", force="grid", keep_history=keep, n_rep=R)
                    p.run(1); p.loglik()
                    ms = min(p.time(9, 3) for _ in range(2))
                    row.append("ppt %s %.2f us/step" % (ppt, ms))
                    p.close()
                print("%s N=%d full=%d R=%d: " % (name, N, keep, R) + " | ".join(row), flush=True)
