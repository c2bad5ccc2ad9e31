"""A/B probe (scripts/ab_script.sh "scripts/ab_probe_r03.py"): single filter calls of the metric model at three sizes, the bench
batch, C4 at 64 filters -- for comparing library builds on one box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bssm_amd.plan import FilterPlan
from models_zoo import negbin_m4, mng_c4
mdl = negbin_m4(1000)
out = []
for N in (16384, 32768, 65536):
    p = FilterPlan(mdl, N, method="bsf", force="grid", keep_history=True, n_rep=1)
    p.run(1); p.loglik()
    out.append("N=%d %.2f us/step" % (N, min(p.time(9, 3) for _ in range(3))))
    p.close()
p = FilterPlan(mdl, 65536, method="bsf", force="grid", keep_history=True, n_rep=32)
p.run(1); p.loglik()
ms = min(p.time(9, 3) for _ in range(2))
out.append("batch32 %.3e" % (32 * 65536 * 1000 / ms * 1e3))
p.close()
if "--c4" in sys.argv:
    m8 = mng_c4(2000)
    p = FilterPlan(m8, 16384, method="bsf", force="grid", keep_history=True, n_rep=48)
    p.run(1); p.loglik()
    ms = min(p.time(9, 2) for _ in range(2))
    out.append("C4x48 %.3e" % (48 * 16384 * 2000 / ms * 1e3))
    p.close()
print(" | ".join(out), flush=True)
