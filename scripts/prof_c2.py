import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from bssm_amd.plan import FilterPlan
from models_zoo import negbin_m4, negbin_ar1
which = sys.argv[1] if len(sys.argv) > 1 else "grid"
if which == "grid":
    p = FilterPlan(negbin_m4(1000), 65536, method="bsf", force="grid", keep_history=True)
else:
    p = FilterPlan(negbin_m4(1000), 2048, method="bsf", force="wg", n_rep=512, keep_history=False)
p.run(1); p.loglik()
print(which, p.time(2, 3), "ms/run")
