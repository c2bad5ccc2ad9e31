"""Profiling driver: python scripts/prof_c2.py grid|wg [n_rep] [T] -- one plan, a few runs (use under rocprofv3)."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
from bssm_amd.plan import FilterPlan
from models_zoo import negbin_m4
which = sys.argv[1] if len(sys.argv) > 1 else "grid"
n_rep = int(sys.argv[2]) if len(sys.argv) > 2 else (1 if which == "grid" else 512)
T = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
if which == "grid":
    p = FilterPlan(negbin_m4(T), 65536, method="bsf", force="grid", keep_history=True, n_rep=n_rep)
else:
    p = FilterPlan(negbin_m4(T), 2048, method="bsf", force="wg", n_rep=n_rep, keep_history=False)
p.run(1); p.loglik()
print(which, n_rep, p.time(1, 2), "ms/run")
