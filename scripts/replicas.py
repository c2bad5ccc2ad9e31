import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from bssm_amd.plan import FilterPlan
from models_zoo import negbin_m4, negbin_ar1
mdl = negbin_m4(1000)
for R in [int(a) for a in sys.argv[1:]] or [1, 5, 6, 8, 10, 12]:
    p = FilterPlan(mdl, 65536, method="bsf", force="grid", keep_history=True, n_rep=R)
    p.run(1); p.loglik()
    ms = p.time(2, 3)
    print(f"R={R}: {ms:.3f} ms/run {R*65536*1000/ms*1e3:.3e} p-steps/s  {R*65536*1000/ms*1e3*88/1e9:.0f} GB/s alg", flush=True)
    p.close()
