import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from bssm_amd.plan import FilterPlan
from models_zoo import mng_c4
mdl = mng_c4(2000)
for R, keep in ((16, True), (32, True), (64, True), (64, False), (128, False)):
    p = FilterPlan(mdl, 16384, method="bsf", force="grid", keep_history=keep, n_rep=R)
    p.run(1); ll = p.loglik()
    ms = p.time(2, 3)
    bps = 16 * 8 + (24 if keep else 16)
    print(f"C4 R={R} hist={keep}: {ms:.3f} ms/run {R*16384*2000/ms*1e3:.3e} p-steps/s  {R*16384*2000/ms*1e3*bps/1e9:.0f} GB/s alg  ll={ll[0]:.3f}", flush=True)
    p.close()
