"""Cache-policy A/B probe (scripts/ab_script.sh "scripts/nt_probe.py" 12): step kernels in the shapes the non-temporal
loads / stores touch -- full output and log-likelihood only, m = 1 / 4 / 8, batches and single calls."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from bssm_amd.plan import FilterPlan
from models_zoo import negbin_m4, negbin_ar1, mng_c4
cases = [("C2-m4", negbin_m4(1000), 65536, 1000, (1, 8, 32)), ("C2-m1", negbin_ar1(1000), 65536, 1000, (12, 32)),
         ("C4", mng_c4(2000), 16384, 2000, (16, 64))]
for name, mdl, N, T, Rs in cases:
    for R in Rs:
        for keep in (True, False):
            p = FilterPlan(mdl, N, method="bsf", force="grid", keep_history=keep, n_rep=R)
            p.run(1); p.loglik()
            ms = min(p.time(2, 3) for _ in range(2))
            print(f"{name} R={R} full={int(keep)}: {R*N*T/ms*1e3:.3e} p-steps/s {FilterPlan.last_kernel()}", flush=True)
            p.close()
