"""Single-call latency of the metric filter (m = 4, N = 65 536 / 32 768 / 16 384, T = 1000, full output and log-likelihood only)
with and without the speculative coarse rows (BSSM_AMD_GRID_SPEC) and the factored weights layout, interleaved:
python scripts/single_latency_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bssm_amd.plan import FilterPlan
from models_zoo import negbin_m4, negbin_ar1
for name, mk in (("m4", negbin_m4), ("m1", negbin_ar1)):
    mdl = mk(1000)
    for N in (65536, 32768, 16384):
        for keep in (True, False):
            row = []
            for rnd in range(2):
                for spec in ("0", "1"):
                    for fw in ((False, True) if keep else (False,)):
                        os.environ["BSSM_AMD_GRID_SPEC"] = spec
                        p = FilterPlan(mdl, N, method="bsf", force="grid", keep_history=keep, n_rep=1, factored_weights=fw)
                        p.run(1); ll = p.loglik()[0]
                        ms = min(p.time(9, 3) for _ in range(3))
                        row.append("spec=%s fw=%d %.2f" % (spec, fw, ms))
                        p.close()
            print("%s N=%d full=%d us/step: " % (name, N, keep) + " | ".join(row) + "  ll=%.6f  %s" % (ll, FilterPlan.last_kernel()), flush=True)
