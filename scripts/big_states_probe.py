"""Throughput of the matrix-core filters of 17 ... 64 state dimensions (pf_big.h) on bsm_ng models with a seasonal component:
period 24 (m = 25, MP = 32) and period 52 (m = 53, MP = 64), T = 200, bootstrap filter and psi-APF, log-likelihood only, batches of
independent filters -- both kernel shapes.  Roofline: the fp64 matrix cores (78.6 TFLOP/s dense on MI355X; the transition of a particle is
2 MP (m4 + k4) flops on the padded tiles, m4 / k4 = m, k rounded up to four) -> profiles/r05_big_states.txt"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
from bssm_amd import models as M
from bssm_amd.plan import FilterPlan
T = 200
rng = np.random.default_rng(1)
for period in (24, 52):
    t = np.arange(T)
    y = rng.poisson(np.exp(1.5 + 0.4 * np.sin(2 * np.pi * t / period))).astype(float)
    m = 2 + period - 1
    model = M.bsm_ng(y, sd_level=0.1, sd_slope=0.01, sd_seasonal=0.05, period=period, distribution="poisson",
                     a1=np.r_[1.5, 0.0, np.zeros(period - 1)], P1=np.diag(np.r_[1.0, 0.01, np.full(period - 1, 0.05)]))
    MP = 32 if m <= 32 else 64
    for method, ncnt in (("bsf", 3), ("psi", m)):
        m4, k4 = (m + 3) // 4 * 4, (ncnt + 3) // 4 * 4
        flops = 2.0 * MP * (m4 + k4)
        for force, N, R in (("wg", 256, 1024), ("wg", 256 if MP == 64 else 512, 256), ("row", 256, 1024), ("auto", 256, 1024), ("auto", 64, 2048), ("row", 1024, 256), ("row", 4096, 64)):
            if force == "auto": os.environ.pop("BSSM_AMD_FORCE_PATH", None)
            else: os.environ["BSSM_AMD_FORCE_PATH"] = force
            plan = FilterPlan(model, N, method=method, keep_history=False, n_rep=R)
            plan.run(1)
            ll = plan.loglik()
            t0 = time.perf_counter()
            for s in range(3):
                plan.run(2 + s)
            plan.loglik()
            e = (time.perf_counter() - t0) / 3
            ps = float(N) * R * T / e
            print("period %2d m=%2d %s %-4s N=%5d x %4d filters: %8.2f ms  %.3g particle-steps/s  %.2f TFLOP/s on the tiles = %.3f of 78.6  [%s] logLik %.3f"
                  % (period, m, method, force, N, R, e * 1e3, ps, ps * flops / 1e12, ps * flops / 78.6e12, FilterPlan.last_kernel(), ll[0]), flush=True)
            plan.close()
