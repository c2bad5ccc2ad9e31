"""A grid of single calls -- state dimension x particles x entry point -- to spot a configuration that is out of line with its neighbours
(round 6: the 16-state set-up smoother was 8 x slower than the 32-state one and nothing in the bench line showed it).
Usage (GPU box): python scripts/sweep_throughput.py  -> gpurun_out/r06_sweep_throughput.txt"""
import os
import sys
import time

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np  # noqa: E402
from bssm_amd import filters as F, models as M  # noqa: E402

out = open(os.path.join(root, "gpurun_out", "r06_sweep_throughput.txt"), "w")


def say(s):
    print(s, flush=True)
    out.write(s + "\n")


def model_of(m, n, rng):
    y = rng.poisson(4.0, n).astype(float)
    if m == 1:
        return M.bsm_ng(y, sd_level=0.1, distribution="poisson")
    if m == 2:
        return M.bsm_ng(y, sd_level=0.1, sd_slope=0.01, distribution="poisson", P1=np.diag([1.0, 0.1]))
    k = min(m, 3)
    T = 0.95 * np.linalg.qr(rng.normal(size=(m, m)))[0]
    return M.ssm_ung(y, Z=rng.normal(0, 0.5 / np.sqrt(m), m), T=T, R=0.1 * rng.normal(size=(m, k)), a1=np.zeros(m), P1=0.1 * np.eye(m), distribution="poisson")


def best(fn, reps=3):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts)


n = 200
say("T = %d; ms per call (particle-steps/s): bootstrap_filter full output | logLik bsf | logLik psi | particle_smoother psi" % n)
for m in (1, 2, 3, 4, 5, 8, 9, 13, 16, 17, 25, 32, 33, 53, 64):
    rng = np.random.default_rng(m)
    model = model_of(m, n, rng)
    for N in (100, 1000, 8192, 16384, 65536):
        row = []
        for name, fn in (("bsf", lambda: F.bootstrap_filter(model, N, seed=1)), ("ll_bsf", lambda: F.logLik(model, N, method="bsf", seed=1)),
                         ("ll_psi", lambda: F.logLik(model, N, method="psi", seed=1)), ("sm_psi", lambda: F.particle_smoother(model, N, method="psi", seed=1))):
            if name in ("bsf", "sm_psi") and m * N * n * 8 > 6e9:
                row.append("      -      ")
                continue
            try:
                t = best(fn, reps=2 if N >= 65536 else 3)
                row.append("%7.2f (%.1e)" % (t * 1e3, N * n / t))
            except Exception as e:
                row.append("ERR " + repr(e)[:30])
        say("m = %2d N = %6d : %s" % (m, N, " | ".join(row)))
out.close()
