"""post-correction of bsm_ng draws over seasonal periods 4 ... 52 (5 ... 53 states): ms per call and per draw, device set-up.
Usage (GPU box): python scripts/seasonal_probe.py [draws [periods ...]]  -> gpurun_out/r06_seasonal_probe.txt"""
import os
import sys
import time

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np  # noqa: E402
from bssm_amd import filters as F, models as M, post_correct as PC  # noqa: E402

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
periods = [int(a) for a in sys.argv[2:]] or [4, 7, 12, 24, 52]
out = open(os.path.join(root, "gpurun_out", "r06_seasonal_probe.txt"), "w")
for period in periods:
    n = 2 * period + 48
    rng = np.random.default_rng(5)
    tt = np.arange(n)
    y = rng.poisson(np.exp(1.6 + 0.4 * np.sin(2 * np.pi * tt / period))).astype(float)
    y[7] = np.nan
    model = M.bsm_ng(y, sd_level=M.halfnormal(0.1, 1), sd_slope=M.halfnormal(0.02, 1), sd_seasonal=M.halfnormal(0.05, 1), period=period,
                     distribution="poisson", P1=np.diag([1.0, 0.1] + [0.5] * (period - 1)), a1=np.array([1.6] + [0.0] * period))
    mode = np.asarray(F.psi_setup(model)["mode"]).reshape(-1, 1)
    theta = np.column_stack([rng.uniform(0.05, 0.2, ns), rng.uniform(0.01, 0.03, ns), rng.uniform(0.02, 0.08, ns)])
    mo = dict(theta=theta, counts=np.ones(ns, dtype=np.uint32), modes=np.repeat(mode, ns, axis=1) + rng.normal(0, 0.03, (n, ns)), output_type=3)
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        r = PC.post_correct(model, mo, 64, threads=min(32, os.cpu_count() or 1), is_type="is2", seed=6)
        ts.append(time.perf_counter() - t0)
    s = "period %2d: m = %2d, T = %3d, %d draws x psi-APF N = 64: %.1f ms (first call %.1f), %.1f us per draw; finite: %s" % (
        period, period + 1, n, ns, min(ts[1:]) * 1e3, ts[0] * 1e3, min(ts[1:]) * 1e6 / ns, bool(np.all(np.isfinite(r["loglik"]))))
    print(s, flush=True)
    out.write(s + "\n")
out.close()
