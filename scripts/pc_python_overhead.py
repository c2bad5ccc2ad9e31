"""Where a post-correction call of 256 seasonal draws spends its wall time: the Python mirror (expand_models, packing the draws into C structs)
against the C ABI call itself.  Usage (GPU box): python scripts/pc_python_overhead.py [period]"""
import cProfile
import os
import pstats
import sys
import time

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np  # noqa: E402
from bssm_amd import filters as F, models as M, post_correct as PC  # noqa: E402

period = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n = 3 * period if period == 24 else (2 * period if period > 12 else 2 * period + 48)
rng = np.random.default_rng(5)
y = rng.poisson(np.exp(1.6 + 0.4 * np.sin(2 * np.pi * np.arange(n) / period))).astype(float)
model = M.bsm_ng(y, sd_level=M.halfnormal(0.1, 1), sd_slope=M.halfnormal(0.02, 1), sd_seasonal=M.halfnormal(0.05, 1), period=period,
                 distribution="poisson", P1=np.diag([1.0, 0.1] + [0.5] * (period - 1)), a1=np.array([1.6] + [0.0] * period))
mode = np.asarray(F.psi_setup(model)["mode"]).reshape(-1, 1)
theta = np.column_stack([rng.uniform(0.05, 0.2, ns), rng.uniform(0.01, 0.03, ns), rng.uniform(0.02, 0.08, ns)])
mo = dict(theta=theta, counts=np.ones(ns, dtype=np.uint32), modes=np.repeat(mode, ns, axis=1) + rng.normal(0, 0.03, (n, ns)), output_type=3)
for _ in range(2):
    t0 = time.perf_counter()
    r = PC.post_correct(model, mo, 64, threads=8, is_type="is2", seed=6)
    print("call %.2f ms; library's own split: %s" % ((time.perf_counter() - t0) * 1e3, r["timing_ms"]))
pr = cProfile.Profile()
pr.enable()
r = PC.post_correct(model, mo, 64, threads=8, is_type="is2", seed=6)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
