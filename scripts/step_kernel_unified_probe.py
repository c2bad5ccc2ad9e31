"""One step-kernel generation (round 5): the model classes that stepped through the round-1 kernel until the end of round 4 --
sixteen-dimensional linear states (bsm_ng with a monthly seasonal, m = 13) and run-time compiled user models above the row family's
16 384 particles -- timed on whatever build PYTHONPATH points at (before: a worktree of the commit that still had pf_grid_step).
  python scripts/step_kernel_unified_probe.py"""
import os
import sys
import time

import numpy as np

if not os.environ.get("PYTHONPATH"): sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))   # (a PYTHONPATH selects the build under test)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from bssm_amd import filters as F, models as M   # noqa: E402
from bssm_amd.plan import FilterPlan             # noqa: E402


def best(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts)


rng = np.random.default_rng(13)
n = 200
tt = np.arange(n)
y = rng.poisson(np.exp(1.5 + 0.5 * np.sin(2 * np.pi * tt / 12.0))).astype(float)
a1 = np.zeros(13); a1[0] = 1.5
seasonal = M.bsm_ng(y, sd_level=0.05, sd_slope=0.005, sd_seasonal=0.02, distribution="poisson", period=12, a1=a1, P1=0.05 * np.eye(13))
for N, reps in ((16384, 16), (65536, 1), (65536, 8), (100000, 4)):
    os.environ["BSSM_AMD_FORCE_PATH"] = "grid"
    t = best(lambda: F.logLik_batch(seasonal, N, reps, method="bsf", seed=3))
    print(f"bsm_ng period 12 (m = 13), T = {n}, {reps:2d} filter(s) x {N:6d} particles: {t * 1e3:8.2f} ms  {reps * N * n / t:.3e} particle-steps/s  [{FilterPlan.last_kernel()}]", flush=True)
os.environ.pop("BSSM_AMD_FORCE_PATH")

import importlib.util   # the six-state user model of tests/test_gpu_functors.py
spec = importlib.util.spec_from_file_location("tf", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "test_gpu_functors.py"))
tf = importlib.util.module_from_spec(spec); spec.loader.exec_module(tf)
m, k, n6 = 6, 4, 200
Rm = np.zeros((m, k)); Rm[0, 0] = 0.05; Rm[1, 0] = 0.02; Rm[2, 1] = 0.3; Rm[3, 1] = -0.1; Rm[3, 2] = 0.2; Rm[4, 2] = 0.1; Rm[5, 3] = 0.4
y6 = rng.normal(size=n6)
a16 = np.array([1.0, 0.0, 0.3, 0.0, 0.0, 0.4]); P16 = np.diag([0.5, 0.01, 0.2, 0.1, 0.0, 0.3])
u = M.ssm_nlg_user(y6, tf.USER_SIX, theta=np.array([0.85]), known_params=np.zeros(0), a1=a16, P1=P16, H=0.6, n_states=m, R=Rm)
for N, reps in ((32768, 1), (32768, 8), (100000, 1), (100000, 4)):
    t = best(lambda: F.logLik_batch(u, N, reps, method="bsf", seed=3))
    print(f"user model, 6 states, T = {n6}, {reps:2d} filter(s) x {N:6d} particles: {t * 1e3:8.2f} ms  {reps * N * n6 / t:.3e} particle-steps/s  [{FilterPlan.last_kernel()}]", flush=True)
