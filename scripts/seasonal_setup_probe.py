"""psi set-up of monthly-seasonal bsm_ng draws (m = 13, padded to 16 on the device: ung_setup_kernel<16>, round 5) inside a post-correction
call: the device set-up (default) against the host set-up (BSSM_AMD_DEVICE_SETUP=0, `threads` host threads) -> profiles/r05_seasonal_setup.txt"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
from bssm_amd import models as M, post_correct as PC, filters as F

rng = np.random.default_rng(5)
for period, n, ns in ((12, 200, 2048), (12, 200, 256), (7, 200, 2048)):
    m = period + 1
    tt = np.arange(n)
    y = rng.poisson(np.exp(1.6 + 0.4 * np.sin(2 * np.pi * tt / period))).astype(float)
    model = M.bsm_ng(y, sd_level=M.halfnormal(0.1, 1), sd_slope=M.halfnormal(0.02, 1), sd_seasonal=M.halfnormal(0.05, 1),
                     period=period, distribution="poisson", P1=np.diag([1.0, 0.1] + [0.5] * (period - 1)), a1=np.array([1.6] + [0.0] * period))
    theta = np.column_stack([rng.uniform(0.05, 0.2, ns), rng.uniform(0.01, 0.03, ns), rng.uniform(0.02, 0.08, ns)])
    counts = np.ones(ns, dtype=np.uint32)
    mode = F.psi_setup(model)["mode"]
    modes = np.repeat(np.asarray(mode).reshape(-1, 1), ns, axis=1)
    mo = dict(theta=theta, counts=counts, modes=modes, output_type=3)
    row = []
    ll = {}
    for dev, threads in (("2", 8), ("0", 8), ("0", 1)):
        if dev == "0" and threads == 1 and ns > 256: continue
        os.environ["BSSM_AMD_DEVICE_SETUP"] = dev
        PC.post_correct(model, mo, 64, threads=threads, is_type="is2", seed=6)
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter(); r = PC.post_correct(model, mo, 64, threads=threads, is_type="is2", seed=6); best = min(best, time.perf_counter() - t0)
        ll[dev] = r["loglik"]
        row.append("%s %8.1f ms" % ("device set-up" if dev == "2" else "host set-up, %d thread(s)" % threads, best * 1e3))
    print("bsm_ng period %2d (m = %2d), T = %d, %4d draws x psi-APF N = 64 | " % (period, m, n, ns) + " | ".join(row)
          + " | |dloglik| device vs host: max %.1e, median %.1e, draws above 1e-6: %d" % (np.abs(ll["2"] - ll["0"]).max(), np.median(np.abs(ll["2"] - ll["0"])), int((np.abs(ll["2"] - ll["0"]) > 1e-6).sum())), flush=True)
