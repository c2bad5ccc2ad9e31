"""psi set-up of bsm_ng draws with 25 (period 24) and 53 (period 52) states: the device set-up (padded to 32 / 64, lg_smoother_wave.h +
lg_factors_rows.h) beside the host set-up -- differences of the log-likelihoods and the time of a post-correction call.
Usage (GPU box): python scripts/big_setup_probe.py [draws ...]  -> gpurun_out/r06_big_setup.txt"""
import os
import sys
import time

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np  # noqa: E402
from bssm_amd import filters as F, models as M, post_correct as PC  # noqa: E402

out = open(os.path.join(root, "gpurun_out", "r06_big_setup.txt"), "w") if os.path.isdir(os.path.join(root, "gpurun_out")) else None


def say(s):
    print(s, flush=True)
    if out:
        out.write(s + "\n")
        out.flush()


draws = [int(a) for a in sys.argv[1:]] or [8, 256]
host_threads = min(32, os.cpu_count() or 1)
for period, n in ((24, 72), (52, 104)):
    rng = np.random.default_rng(5)
    m = period + 1
    tt = np.arange(n)
    y = rng.poisson(np.exp(1.6 + 0.4 * np.sin(2 * np.pi * tt / period))).astype(float)
    y[7] = np.nan
    model = M.bsm_ng(y, sd_level=M.halfnormal(0.1, 1), sd_slope=M.halfnormal(0.02, 1), sd_seasonal=M.halfnormal(0.05, 1), period=period,
                     distribution="poisson", P1=np.diag([1.0, 0.1] + [0.5] * (period - 1)), a1=np.array([1.6] + [0.0] * period))
    mode = np.asarray(F.psi_setup(model)["mode"]).reshape(-1, 1)
    for ns in draws:
        theta = np.column_stack([rng.uniform(0.05, 0.2, ns), rng.uniform(0.01, 0.03, ns), rng.uniform(0.02, 0.08, ns)])
        mo = dict(theta=theta, counts=np.ones(ns, dtype=np.uint32), modes=np.repeat(mode, ns, axis=1) + rng.normal(0, 0.03, (n, ns)), output_type=3)
        res = {}
        for dev in ("2", "0"):
            if dev == "0" and ns > 64:
                continue   # (the host set-up of many big draws takes minutes)
            os.environ["BSSM_AMD_DEVICE_SETUP"] = dev
            t0 = time.perf_counter()
            r = PC.post_correct(model, mo, 64, threads=host_threads, is_type="is2", seed=6)
            t1 = time.perf_counter() - t0
            t0 = time.perf_counter()
            r = PC.post_correct(model, mo, 64, threads=host_threads, is_type="is2", seed=6)
            t2 = time.perf_counter() - t0
            res[dev] = r
            if dev == "2":
                import ctypes
                from bssm_amd import _lib
                cnt = (ctypes.c_uint64 * 2)()
                _lib.lib().bssm_debug_thin_counts(cnt)
                say("   conditional factors: %d of %d (draw, time point >= 1) workgroups took the full decomposition" % (cnt[0], cnt[1]))
            say("m = %d, T = %d, %d draws, %s set-up (%d host threads): first call %.1f ms, second %.1f ms; finite: %s" % (
                m, n, ns, "device" if dev == "2" else "host", host_threads, t1 * 1e3, t2 * 1e3, bool(np.all(np.isfinite(r["loglik"])))))
        if "0" in res:
            d = np.abs(res["2"]["loglik"] - res["0"]["loglik"])
            say("   |dloglik| device vs host: max %.2e median %.2e" % (d.max(), np.median(d)))
