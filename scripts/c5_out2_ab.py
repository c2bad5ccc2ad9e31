"""C5 post-correction with summaries (output 2): deferred summary kernel (beside the next chunk's filters) vs synchronous
(the default; BSSM_AMD_DEFER_SUMMARY=1 defers), interleaved: python scripts/c5_out2_ab.py"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import bench
from bssm_amd import filters as F, post_correct as PC
ns, nsim = 2340, 4096
model, theta, counts, n = bench._c5_job(ns)
mode = F.psi_setup(model)["mode"].reshape(-1, order="F")
modes = np.repeat(mode[:, None], ns, axis=1)
res = {}
for rnd in range(3):
    for sync in ("0", "1"):
        if sync == "0": os.environ["BSSM_AMD_DEFER_SUMMARY"] = "1"
        else: os.environ.pop("BSSM_AMD_DEFER_SUMMARY", None)
        best = 1e9
        for it in range(3):
            t0 = time.perf_counter()
            loc = PC._run_local_block_nlg(model, theta, modes, counts, 0, ns, nsim, 1, 2, 2, 32)
            best = min(best, time.perf_counter() - t0)
        res.setdefault(sync, []).append(best)
        chk = float(np.sum(loc["alphahat_i"])) + float(np.sum(loc["Vt_i"]))
        print("sync=%s %.1f ms  checksum %.12g  timing %s" % (sync, best * 1e3, chk, {k: round(v, 1) for k, v in loc["timing_ms"].items() if isinstance(v, float)}), flush=True)
