"""Wall time of the drop-in entry points with host buffers in and out (PCIe-inclusive), BASELINE configs C2-m4 and C3.
python scripts/dropin_probe.py   (BSSM_AMD_NO_OUTPIPE=1 for the plain-copy path)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bssm_amd import filters as F, models as M  # noqa: E402
from models_zoo import negbin_m4, negbin_ar1  # noqa: E402


def timed(fn, reps):
    # the call alone: freeing the previous result (GBs of pageable memory: ~90 ms of munmap) is outside the timed region
    out = fn()
    best = 1e9
    for _ in range(reps):
        del out
        t0 = time.perf_counter(); out = fn(); best = min(best, time.perf_counter() - t0)
    return best, out


res = {}
mdl = negbin_m4(1000, seed=103)
e, out = timed(lambda: F.bootstrap_filter(mdl, 65536, seed=1), 3)
res["C2-m4 bootstrap_filter (alpha, weights, summaries to host)"] = dict(ms=e * 1e3, psteps_s=65536e3 / e, logLik=out["logLik"])
e, out = timed(lambda: F.bootstrap_filter(mdl, 65536, seed=1, return_indices=True), 3)
res["C2-m4 bootstrap_filter + ancestor table"] = dict(ms=e * 1e3, psteps_s=65536e3 / e)
m1 = negbin_ar1(1000, seed=102)
e, out = timed(lambda: F.bootstrap_filter(m1, 65536, seed=1), 3)
res["C2 (m=1) bootstrap_filter"] = dict(ms=e * 1e3, psteps_s=65536e3 / e)
y = np.loadtxt(os.path.join(ROOT, "tests", "golden", "exchange.txt"))
svm = M.svm(y, rho=M.uniform(0.98, -0.999, 0.999), sd_ar=M.halfnormal(0.2, 5), sigma=M.halfnormal(1.0, 2))
e, out = timed(lambda: F.particle_smoother(svm, 32768, method="psi", seed=1), 3)
res["C3 psi particle_smoother"] = dict(ms=e * 1e3, psteps_s=32768 * len(y) / e)
e, out = timed(lambda: F.particle_smoother(svm, 32768, method="bsf", seed=1), 3)
res["C3 bsf particle_smoother"] = dict(ms=e * 1e3, psteps_s=32768 * len(y) / e)
print(json.dumps(res, indent=1))
