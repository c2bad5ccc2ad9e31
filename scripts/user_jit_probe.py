"""Cold build times of a user-defined model: in process through hiprtc (bssm_user_model_jit: one kernel variant per call shape, on demand)
against the hipcc-built shared library (bssm_user_model_build: every variant of every sampler, six compiler processes side by side).
Empty caches for both.  -> profiles/r05_user_jit.txt"""
import os, sys, tempfile, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
tmp = tempfile.mkdtemp(prefix="bssm_jit_probe_"); os.chmod(tmp, 0o700)
os.environ["BSSM_AMD_USER_CACHE"] = tmp
os.environ["BSSM_AMD_TRACE"] = "1"
from bssm_amd import filters as F, models as M
import test_gpu_functors as T
b, _ = T.MODELS["nlg_growth"]
th, kp = np.asarray(b["theta"]), np.asarray(b["known_params"])
F.bootstrap_filter(b, 64, seed=1)   # the library itself warm (device arena, module load)
def mk(jit): return M.ssm_nlg_user(b["y"], T.USER_GROWTH, theta=th, known_params=kp, a1=kp[2:4], P1=np.diag(kp[4:6]), H=np.exp(th[0]), R_diag=np.exp(th[1:3]), jit=jit)
for jit in (True, False):
    u = mk(jit)
    for N in (300, 300, 2000, 12000, 12000):
        t0 = time.perf_counter()
        out = F.bootstrap_filter(u, N, seed=4)
        print("%s  N = %5d bootstrap_filter: %8.1f ms  logLik %.6f" % ("hiprtc, in process" if jit else "hipcc, shared library", N, (time.perf_counter() - t0) * 1e3, out["logLik"]), flush=True)
