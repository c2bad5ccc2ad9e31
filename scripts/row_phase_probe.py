"""Phase timeline of the row family's step kernel (pf_row_step.h) on a tail-round sized batch: shader-clock stamps of one wave
(workgroup 3 of filter 0, wave 1) at the phase boundaries, averaged over the time steps.  Needs a library built with
-DBSSM_ROW_STAMPS (`make -C bssm_amd/csrc rowstamps` -> ab_rowstamps.so, copied over bssm_amd/libbssm_amd.so for the run: every
stamp waits for the wave's outstanding memory operations, so latency is charged to the phase that issued the loads).
python scripts/row_phase_probe.py [replicas] [particles] [T]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["BSSM_AMD_FORCE_PATH"] = "row"
import numpy as np
from bssm_amd import _lib, models as M
from bssm_amd.plan import FilterPlan
R = int(sys.argv[1]) if len(sys.argv) > 1 else 37
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
T = int(sys.argv[3]) if len(sys.argv) > 3 else 300
rng = np.random.default_rng(3)
y = rng.poisson(4, T).astype(float)
model = M.bsm_ng(y, sd_level=0.3, sd_slope=0.05, distribution="poisson", P1=np.diag([1.0, 0.1]), a1=[1.0, 0.0])
plan = FilterPlan(model, N, method="psi", keep_history=False, n_rep=R)
plan.run(1)
L = _lib.lib()
L.bssm_plan_debug_timeline.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_double)]
names = ["steps", "tables+partials+blk", "barrier", "source rows", "row loads+normals", "stage+bisect", "gather", "transition", "store+weight", "stats+publish", "-", "-"]
for rep in range(2):
    out = (C.c_double * 12)()
    _lib.check(L.bssm_plan_debug_timeline(plan._h, 2 + rep, out))
    v = np.array(out[:])
    print("kernel", FilterPlan.last_kernel(), "steps", int(v[0]), "total counter ticks per step %.0f" % v[1:].sum())
    for k in range(1, 10):
        print("  %-22s %8.1f  %5.1f %%" % (names[k], v[k], 100 * v[k] / v[1:].sum()))
