"""Phase timeline of the round-2 step kernel on the bench workload (C2-m4: 32 filters x 65 536 particles, full output):
shader-clock stamps of one wave (block 3 of filter 0, wave 1) at the phase boundaries, averaged over the time steps.
Needs a library built with -DBSSM_V2_STAMPS (analysis build: every stamp waits for the wave's outstanding memory
operations, so latency is charged to the phase that issued the loads).
python scripts/v2_phase_probe.py [replicas] [particles] [T]"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bssm_amd import _lib
from bssm_amd.plan import FilterPlan
from models_zoo import negbin_m4

c4 = "--c4" in sys.argv  # configuration C4 instead: ssm_mng m = 8, p = 4, N = 16 384, T = 2000, 64 filters in flight
args = [a for a in sys.argv[1:] if not a.startswith("--")]
R = int(args[0]) if len(args) > 0 else (64 if c4 else 32)
N = int(args[1]) if len(args) > 1 else (16384 if c4 else 65536)
T = int(args[2]) if len(args) > 2 else (2000 if c4 else 1000)
if c4:
    from models_zoo import mng_c4
    plan = FilterPlan(mng_c4(T), N, method="bsf", keep_history=True, force="grid", n_rep=R)
else:
    plan = FilterPlan(negbin_m4(T, seed=103), N, method="bsf", keep_history=True, force="grid", n_rep=R)
plan.run(1)
L = _lib.lib()
L.bssm_plan_debug_timeline.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_double)]
names = ["steps", "loads", "combine+barrier", "rmw", "phaseA", "phaseB", "phaseC", "slow", "phaseD", "phaseE", "density", "stats"]
for rep in range(2):
    out = (C.c_double * 12)()
    _lib.check(L.bssm_plan_debug_timeline(plan._h, 2 + rep, out))
    v = np.array(out[:])
    print("kernel", FilterPlan.last_kernel(), "steps", int(v[0]), "total cycles per step %.0f" % v[1:].sum())
    for k in range(1, 12):
        print("  %-16s %8.0f  %5.1f %%" % (names[k], v[k], 100 * v[k] / v[1:].sum()))
