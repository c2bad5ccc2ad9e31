"""Shader-clock phase timeline of the round-1 step kernel for ONE filter of 65 536 particles (latency regime): stamps of
thread 64 of block 3 (bssm_plan_debug_timeline; the DBG instantiation pf_grid_step<4,4,1,true>).
python scripts/r1_timeline.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bssm_amd import _lib
from bssm_amd.plan import FilterPlan
from models_zoo import negbin_m4
plan = FilterPlan(negbin_m4(1000, seed=103), 65536, method="bsf", keep_history=True, force="grid", n_rep=1)
plan.run(1)
L = _lib.lib()
L.bssm_plan_debug_timeline.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_double)]
for rep in range(2):
    out = (C.c_double * 12)()
    _lib.check(L.bssm_plan_debug_timeline(plan._h, 2 + rep, out))
    v = np.array(out[:])
    print("kernel", FilterPlan.last_kernel(), "steps", int(v[0]), "cycles per step inside the kernel %.0f (%.2f us at 2.4 GHz)" % (v[1:].sum(), v[1:].sum() / 2400))
    print("  deltas between stamps 0..11:", " ".join("%.0f" % x for x in v[1:]))
ms = plan.time(9, 3)
print("wall per step %.2f us" % ms)
