import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from bssm_amd import _lib
from bssm_amd.plan import FilterPlan
from models_zoo import negbin_m4
p = FilterPlan(negbin_m4(1000), 65536, method="bsf", force="grid", keep_history=True)
p.run(1); p.loglik()
out = (C.c_double * 12)()
L = _lib.lib()
L.bssm_plan_debug_timeline.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_double)]
_lib.check(L.bssm_plan_debug_timeline(p._h, 3, out))
names = ["samples", "0->1 partial loads+rng", "1->2 combine", "2->3 level1 search", "3->4 coarse", "4->5 fine", "5->6 gather",
         "6->7 propagate+store", "7->8 density", "8->9 block max+exp", "9->10 scan", "10->11 stores"]
tot = sum(out[1:])
for nme, v in zip(names, out):
    print(f"{nme:28s} {v:10.0f}")
print("total ticks", tot)
