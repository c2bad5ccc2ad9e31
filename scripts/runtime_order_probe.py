import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
order = sys.argv[1]
import numpy as np
def use_lib():
    from bssm_amd import filters as F
    from models_zoo import poisson_local_level
    out = F.bootstrap_filter(poisson_local_level(20), 100, seed=1)
    print("lib ok", out["logLik"])
def use_torch():
    import torch
    print("torch cuda available:", torch.cuda.is_available())
    x = torch.ones(4, device="cuda") * 2
    print("torch ok", x.sum().item())
try:
    if order == "lib_first":
        use_lib(); use_torch(); use_lib()
    else:
        use_torch(); use_lib(); use_torch()
except Exception as e:
    print("FAILED:", type(e).__name__, str(e)[:300])
with open("/proc/self/maps") as f:
    libs = sorted({l.split()[-1] for l in f if ("hip" in l or "hsa" in l) and ".so" in l})
print("\n".join(libs))
