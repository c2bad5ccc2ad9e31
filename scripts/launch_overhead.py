"""Is the one-launch-per-time-step shape launch-bound?  Host time to ENQUEUE the 1001 launches of one filter vs the time
the device needs to execute them (single filter, N = 65536, T = 1000)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from bssm_amd.plan import FilterPlan
from models_zoo import negbin_ar1, negbin_m4
st = torch.cuda.current_stream().cuda_stream
for name, mdl in (("m=1", negbin_ar1(1000)), ("m=4", negbin_m4(1000))):
    for keep in (False, True):
        p = FilterPlan(mdl, 65536, method="bsf", force="grid", keep_history=keep)
        p.run(1, hip_stream=st); torch.cuda.synchronize()
        enq, tot = [], []
        for s in range(5):
            t0 = time.perf_counter(); p.run(2 + s, hip_stream=st); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
            enq.append(t1 - t0); tot.append(t2 - t0)
        print(f"{name} history={keep}: enqueue {min(enq)*1e3:.2f} ms ({min(enq)/1001*1e6:.2f} us per launch), device-complete {min(tot)*1e3:.2f} ms")
        p.close()
