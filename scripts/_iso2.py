import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, time
import torch
import bench
from bssm_amd import filters as F, post_correct as PC
from bssm_amd.plan import FilterPlan
from models_zoo import negbin_m4_informative
torch.cuda.set_device(0)
if sys.argv[1] == "plan_first":
    stream = torch.cuda.current_stream().cuda_stream
    plan = FilterPlan(negbin_m4_informative(1000, seed=103), 65536, method="bsf", keep_history=True, force="grid", n_rep=int(sys.argv[2]) if len(sys.argv) > 2 else 32, factored_weights=True)
    plan.run(5, hip_stream=stream); torch.cuda.synchronize()
    plan.close()
ns = 293
model, theta, counts, n = bench._c5_job(ns)
mode = F.psi_setup(model)["mode"].reshape(-1, order="F")
modes = np.repeat(mode[:, None], ns, axis=1)
for it in range(4):
    if it == 3:
        os.environ["BSSM_AMD_TRACE"] = "1"
    t0 = time.perf_counter()
    loc = PC._run_local_block_nlg(model, theta, modes, counts, 0, ns, 4096, 1, 2, 3, 32)
    el = time.perf_counter() - t0
print("%s: 293 draws %.2f ms; timing %s" % (" ".join(sys.argv[1:]), el * 1e3, loc["timing_ms"]), flush=True)
