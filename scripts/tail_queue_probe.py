"""Does the tail of a post-correction chunk still run BESIDE its last round when the process did other GPU work first?
One 293-draw shard of C5 (one round of 256 workgroup-resident filters + 37 through the row family), device phase of the call, in a fresh
process or behind a full-output filter batch; environment switches are passed through (BSSM_AMD_PREWARM_STREAMS=0, BSSM_AMD_TAIL_CONCURRENT=0,
BSSM_AMD_TAIL_SPLIT=0, GPU_MAX_HW_QUEUES=n).  Usage: python scripts/tail_queue_probe.py fresh | plan_first [filters]
-> profiles/r05_tail_queue.txt"""
import os
import sys
import time

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from bssm_amd import filters as F, post_correct as PC  # noqa: E402
from bssm_amd.plan import FilterPlan  # noqa: E402
from models_zoo import negbin_m4_informative  # noqa: E402

torch.cuda.set_device(0)
order = sys.argv[1] if len(sys.argv) > 1 else "fresh"
if order == "plan_first":
    stream = torch.cuda.current_stream().cuda_stream
    plan = FilterPlan(negbin_m4_informative(1000, seed=103), 65536, method="bsf", keep_history=True, force="grid",
                      n_rep=int(sys.argv[2]) if len(sys.argv) > 2 else 2, factored_weights=True)
    plan.run(5, hip_stream=stream)
    torch.cuda.synchronize()
    plan.close()
for ns in (293, 2340):
    model, theta, counts, n = bench._c5_job(ns)
    mode = F.psi_setup(model)["mode"].reshape(-1, order="F")
    modes = np.repeat(mode[:, None], ns, axis=1)
    best = 1e9
    for it in range(4):
        t0 = time.perf_counter()
        loc = PC._run_local_block_nlg(model, theta, modes, counts, 0, ns, 4096, 1, 2, 3, 32)
        if it:
            best = min(best, time.perf_counter() - t0)
    print("%-10s %5d draws: call %.2f ms, device phase %.2f ms" % (order, ns, best * 1e3, loc["timing_ms"]["device"]), flush=True)
