"""Round 6: is the overlap of a post-correction chunk's tail with its last round independent of what the process did before?
The 293-draw shard of C5 (256 workgroup-resident filters + 37 through the row family), device phase of the call, in four process
states: fresh | plan_first (a full-output filter batch ran first) | foreign (the host process created and used 16 streams of its own
before the library's first call) | foreign_plan (both).  Also as four logical shards of device 0 (run_multi_device: every shard thread
pre-warms its device).  Usage: python scripts/tail_queue_probe_r06.py <state>   -> profiles/r06_tail_queue.txt"""
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
state = sys.argv[1] if len(sys.argv) > 1 else "fresh"
keep = []
if state.startswith("foreign"):
    x = torch.zeros(1 << 20, device="cuda")
    for q in range(16):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            x.add_(1.0)
        keep.append(s)
    torch.cuda.synchronize()
if state in ("plan_first", "foreign_plan"):
    stream = torch.cuda.current_stream().cuda_stream
    plan = FilterPlan(negbin_m4_informative(1000, seed=103), 65536, method="bsf", keep_history=True, force="grid", n_rep=2, factored_weights=True)
    plan.run(5, hip_stream=stream)
    torch.cuda.synchronize()
    plan.close()
ns = 293
model, theta, counts, n = bench._c5_job(ns)
mode = F.psi_setup(model)["mode"].reshape(-1, order="F")
modes = np.repeat(mode[:, None], ns, axis=1)
best, dev = 1e9, 1e9
for it in range(6):
    t0 = time.perf_counter()
    loc = PC._run_local_block_nlg(model, theta, modes, counts, 0, ns, 4096, 1, 2, 3, 32)
    if it:
        best = min(best, time.perf_counter() - t0)
        dev = min(dev, loc["timing_ms"]["device"])
print("%-13s %5d draws: call %.2f ms, device phase %.2f ms" % (state, ns, best * 1e3, dev), flush=True)
# the SAME 293-draw job over four logical shards of device 0 (73 / 74 draws each, every shard thread pre-warms its device): same bits
mo1 = dict(theta=theta, counts=counts, modes=modes.reshape(2, n, ns, order="F"), posterior=np.zeros(ns), output_type=3)
single = PC.post_correct(model, mo1, 4096, threads=32, is_type="is2", seed=1, devices=[0])
best4 = 1e9
for it in range(5):
    t0 = time.perf_counter()
    r4 = PC.post_correct(model, mo1, 4096, threads=32, is_type="is2", seed=1, devices=[0, 0, 0, 0])
    if it:
        best4 = min(best4, time.perf_counter() - t0)
identical = bool(np.array_equal(r4["weights"], single["weights"]))
import ctypes as C  # noqa: E402
from bssm_amd import _lib  # noqa: E402
tried = C.c_int32(0)
st = _lib.lib().bssm_amd_tail_overlap_state(C.byref(tried))
print("%-13s %5d draws over devices [0,0,0,0]: call %.2f ms, weights identical to the single-shard call: %s; tail stream %s after %d candidate(s)"
      % (state, ns, best4 * 1e3, identical, {0: "not calibrated", 1: "measured beside the null stream", 2: "NOT concurrent"}[st], tried.value), flush=True)
import json  # noqa: E402
print("JSON " + json.dumps({"state": state, "single_device_phase_ms": dev, "single_call_ms": best * 1e3, "four_shards_call_ms": best4 * 1e3,
                            "identical": identical, "overlap_state": int(st), "candidates": int(tried.value)}), flush=True)
if "--short" in sys.argv:
    sys.exit(0)
# four logical shards of device 0, 293 draws each (what one node's ranks would each see, through run_multi_device's shard threads)
ns4 = 4 * ns
model, theta, counts, n = bench._c5_job(ns4)
modes = np.repeat(mode[:, None], ns4, axis=1)
mo = dict(theta=theta, counts=counts, modes=modes.reshape(2, n, ns4, order="F"), posterior=np.zeros(ns4), output_type=3)
best = 1e9
for it in range(4):
    t0 = time.perf_counter()
    r = PC.post_correct(model, mo, 4096, threads=32, is_type="is2", seed=1, devices=[0, 0, 0, 0])
    if it:
        best = min(best, time.perf_counter() - t0)
print("%-13s 4 logical shards x %d draws on device 0: call %.2f ms (shards share the GPU: ~4 x one shard's device phase)" % (state, ns, best * 1e3), flush=True)
