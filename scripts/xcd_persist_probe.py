"""XCD-resident persistent step kernel (pf_xcd_persist_v2, BSSM_AMD_XCD_PERSIST=1) against one launch per step: time per
pass and per step, and bit-identity of the outputs, by shape and filters in flight -> profiles/r04_xcd_persistent.txt

    python scripts/xcd_persist_probe.py [c4|m4|all]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from bssm_amd import _lib  # noqa: E402
from bssm_amd.plan import FilterPlan  # noqa: E402
from models_zoo import mng_c4, negbin_m4_informative  # noqa: E402

st = torch.cuda.current_stream().cuda_stream
what = sys.argv[1] if len(sys.argv) > 1 else "all"


def run(model, N, R, T, persist, check=None, keep=True, ppt=None):
    if persist:
        os.environ["BSSM_AMD_XCD_PERSIST"] = "1"
    else:
        os.environ.pop("BSSM_AMD_XCD_PERSIST", None)
    if ppt:
        os.environ["BSSM_AMD_GRID_PPT"] = str(ppt)
    else:
        os.environ.pop("BSSM_AMD_GRID_PPT", None)
    p = FilterPlan(model, N, method="bsf", n_rep=R, keep_history=keep, force="grid")
    p.run(3, hip_stream=st); torch.cuda.synchronize()
    k = FilterPlan.last_kernel()
    reps = 3 if R * N >= 1 << 19 else 8
    t0 = time.perf_counter()
    for q in range(reps):
        p.run(5 + q, hip_stream=st)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / reps
    p.run(77, hip_stream=st)
    ll = p.loglik(st)
    ix = p.fetch(R - 1, hip_stream=st, alpha=False, weights=False)["indices"] if keep else None
    p.close()
    _lib.lib().bssm_amd_trim_cache()
    same = None
    if check is not None:
        same = bool(np.array_equal(ll, check[0]) and (ix is None or np.array_equal(ix, check[1])))
    return el, k, (ll, ix), same


def sweep(name, model, N, T, counts, ppt=None):
    for R in counts:
        e0, k0, ref, _ = run(model, N, R, T, False, ppt=ppt)
        e1, k1, _, same = run(model, N, R, T, True, check=ref, ppt=ppt)
        print("%-8s N=%6d filters=%3d  per-step launches: %8.2f ms (%6.2f us/step, %.3g p-steps/s) %s | persistent: %8.2f ms (%6.2f us/step, %.3g p-steps/s) %s | x%.2f | identical: %s"
              % (name, N, R, e0 * 1e3, e0 / T * 1e6, N * float(T) * R / e0, k0, e1 * 1e3, e1 / T * 1e6, N * float(T) * R / e1, k1, e0 / e1, same), flush=True)


if what == "dbg":   # where the time goes: steps without a body (barrier cost), steps without waiting (body cost; results are garbage)
    mdl = mng_c4(2000)
    for R in (1, 64):
        for dbg, tag in ((0, "normal"), (2, "no body: barriers only"), (4, "no waiting: bodies only"), (16, "body inlined in the loop"), (20, "inlined, no waiting")):
            os.environ["BSSM_AMD_XCD_PERSIST_DBG"] = str(dbg)
            e1, k1, _, _ = run(mdl, 16384, R, 2000, True)
            print("C4 filters=%2d %-26s %8.2f ms  %6.2f us/step  %s" % (R, tag, e1 * 1e3, e1 / 2000 * 1e6, k1), flush=True)
    os.environ.pop("BSSM_AMD_XCD_PERSIST_DBG", None)
if what in ("c4", "all"):
    sweep("C4", mng_c4(2000), 16384, 2000, (1, 8, 16, 32, 64))
    sweep("C4 ppt1", mng_c4(2000), 16384, 2000, (1, 8), ppt=1)
if what in ("m4", "all"):
    sweep("m4", negbin_m4_informative(1000), 16384, 1000, (1, 8, 32, 64))
    sweep("m4", negbin_m4_informative(1000), 65536, 1000, (1, 8, 32))
