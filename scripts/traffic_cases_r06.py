"""One case per run for the HBM-traffic counter passes of round 6 (scripts/profile_traffic_r06.sh): a batch of filters stepped a few times, then the
calibration copy (2 GiB read + 2 GiB written: the gfx950 FETCH_SIZE factor).  Usage: python scripts/traffic_cases_r06.py big24|big52|c4"""
import ctypes
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np  # noqa: E402
from bssm_amd import _lib, models as M  # noqa: E402
from bssm_amd.plan import FilterPlan  # noqa: E402

case = sys.argv[1]
if case.startswith("big"):
    period, T = int(case[3:]), 40
    rng = np.random.default_rng(1)
    y = rng.poisson(np.exp(1.5 + 0.4 * np.sin(2 * np.pi * np.arange(T) / period))).astype(float)
    model = M.bsm_ng(y, sd_level=0.1, sd_slope=0.01, sd_seasonal=0.05, period=period, distribution="poisson",
                     a1=np.r_[1.5, 0.0, np.zeros(period - 1)], P1=np.diag(np.r_[1.0, 0.01, np.full(period - 1, 0.05)]))
    plan = FilterPlan(model, 1024, method="bsf", n_rep=256, keep_history=False)
else:
    from models_zoo import mng_c4  # noqa: E402
    plan = FilterPlan(mng_c4(40), 16384, method="bsf", keep_history=True, force="grid", n_rep=64)
plan.run(1)
plan.loglik()
print("kernel:", FilterPlan.last_kernel(), flush=True)
plan.close()
_lib.check(_lib.lib().bssm_debug_stream_copy(ctypes.c_uint64(1 << 28)))
