"""Probe of the in-library RCCL exchange (bssm_amd/csrc/rccl_gather.cpp) on the box: a size-one communicator on the one
visible GPU, with NCCL_DEBUG output.  Usage: NCCL_DEBUG=INFO python scripts/rccl_probe.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from bssm_amd import _lib, models as M, post_correct as PC  # noqa: E402

rng = np.random.default_rng(4)
y = rng.poisson(4.0, 30).astype(float)
model = M.bsm_ng(y, sd_level=M.halfnormal(0.2, 1), sd_slope=M.halfnormal(0.02, 1), distribution="poisson", P1=np.diag([1.0, 0.1]))
ns = 12
theta = np.log(np.column_stack([rng.uniform(0.1, 0.4, ns), rng.uniform(0.01, 0.05, ns)]))
counts = rng.integers(1, 5, ns).astype(np.uint32)
from bssm_amd import filters as F  # noqa: E402
mode = F.psi_setup(model)["mode"].reshape(-1, order="F")
modes = np.repeat(mode[:, None], ns, axis=1)
mo = dict(theta=theta, counts=counts, modes=modes, output_type=3)
print("rccl available:", _lib.lib().bssm_amd_rccl_available(), _lib.lib().bssm_amd_last_error())
one = PC.post_correct(model, mo, 32, is_type="is2", seed=5)
try:
    allg = PC.post_correct(model, mo, 32, is_type="is2", seed=5, devices="all")
    print("gather:", allg["timing_ms"], "identical:", np.array_equal(one["weights"], allg["weights"]))
except Exception as e:  # noqa: BLE001
    print("FAILED:", e)
