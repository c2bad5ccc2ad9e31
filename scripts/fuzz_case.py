"""One case of scripts/fuzz_parity.py again, with the psi set-up on the device, on the host and with the full decomposition everywhere, against the
oracle.  Usage (GPU box): python scripts/fuzz_case.py <seed0> <case index>"""
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "scripts"))
import numpy as np  # noqa: E402

seed0, idx = int(sys.argv[1]), int(sys.argv[2])
sys.argv = [sys.argv[0], "0"]
import fuzz_parity as FZ  # noqa: E402  (runs zero cases)
from bssm_amd import filters as F, models as M  # noqa: E402
from oracle import oracle as O  # noqa: E402

c = FZ.draw_case(np.random.default_rng(seed0 * 100000 + idx))
print(c["desc"])
model, n, N = c["model"], c["n"], c["N"]
arr = M.model_arrays(model)
seed = 3 + idx
ref = (O.mng_psi_filter if c["multi"] else O.psi_filter)(arr, N, seed=seed) if c["psi"] else (O.mng_bsf_filter if c["multi"] else O.bsf_filter)(arr, N, seed=seed)
for label, env in (("device set-up", {}), ("host set-up", {"BSSM_AMD_DEVICE_SETUP": "0"}), ("device, full factors", {"BSSM_AMD_FACTORS_FULL": "1"})):
    for k in ("BSSM_AMD_DEVICE_SETUP", "BSSM_AMD_FACTORS_FULL"):
        os.environ.pop(k, None)
    os.environ.update(env)
    if c["path"]:
        os.environ["BSSM_AMD_FORCE_PATH"] = c["path"]
    if c["psi"]:
        out = F.particle_smoother(model, N, method="psi", seed=seed, return_indices=True)
        sm = O.filter_smoother(ref["alpha"], ref["indices"])[:, :n, :]
    else:
        out = F.bootstrap_filter(model, N, seed=seed, return_indices=True)
        sm = ref["alpha"]
    a = np.transpose(out["alpha"], (1, 0, 2))
    d = np.abs(a - sm) / np.maximum(1.0, np.abs(sm))
    print("%-22s idx %s dll %.2e weights %.1e states %.1e; worst (component, t, particle) %s; per time point max: %s" % (
        label, np.array_equal(out["indices"], ref["indices"]), abs(out["logLik"] - ref["logLik"]), FZ.rel(out["weights"], ref["weights"]), d.max(),
        np.unravel_index(np.argmax(d), d.shape), np.array2string(d.max(axis=(0, 2)), precision=1, max_line_width=400)))
print("logLik device %r oracle %r" % (out["logLik"], ref["logLik"]))
