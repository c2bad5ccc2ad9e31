"""Generate golden input/output vectors for the particle-filter path from the CPU oracle.

The reference cannot run here (no R / Rcpp / Armadillo / sitmo), so these vectors pin the ORACLE
(regression) and give the GPU box committed fixtures to compare against; the oracle itself is pinned by
the analytic checks in tests/test_oracle_pins.py.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from bssm_amd import models as M  # noqa: E402
from models_zoo import mng_zoo, small_zoo  # noqa: E402
from oracle import oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    zoo = small_zoo()
    pack = {}
    for name in sorted(zoo):
        arr = M.model_arrays(zoo[name])
        N, seed = 24, 17
        bsf = O.bsf_filter(arr, N, seed=seed)
        pack[f"{name}/bsf/logLik"] = np.array(bsf["logLik"])
        pack[f"{name}/bsf/alpha"] = bsf["alpha"]
        pack[f"{name}/bsf/weights"] = bsf["weights"]
        pack[f"{name}/bsf/indices"] = bsf["indices"]
        if not name.startswith("gaussian"):
            psi = O.psi_filter(arr, N, seed=seed)
            pack[f"{name}/psi/logLik"] = np.array(psi["logLik"])
            pack[f"{name}/psi/alpha"] = psi["alpha"]
            pack[f"{name}/psi/weights"] = psi["weights"]
            pack[f"{name}/psi/indices"] = psi["indices"]
            su = psi["setup"]
            for k in ("approx_y", "approx_H", "mode", "scales", "alphahat", "Vt", "Ct"):
                pack[f"{name}/setup/{k}"] = su[k]
            pack[f"{name}/setup/approx_loglik"] = np.array(su["approx_loglik"])
    for name, mdl in sorted(mng_zoo().items()):
        arr = M.model_arrays(mdl)
        bsf = O.mng_bsf_filter(arr, 24, seed=17)
        psi = O.mng_psi_filter(arr, 24, seed=17)
        for tag, res in (("bsf", bsf), ("psi", psi)):
            pack[f"mng/{name}/{tag}/logLik"] = np.array(res["logLik"])
            pack[f"mng/{name}/{tag}/alpha"] = res["alpha"]
            pack[f"mng/{name}/{tag}/indices"] = res["indices"]
    np.savez_compressed(os.path.join(OUT, "golden_filters.npz"), **pack)
    print("wrote", len(pack), "arrays;", os.path.getsize(os.path.join(OUT, "golden_filters.npz")), "bytes")


if __name__ == "__main__":
    main()
