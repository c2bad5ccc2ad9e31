import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from bssm_amd import filters as F, models as M
from models_zoo import mng_zoo
from oracle import oracle as O
zoo = mng_zoo()
for name in sorted(zoo):
    model = zoo[name]
    for path in ("wg", "grid"):
        os.environ["BSSM_AMD_FORCE_PATH"] = path
        out = F.bootstrap_filter(model, 20, seed=6, return_indices=True)
        ref = O.mng_bsf_filter(M.model_arrays(model), 20, seed=6)
        a = np.transpose(out["alpha"], (1, 0, 2))
        bad_t = [t for t in range(model.n) if not np.array_equal(out["indices"][:, t], ref["indices"][:, t])]
        print(name, path, "ll", out["logLik"], ref["logLik"], "first bad idx col", bad_t[:3], "alpha0 diff", np.abs(a[:, 0] - ref["alpha"][:, 0]).max(),
              "w0 diff", np.abs(out["weights"][:, 0] - ref["weights"][:, 0]).max(), "alpha1 diff", np.abs(a[:, 1] - ref["alpha"][:, 1]).max())
