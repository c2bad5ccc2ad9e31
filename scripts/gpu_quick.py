"""First-contact GPU script: one parity check + rough timings of both kernel shapes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from bssm_amd import filters as F, models as M
from bssm_amd.plan import FilterPlan
from models_zoo import negbin_m4, negbin_ar1, poisson_local_level
from oracle import oracle as O

model = poisson_local_level(50, seed=1)
for path in ("wg", "grid"):
    os.environ["BSSM_AMD_FORCE_PATH"] = path
    out = F.bootstrap_filter(model, 256, seed=1, return_indices=True)
    ref = O.bsf_filter(M.model_arrays(model), 256, seed=1)
    print(path, "loglik", out["logLik"], ref["logLik"], "idx equal", np.array_equal(out["indices"], ref["indices"]),
          "alpha maxdiff", np.abs(np.transpose(out["alpha"], (1, 0, 2)) - ref["alpha"]).max(), flush=True)
del os.environ["BSSM_AMD_FORCE_PATH"]

for name, mdl, N, kw in (("C2-m1 grid", negbin_ar1(1000), 65536, dict(force="grid")),
                         ("C2-m4 grid", negbin_m4(1000), 65536, dict(force="grid")),
                         ("C2-m4 grid hist", negbin_m4(1000), 65536, dict(force="grid", keep_history=True)),
                         ("m4 wg N=2048 x512", negbin_m4(1000), 2048, dict(force="wg", n_rep=512)),
                         ("m1 wg N=4096 x512", negbin_ar1(1000), 4096, dict(force="wg", n_rep=512)),
                         ("C2-m4 grid hist x8", negbin_m4(1000), 65536, dict(force="grid", keep_history=True, n_rep=8)),
                         ("C2-m4 grid x8", negbin_m4(1000), 65536, dict(force="grid", n_rep=8)),
                         ("m4 wg N=1024 x1024", negbin_m4(200), 1024, dict(force="wg", n_rep=1024)),
                         ("m4 wg N=64 x8192", negbin_m4(200), 64, dict(force="wg", n_rep=8192))):
    t0 = time.time()
    p = FilterPlan(mdl, N, method="bsf", **kw)
    p.run(1); ll = p.loglik()
    t1 = time.time()
    ms = p.time(2, 3)
    psteps = p.n_filters * N * mdl.n
    print(f"{name}: {p.info()} first-run {t1-t0:.2f}s  {ms:.3f} ms/run  {psteps/ms*1e3:.3e} p-steps/s  ll[0]={ll[0]:.4f} sd={ll.std():.4f}", flush=True)
    p.close()
