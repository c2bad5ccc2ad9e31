"""Throughput of the workgroup-resident filter at the small particle counts IS post-correction uses in the
reference's own benchmarks (N = 10..200, benchmarks/replications.Rmd:27-62)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from bssm_amd.plan import FilterPlan
from models_zoo import negbin_m4, negbin_ar1, poisson_local_level
for name, mdl in (("m1 poisson T=200", poisson_local_level(200)), ("m4 negbin T=200", negbin_m4(200))):
    for method in ("bsf", "psi"):
        for N, F in ((10, 16384), (32, 16384), (64, 16384), (200, 8192), (256, 8192), (512, 4096), (1024, 2048)):
            p = FilterPlan(mdl, N, method=method, force="wg", n_rep=F)
            p.run(1); p.loglik()
            ms = p.time(2, 3)
            print(f"{name} {method} N={N} F={F}: {ms:.3f} ms  {F*N*200/ms*1e3:.3e} p-steps/s  {F/ms*1e3:.3e} filters/s", flush=True)
            p.close()
