"""Latency of ONE log-likelihood call at the sizes of the reference's pseudo-marginal MCMC benchmark
(benchmarks/replications.Rmd:27-62: bsm_ng Poisson, m=2, T=100; bsf N=200, psi N=10): this is the inner call of
run_mcmc(mcmc_type="pm"), 6.5 ms per iteration in the reference's published run (replications.html:388)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bssm_amd import filters as F, models as M
y = np.loadtxt(os.path.join(ROOT, "tests", "golden", "poisson_series.txt"))[:100]
model = M.bsm_ng(y, sd_level=M.halfnormal(0.1, 1), sd_slope=M.halfnormal(0.01, 0.1), distribution="poisson", P1=np.diag([10.0, 0.1]))
for method, N in (("bsf", 200), ("psi", 10), ("psi", 100), ("bsf", 1000)):
    F.logLik(model, N, method=method, seed=1)
    t0 = time.perf_counter()
    for s in range(200):
        ll = F.logLik(model, N, method=method, seed=s)
    e = (time.perf_counter() - t0) / 200
    print(f"logLik {method} N={N} T=100 m=2: {e*1e3:.3f} ms per call ({ll:.3f})", flush=True)

# the same inner step on a device-resident plan: theta update in place + filter + log-likelihood + one sampled trajectory
from bssm_amd.plan import FilterPlan
for method, N in (("bsf", 200), ("psi", 10)):
    plan = FilterPlan(model, N, method=method, keep_history=True)
    plan.update(model); plan.run(1); plan.loglik()
    t0 = time.perf_counter()
    for s in range(200):
        plan.update(model)
        plan.run(s)
        ll = plan.loglik()[0]
        tr = plan.sample_or_summarise(0, s, sample=True)
    e = (time.perf_counter() - t0) / 200
    print(f"plan: update + run + loglik + sampled trajectory, {method} N={N}: {e*1e3:.3f} ms per iteration ({ll:.3f})", flush=True)
    plan.close()

# component split of the plan iteration
for method, N in (("bsf", 200), ("psi", 10)):
    plan = FilterPlan(model, N, method=method, keep_history=True)
    plan.update(model); plan.run(1); plan.loglik()
    R = 300
    t0 = time.perf_counter()
    for s in range(R):
        plan.update(model)
    t_up = (time.perf_counter() - t0) / R
    t0 = time.perf_counter()
    for s in range(R):
        plan.run(s); plan.loglik()
    t_run = (time.perf_counter() - t0) / R
    t0 = time.perf_counter()
    for s in range(R):
        plan.sample_or_summarise(0, s, sample=True)
    t_s = (time.perf_counter() - t0) / R
    dev = plan.time(1, 50)
    print(f"split {method} N={N}: update {t_up*1e3:.3f} ms, run + loglik {t_run*1e3:.3f} ms (device {dev:.3f} ms), sample {t_s*1e3:.3f} ms", flush=True)
    plan.close()
