"""Times the five BASELINE.json configs (C1..C5) on one MI355X and prints one JSON object.
C1 is the CPU plumbing case (oracle only); C2..C5 go through the C ABI.  Usage: python scripts/configs.py [out.json]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from bssm_amd import filters as F, models as M, post_correct as PC
from bssm_amd.plan import FilterPlan
from models_zoo import negbin_ar1, negbin_m4, mng_c4, poisson_local_level
from oracle import oracle as O

res = {}


def timed(fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    return (time.perf_counter() - t0) / reps, out


ONLY = os.environ.get("CONFIGS_ONLY", "").split(",") if os.environ.get("CONFIGS_ONLY") else None


def want(name):
    return ONLY is None or name in ONLY


# ---- C1: bsm_ng Poisson local level, T=200, N=100: CPU restatement (plumbing) + the device call beside it
if want("C1"):
    mdl = poisson_local_level(200, seed=101)
    arr = M.model_arrays(mdl)
    e, ref = timed(lambda: O.bsf_filter(arr, 100, seed=1), 20)
    eg, out = timed(lambda: F.bootstrap_filter(mdl, 100, seed=1), 20)
    res["C1"] = dict(workload="bsm_ng Poisson local level T=200 N=100 bootstrap_filter", cpu_ms=e * 1e3, gpu_call_ms=eg * 1e3,
                     cpu_psteps_s=100 * 200 / e, loglik_cpu=ref["logLik"], loglik_gpu=out["logLik"])


# ---- C2: ssm_ung negbin AR(1), T=1000, m=1, N=65536
if want("C2"):
    for name, mk, m in (("C2", negbin_ar1, 1), ("C2-m4", negbin_m4, 4)):
        mdl = mk(1000)
        row = dict(workload="%s negbin T=1000 m=%d N=65536 bootstrap filter" % (name, m))
        for R, keep in ((1, True), (1, False), (12, True), (32, False)):
            p = FilterPlan(mdl, 65536, method="bsf", force="grid", keep_history=keep, n_rep=R)
            p.run(1); p.loglik()
            ms = p.time(2, 3)
            bps = 16 * m + (24 if keep else 16)
            row["R%d_%s" % (R, "full" if keep else "loglik")] = dict(ms=ms, psteps_s=R * 65536e3 / ms * 1e3,
                                                                    alg_GBs=R * 65536e3 / ms * 1e3 * bps / 1e9)
            p.close()
        # the drop-in call with host buffers (PCIe-inclusive)
        e, out = timed(lambda: F.bootstrap_filter(mdl, 65536, seed=1), 1)
        row["dropin_call_ms"] = e * 1e3
        row["dropin_psteps_s"] = 65536e3 / e
        res[name] = row


# ---- C3: svm on exchange, psi-APF N=32768 + back-trace + summary
if want("C3"):
    y = np.loadtxt(os.path.join(ROOT, "tests", "golden", "exchange.txt"))
    svm = M.svm(y, rho=M.uniform(0.98, -0.999, 0.999), sd_ar=M.halfnormal(0.2, 5), sigma=M.halfnormal(1.0, 2))
    e, out = timed(lambda: F.particle_smoother(svm, 32768, method="psi", seed=1), 3)
    e2, out2 = timed(lambda: F.particle_smoother(svm, 32768, method="bsf", seed=1), 3)
    e3, ll = timed(lambda: F.logLik(svm, 32768, method="psi", seed=1), 3)
    res["C3"] = dict(workload="svm exchange n=%d psi-APF N=32768 particle_smoother (host buffers in/out)" % len(y),
                     psi_smoother_ms=e * 1e3, psi_psteps_s=32768 * len(y) / e, bsf_smoother_ms=e2 * 1e3,
                     loglik_only_ms=e3 * 1e3, loglik_only_psteps_s=32768 * len(y) / e3,
                     loglik_psi=out["logLik"], loglik_bsf=out2["logLik"])


# ---- C4: ssm_mng m=8 p=4 T=2000 N=16384 (fp64 MFMA transition)
if want("C4"):
    mdl = mng_c4(2000)
    row = dict(workload="ssm_mng m=8 k=8 p=4 T=2000 N=16384 bootstrap filter")
    for R, keep in ((1, True), (16, True), (64, True), (64, False)):
        p = FilterPlan(mdl, 16384, method="bsf", force="grid", keep_history=keep, n_rep=R)
        p.run(1); p.loglik()
        ms = p.time(2, 3)
        bps = 16 * 8 + (24 if keep else 16)
        row["R%d_%s" % (R, "full" if keep else "loglik")] = dict(ms=ms, psteps_s=R * 16384 * 2000 / ms * 1e3,
                                                                alg_GBs=R * 16384 * 2000 / ms * 1e3 * bps / 1e9,
                                                                kernel=FilterPlan.last_kernel())
        p.close()
    res["C4"] = row
    # the 64-filter plan held 160 GB of device memory: hand the cached blocks back now (hipFree of that much takes
    # seconds and would otherwise be charged to the first call of the next configuration)
    from bssm_amd import _lib as _L
    _L.lib().bssm_amd_trim_cache()


# ---- C5: ssm_nlg growth, T=300, 2340 retained draws, psi N=4096, one GPU's view (all draws on this GPU)
if want("C5"):
    rng = np.random.default_rng(105)
    n, K, dT = 300, 500.0, 0.1
    r = np.zeros(n); pp = np.zeros(n); pp[0] = 50.0
    for t in range(1, n):
        r[t] = r[t - 1] + rng.normal(0, 0.05)
        rr = np.exp(r[t - 1]) / (1 + np.exp(r[t - 1]))
        pp[t] = K * pp[t - 1] * np.exp(rr * dT) / (K + pp[t - 1] * (np.exp(rr * dT) - 1)) + rng.normal(0, 1)
    yg = pp + rng.normal(0, 1, n)
    th0 = np.array([0.0, np.log(0.05), 0.0])
    growth = M.ssm_nlg(yg, "nlg_growth", theta=th0, known_params=[dT, K, 0.0, 50.0, 1.0, 100.0])
    ns = int(os.environ.get("C5_DRAWS", "2340"))
    theta = th0 + rng.normal(0, 0.05, (ns, 3))
    counts = rng.geometric(0.234, ns).astype(np.uint32)
    t0 = time.perf_counter()
    modes = np.stack([F.psi_setup(M.BssmModel(growth.classes, **{**growth, "theta": th}))["mode"].reshape(-1, order="F")
                      for th in theta[:64]], axis=1)
    e_mode = (time.perf_counter() - t0) / 64
    modes = np.tile(modes, (1, (ns + 63) // 64))[:, :ns]  # modes of nearby draws: approximate_for_is re-linearises at them
    row = dict(workload="ssm_nlg growth T=300 m=2, %d retained draws, psi-APF N=4096, IS2 post-correction on ONE GPU" % ns,
               host_mode_search_ms_per_draw=e_mode * 1e3)
    # one untimed call per output type first: the previous configuration released > 100 GB of device memory, which the
    # driver reclaims lazily -- the first large allocation afterwards waits seconds for it (not a property of this job)
    for ot in (3, 2, 1):
        PC.post_correct(growth, dict(theta=theta, counts=counts, modes=modes.reshape(2, n, ns, order="F"), output_type=ot),
                        4096, threads=8, is_type="is2", seed=1)
    for ot in (3, 2, 1):
        for thr in (1, 8, 32):
            mo = dict(theta=theta, counts=counts, modes=modes.reshape(2, n, ns, order="F"), output_type=ot)
            t0 = time.perf_counter()
            out = PC.post_correct(growth, mo, 4096, threads=thr, is_type="is2", seed=1)
            e = time.perf_counter() - t0
            row["output%d_threads%d" % (ot, thr)] = dict(s=e, psteps_s=ns * 4096.0 * n / e, draws_s=ns / e,
                                                        host_setup_ms=out["timing_ms"]["host_setup"], device_ms=out["timing_ms"]["device"],
                                                        finite=bool(np.all(np.isfinite(out["loglik"]))))
    row["host_cpus"] = os.cpu_count()
    res["C5"] = row
    # the whole workflow that produces and corrects those draws: run_mcmc(mcmc_type = "is2") = approximate chain on the
    # host (one mode search per iteration) + the post-correction above on its own stored values and modes
    from bssm_amd import mcmc
    t0 = time.perf_counter()
    fit = mcmc.run_mcmc(growth, 20000, particles=4096, mcmc_type="is2", sampling_method="psi", output_type="theta",
                        seed=1, threads=8)  # BASELINE config 5: 20 000 iterations, half of them burn-in
    e = time.perf_counter() - t0
    res["C5-chain"] = dict(workload="run_mcmc(ssm_nlg growth T=300, iter=20000, is2, psi-APF N=4096): approximate chain + IS correction",
                           s=e, chain_s=fit["time_chain"], postcorrection_s=fit["time_postcorrection"],
                           n_stored=int(fit["counts"].size), acceptance_rate=fit["acceptance_rate"],
                           psteps_s=fit["counts"].size * 4096.0 * n / fit["time_postcorrection"],
                           theta_mean=[float(v) for v in mcmc.summary(fit)["Mean"]],
                           finite=bool(np.all(np.isfinite(fit["weights"]))))

print(json.dumps(res, indent=1))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
