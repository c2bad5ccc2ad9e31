"""Re-runs the reference's MCMC replication document (benchmarks/replications.Rmd:27-100) on one MI355X and prints the
posterior means, standard errors and run times next to the numbers the knitted document reports (quoted in
tests/golden/reference_test_vectors.json).  SPDK is not a particle-filter path and is left out.
Usage: python scripts/replications.py [out.json]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from bssm_amd import mcmc, models as M


def asymptotic_se(x, w):
    return float(np.sqrt(mcmc.asymptotic_var(x, w)))


def main():
    y = np.loadtxt(os.path.join(ROOT, "tests", "golden", "poisson_series.txt"))
    s = np.std(np.log(np.maximum(0.1, y)), ddof=1)
    model = M.bsm_ng(y, sd_level=M.uniform(0.115, 0, 2 * s), sd_slope=M.uniform(0.004, 0, 2 * s), P1=np.diag([0.1, 0.1]),
                     distribution="poisson")
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_test_vectors.json")))["mcmc_replications_poisson"]["table"]
    rows = []
    threads = int(os.environ.get("BSSM_THREADS", "8"))
    mcmc.run_mcmc(model, 200, particles=10, mcmc_type="pm", seed=1)  # library / device warm-up
    grid = (("approx", "psi"), ("da", "bsf"), ("da", "psi"), ("is1", "bsf"), ("is1", "psi"), ("is2", "bsf"), ("is2", "psi"),
            ("pm", "bsf"), ("pm", "psi"))
    for local, typ, meth in [(lo, t, m) for lo in (True, False) for t, m in grid]:
        t0 = time.perf_counter()
        res = mcmc.run_mcmc(model, 10000, particles={"bsf": 200, "psi": 10}[meth], mcmc_type=typ, sampling_method=meth,
                            end_adaptive_phase=True, seed=2021, threads=threads, local_approx=local)
        el = time.perf_counter() - t0
        w = res["counts"] * (res["weights"] if typ in ("is1", "is2") else 1.0)
        series = dict(sd_level=res["theta"][:, 0], sd_slope=res["theta"][:, 1])
        if "alpha" in res:  # the approximate chain keeps no state draws here (approx_state_posterior is not a particle path)
            series.update(u_1=res["alpha"][0, 0, :], u_100=res["alpha"][99, 0, :])
        for var, x in series.items():
            r = [q for q in ref if q["variable"] == var and q["mcmc_type"] == typ and q["sampling_method"] == meth and q["local_approx"] == local][0]
            mean = float(np.sum(w * x) / np.sum(w))
            rows.append(dict(variable=var, mcmc_type=typ, sampling_method=meth, local_approx=local, mean=mean, se=asymptotic_se(x, w), seconds=el,
                             acceptance_rate=res["acceptance_rate"], reference_mean=r["mean"], reference_se=r["se"],
                             reference_seconds=r["seconds"], reference_acceptance_rate=r["acceptance_rate"],
                             z=(mean - r["mean"]) / np.hypot(r["se"], asymptotic_se(x, w))))
    print("%-9s %-7s %-4s %-6s %10s %9s %10s %9s %6s %8s %8s" % ("variable", "type", "pf", "local", "mean", "se", "ref mean", "ref se", "z", "sec", "ref sec"))
    for r in rows:
        print("%-9s %-7s %-4s %-6s %10.5f %9.5f %10.5f %9.5f %6.2f %8.2f %8.2f" % (
            r["variable"], r["mcmc_type"], r["sampling_method"], r["local_approx"], r["mean"], r["se"], r["reference_mean"], r["reference_se"], r["z"],
            r["seconds"], r["reference_seconds"]))
    if len(sys.argv) > 1:
        json.dump(rows, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
