"""Writes tests/golden/reference_test_vectors.json: the INPUTS of the reference tests that hold hard-coded expected
values (regenerated with the emulation of R's RNG in r_rng.py -- the reference draws them with set.seed()/rnorm()) next
to the EXPECTED VALUES those tests state.  The expected numbers are data quoted from
  tests/testthat/test_approx.R:37-52   (gaussian_approx of an svm model after 2 iterations, fast_smoother means)
  tests/testthat/test_basics.R:183-200 (dmvnorm on a random 3 x 3 covariance, full and with a zeroed row/column)
  tests/testthat/test_ekpf.R:7-40,43-69 (nlg_sin_exp data; ekf / iekf logLik; smoothed means of the particle smoothers)
  vignettes/psi_pf_experiments/*.rds   (the published Monte-Carlo study of the bootstrap filter and the psi-APF on the
                                        growth and AR-exp models: means / SDs over 10 000 runs, single-run truths)
  tests/testthat/test_mcmc.R:39-65     (posterior means of the bssm paper's negative binomial example, IS2-MCMC)
  benchmarks/replications.html         (posterior means, standard errors and run times of every MCMC type on poisson_series)
(tests/testthat/test_sde.R:57-60 states logLik = -17 "within tolerance 1" for a 50-point series whose log-likelihood is
about -83 -- all.equal scales by the actual value, so that test passes for any value below -8.5; it pins nothing and is
not used.)
"""
import json
import os

import numpy as np
from scipy.special import ndtri

from r_rng import r_rnorm, r_set_seed, r_unif

HERE = os.path.dirname(os.path.abspath(__file__))


def sin_exp_series(seed=1, n=10):
    """test_ekpf.R:7-14: y[1] <- rnorm(1, exp(x[1]), 0.1); x[i+1] <- rnorm(1, sin(x[i]), 0.1); y[i+1] <- rnorm(1, exp(x[i+1]), 0.1)
    (rnorm(1, mu, sigma) = mu + sigma * norm_rand())"""
    bg = r_set_seed(seed)

    def norm_rand():
        big = 134217728.0
        u = r_unif(bg)
        u = int(big * u) + r_unif(bg)
        return float(ndtri(u / big))
    x = np.zeros(n); y = np.zeros(n)
    y[0] = np.exp(x[0]) + 0.1 * norm_rand()
    for i in range(n - 1):
        x[i + 1] = np.sin(x[i]) + 0.1 * norm_rand()
        y[i + 1] = np.exp(x[i + 1]) + 0.1 * norm_rand()
    return y


def _norm_stream(seed):
    bg = r_set_seed(seed)

    def norm_rand():
        big = 134217728.0
        u = r_unif(bg)
        u = int(big * u) + r_unif(bg)
        return float(ndtri(u / big))
    return norm_rand


def growth_series():
    """vignettes/psi_pf.Rmd:133-150 (and :216-232): set.seed(1); r <- plogis(cumsum(c(-1.5, rnorm(n - 1, sd = 0.05))));
    p[i] <- rnorm(1, K p[i-1] exp(r[i-1] dT) / (K + p[i-1] (exp(r[i-1] dT) - 1)), 1); y <- p + rnorm(n, 0, 1)"""
    nr = _norm_stream(1)
    n, K, dT = 300, 500.0, 0.1
    e = np.array([0.05 * nr() for _ in range(n - 1)])
    r = 1.0 / (1.0 + np.exp(-np.cumsum(np.concatenate([[-1.5], e]))))
    p = np.zeros(n); p[0] = 50.0
    for i in range(1, n):
        p[i] = K * p[i - 1] * np.exp(r[i - 1] * dT) / (K + p[i - 1] * (np.exp(r[i - 1] * dT) - 1)) + nr()
    return p + np.array([nr() for _ in range(n)])


def ar_exp_series():
    """vignettes/psi_pf.Rmd:318-321: set.seed(1); alpha <- arima.sim(n = 100, list(ar = 0.95), sd = 0.1); y <- rnorm(n, exp(alpha)).
    arima.sim: burn-in n.start = p + ceiling(6 / log(min |root|)) = 1 + 117 innovations drawn first, then the n innovations,
    recursive filter x_t = e_t + 0.95 x_{t-1} from 0, burn-in dropped."""
    nr = _norm_stream(1)
    n, nstart = 100, 1 + int(np.ceil(6.0 / np.log(1.0 / 0.95)))
    e = np.array([0.1 * nr() for _ in range(nstart + n)])
    x, prev = np.zeros(nstart + n), 0.0
    for t in range(nstart + n):
        prev = e[t] + 0.95 * prev
        x[t] = prev
    return np.exp(x[nstart:]) + np.array([nr() for _ in range(n)])


def study_table(prefix, cols):
    """mean / SD over the 10 000 runs per (method, N) of every reported quantity, from the shipped result files
    vignettes/psi_pf_experiments/{gm,ar}_result_{10,100,1000}.rds (needs /root/reference; run in the build container)"""
    from rds_reader import data_frame, load_rds
    out = {}
    for N in (10, 100, 1000):
        df = data_frame(load_rds("/root/reference/vignettes/psi_pf_experiments/%s_result_%d.rds" % (prefix, N)))
        for meth, key in (("BSF", "bsf"), ("PSI", "psi"), ("EKPF", "ekf")):
            sel = df["method"] == meth
            out.setdefault(key, {})[str(N)] = {c: [float(np.mean(df[c][sel])), float(np.std(df[c][sel], ddof=1))] for c in cols}
    return out


out = {
    # The reference's published Monte-Carlo study of its own filters (10 000 replications each): mean and standard
    # deviation of the log-likelihood estimate per method and particle count, decoded from
    # vignettes/psi_pf_experiments/table1.rds / table3.rds; single-run "truth" values from gm_truth.rds / ar_truth.rds.
    "vignette_growth_model": {
        "source": "vignettes/psi_pf.Rmd:122-284, vignettes/psi_pf_experiments/{table1,gm_truth}.rds",
        "y": list(growth_series()), "theta": [0.0, float(np.log(0.05)), 0.0],   # vignette theta (H, R1, R2) = (1, 0.05, 1), unlogged there
        "known_params": [0.1, 500.0, -1.5, 50.0, 1.0, 100.0],
        "loglik_mean_sd": {"psi": {"10": [-606.2218, 0.1201], "100": [-606.2149, 0.0425], "1000": [-606.2138, 0.0157]},
                           "bsf": {"100": [-611.0850, 3.7534], "1000": [-606.6395, 0.9176]},
                           "ekf": {"100": [-607.7500, 1.8476], "1000": [-606.3727, 0.5524]}},   # EKPF N = 10: -629.56 (29.4), heavy-tailed
        "smoother_mean_sd": study_table("gm", ["logLik", "alpha_11", "alpha_21", "V_11", "V_21", "alpha_1n", "alpha_2n", "V_1n", "V_2n"]),
        "truth_bsf_1e5": {"logLik": -606.162991, "alpha_11": -1.2331292, "alpha_21": 50.2930218, "V_11": 0.0509254108,
                          "V_21": 0.695909774, "alpha_1n": -1.11484755, "alpha_2n": 497.234985, "V_1n": 0.252791374, "V_2n": 0.619154356},
    },
    "vignette_ar_exp_model": {
        "source": "vignettes/psi_pf.Rmd:316-404, vignettes/psi_pf_experiments/{table3,ar_truth}.rds",
        "y": list(ar_exp_series()), "theta": [0.0, 0.95, float(np.log(0.1)), 0.0],  # built-in nlg_ar_exp: (mu, rho, log sigma, log H)
        "loglik_mean_sd": {"psi": {"10": [-143.6263, 0.2560], "100": [-143.6000, 0.0932], "1000": [-143.5959, 0.0320]},
                           "bsf": {"10": [-144.2271, 1.2130], "100": [-143.6519, 0.3493], "1000": [-143.6016, 0.1091]},
                           "ekf": {"10": [-144.1348, 1.1224], "100": [-143.6475, 0.3316], "1000": [-143.6014, 0.1045]}},
        "smoother_mean_sd": study_table("ar", ["logLik", "alpha_1", "V_1", "alpha_n", "V_n"]),
        "truth_bsf_1e6": {"logLik": -143.594352, "alpha_1": -0.137113564, "V_1": 0.0677577735, "alpha_n": 0.206234604, "V_n": 0.0518723271},
    },
    "svm_gaussian_approx": {
        "source": "tests/testthat/test_approx.R:37-52",
        "y": list(r_rnorm(123, 5)),  # set.seed(123); rnorm(5)
        "sigma": 1.0, "rho": 0.95, "sd_ar": 0.1, "max_iter": 2, "conv_tol": 1e-8,
        "expected_approx_y": [-1.47548927809174, -11.2190916117862, 0.263154138901814, -121.519769682058, -36.0386937004332],
        "expected_approx_H": [2.01061310553144, 4.84658294043645, 0.712674409714633, 15.6217737012134, 8.54936618861792],
        "expected_fast_smoother": [-0.0999179077423753, -0.101594935319188, -0.0985572218431492, -0.103275329248674,
                                   -0.103028083292436],
    },
    "dmvnorm": {
        "source": "tests/testthat/test_basics.R:183-200",
        "rnorm9": list(r_rnorm(1, 9)),  # set.seed(1); a <- crossprod(matrix(rnorm(9), 3, 3))
        "x": [1.0, 2.0, 3.0], "mean": [-0.3, -0.2, -0.1],
        "expected_logp1": -14.0607446337904, "expected_logp3_row2_zeroed": -12.5587625856078, "tolerance": 1e-6,
    },
    "nlg_sin_exp": {
        "source": "tests/testthat/test_ekpf.R:7-40, 43-64",
        "y": list(sin_exp_series()), "theta": [float(np.log(0.1)), float(np.log(0.1))],
        "expected_ekf_loglik_with_y2_y3_missing": 2.65163101109689,
        "expected_iekf1_loglik_with_y2_y3_missing": 2.61650080342709,   # test_ekpf.R:67; :68-69: iekf_iter 1 == iekf_iter 2
        "expected_smoothed_alphahat_9_10": [0.0263875638744833, 0.0734903567971838], "tolerance": 0.1,
    },
}


def replication_table():
    """benchmarks/replications.html (knitted from benchmarks/replications.Rmd:27-100): one run_mcmc of 1e4 iterations per
    (mcmc_type, sampling_method, local_approx) on bsm_ng(poisson_series): weighted posterior means, their asymptotic
    standard errors, elapsed seconds and acceptance rates as printed in the document."""
    import html
    import re
    txt = html.unescape(re.sub(r"<[^>]+>", "", open("/root/reference/benchmarks/replications.html").read()))
    head = txt.index("arrange(local_approx, variable, mcmc_type, sampling_method)")
    rows, times = {}, {}
    for ln in txt[head:].splitlines():
        mm = re.match(r"## (\d+)\s+(-?[\d.]+)\s+([\d.]+)\s+(\w+)\s+(\w+)\s+(\w+)\s+(TRUE|FALSE)\s*$", ln)
        if mm:
            rows[int(mm.group(1))] = (float(mm.group(2)), float(mm.group(3)), mm.group(4), mm.group(5), mm.group(6), mm.group(7) == "TRUE")
            continue
        mm = re.match(r"## (\d+)\s+([\d.]+)\s+(0\.\d+)\s*$", ln)
        if mm and int(mm.group(1)) not in times:
            times[int(mm.group(1))] = (float(mm.group(2)), float(mm.group(3)))
    table = []
    for i in sorted(rows):
        mean, se, var, typ, meth, local = rows[i]
        table.append(dict(variable=var, mcmc_type=typ, sampling_method=meth, local_approx=local, mean=mean, se=se,
                          seconds=times[i][0], acceptance_rate=times[i][1]))
    assert len(table) == 104, len(table)
    return table


out["mcmc_paper_negbin"] = {
    "source": "tests/testthat/test_mcmc.R:39-65 (\"MCMC results from bssm paper are still correct\")",
    "model": "bsm_ng(negbin_series[,1], xreg = negbin_series[,2], beta = normal(0,0,10), phi = halfnormal(1,10), "
             "sd_level = halfnormal(0.1,1), sd_slope = halfnormal(0.01,0.1), a1 = c(0,0), P1 = diag(c(10,0.1)^2), "
             "distribution = 'negative binomial'); run_mcmc(iter = 6e4, burnin = 1e4, particles = 10, seed = 1)",
    "theta_order": ["coef_1", "phi", "sd_level", "sd_slope"],
    "paper_theta": [-0.912, 5.392, 0.092, 0.003],
    "paper_alpha_time_200": [6.962, 0.006],
    "tolerance_all_equal": 0.01,
}
out["mcmc_replications_poisson"] = {
    "source": "benchmarks/replications.html / benchmarks/replications.Rmd:27-100",
    "model": "s <- sd(log(pmax(0.1, poisson_series))); bsm_ng(poisson_series, sd_level = uniform(0.115, 0, 2*s), "
             "sd_slope = uniform(0.004, 0, 2*s), P1 = diag(0.1, 2), distribution = 'poisson'); iter = 1e4, "
             "end_adaptive_phase = TRUE, particles: bsf 200, psi 10, spdk 10",
    "table": replication_table(),
}
json.dump(out, open(os.path.join(HERE, "reference_test_vectors.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
