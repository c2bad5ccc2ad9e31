"""Writes tests/golden/reference_test_vectors.json: the INPUTS of the reference tests that hold hard-coded expected
values (regenerated with the emulation of R's RNG in r_rng.py -- the reference draws them with set.seed()/rnorm()) next
to the EXPECTED VALUES those tests state.  The expected numbers are data quoted from
  tests/testthat/test_approx.R:37-52   (gaussian_approx of an svm model after 2 iterations, fast_smoother means)
  tests/testthat/test_basics.R:183-200 (dmvnorm on a random 3 x 3 covariance, full and with a zeroed row/column)
  tests/testthat/test_ekpf.R:7-40,43-64 (nlg_sin_exp data; ekf logLik; smoothed means of the particle smoothers)
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


out = {
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
        "expected_smoothed_alphahat_9_10": [0.0263875638744833, 0.0734903567971838], "tolerance": 0.1,
    },
}
json.dump(out, open(os.path.join(HERE, "reference_test_vectors.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
