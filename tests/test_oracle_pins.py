"""Pins of the CPU oracle (oracle/bssm_oracle.c).  The reference cannot be built or run here and its
tests hold no particle-filter golden vectors, so the oracle is pinned by

 * the analytic checks the reference's own tests use (PF == exact Kalman smoother on a linear-Gaussian
   model, psi-APF == bootstrap filter; tests/testthat/test_particle_smoother.R:25-45),
 * closed-form values (Philox known-answer vectors, distribution constants vs scipy, Kalman likelihood
   vs a dense multivariate-normal evaluation),
 * the invariants of tests/testthat/test_bootstrap_filter.R:21-107 on the models defined there,
 * the committed golden vectors (regression)."""
import os

import numpy as np
import pytest
from scipy import stats

from bssm_amd import models as M
from models_zoo import small_zoo

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_philox_known_answers(oracle):
    # Random123 kat_vectors, philox4x32 10 rounds
    assert oracle.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert oracle.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert oracle.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_stream_moments(oracle):
    import ctypes as C
    L = oracle.lib()
    rng, _ = oracle.make_rng(seed=5, stream=2)
    u = np.array([L.orc_rng_uniform(C.byref(rng), 1, 3, i, 0) for i in range(20000)])
    assert 0 <= u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 0.01
    z0, z1 = C.c_double(), C.c_double()
    zs = []
    for i in range(20000):
        L.orc_rng_normal_pair(C.byref(rng), 2, 7, i, 0, C.byref(z0), C.byref(z1))
        zs += [z0.value, z1.value]
    zs = np.array(zs)
    assert abs(zs.mean()) < 0.02 and abs(zs.std() - 1) < 0.02
    assert stats.kstest(zs, "norm").pvalue > 1e-3


def test_packed_block_format_is_what_the_header_says(oracle):
    """PACKED draws (more than two normals): pair p = half p % 2 of block (i, t, p / 2, purpose), 40-bit radius + 24-bit
    angle; the resampling uniform of a filter with such a step draw = word i % 4 of block (i / 4, t, 0, purpose 1).
    Re-derived here from the raw Philox words."""
    import ctypes as C
    L = oracle.lib()
    seed, stream, t = 77, 3, 12
    rng, _ = oracle.make_rng(seed=seed, stream=stream)
    for i in (0, 5, 1023, 65535):
        out = (C.c_double * 7)()
        L.orc_rng_normals(C.byref(rng), 2, t, i, 7, out)
        want = []
        for p in range(4):
            w = oracle.philox([i, t, p // 2, 2], [seed, stream])
            a, b = w[2 * (p % 2)], w[2 * (p % 2) + 1]
            u1 = (((a << 8) | (b >> 24)) + 1) * 2.0 ** -40
            ang = 2 * np.pi * (b & 0xFFFFFF) * 2.0 ** -24
            rad = np.sqrt(-2 * np.log(u1))
            want += [rad * np.cos(ang), rad * np.sin(ang)]
        assert np.allclose(list(out), want[:7], rtol=0, atol=1e-15)
        for cnt in (3, 4, 8):
            assert L.orc_rng_uniform(C.byref(rng), 1, t, i, cnt) == oracle.philox([i >> 2, t, 0, 1], [seed, stream])[i & 3] * 2.0 ** -32
        for cnt in (0, 1, 2):   # WIDE step draw: the spare word of the step block
            assert L.orc_rng_uniform(C.byref(rng), 1, t, i, cnt) == oracle.philox([i, t, 0, 2], [seed, stream])[3] * 2.0 ** -32
        two = (C.c_double * 2)()
        L.orc_rng_normals(C.byref(rng), 2, t, i, 2, two)
        z0, z1 = C.c_double(), C.c_double()
        L.orc_rng_normal_pair(C.byref(rng), 2, t, i, 0, C.byref(z0), C.byref(z1))
        assert list(two) == [z0.value, z1.value]


def test_packed_normals_distribution_1e8(oracle):
    """Quality gate of the PACKED format (round 4: two Box-Muller pairs per Philox block, 40-bit radius, 24-bit angle):
    1e8 normals drawn as the filters draw them (8 per particle-step), against N(0, 1) -- moments, a 4000-cell chi-square
    on equiprobable cells, tail counts out to 6 sigma, and the correlations a packing mistake would show (between the two
    normals of a pair, between the pairs of a block, between blocks, between neighbouring particles)."""
    import ctypes as C
    L = oracle.lib()
    L.orc_rng_normals_bulk.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_double)]
    rng, _ = oracle.make_rng(seed=2024, stream=7)
    cnt, steps, parts = 8, 50, 250000
    z = np.empty((steps, parts, cnt))
    for t in range(steps):
        L.orc_rng_normals_bulk(C.byref(rng), 2, t, 0, parts, cnt, z[t].ctypes.data_as(C.POINTER(C.c_double)))
    flat = z.reshape(-1)
    n = flat.size
    assert n == 100_000_000
    se = 1 / np.sqrt(n)
    assert abs(flat.mean()) < 5 * se
    assert abs(flat.var() - 1) < 5 * np.sqrt(2) * se
    assert abs(np.mean(flat ** 3)) < 5 * np.sqrt(15) * se
    assert abs(np.mean(flat ** 4) - 3) < 5 * np.sqrt(96) * se
    cells = 4000
    edges = stats.norm.ppf(np.arange(1, cells) / cells)
    counts = np.bincount(np.searchsorted(edges, flat), minlength=cells)
    chi2 = float(np.sum((counts - n / cells) ** 2 / (n / cells)))
    assert stats.chi2.sf(chi2, cells - 1) > 1e-4, chi2
    for k in (4.0, 5.0, 6.0):      # two-sided tail counts are Poisson around n * 2 * sf(k)
        exp = n * 2 * stats.norm.sf(k)
        got = int(np.sum(np.abs(flat) > k))
        assert abs(got - exp) <= 5 * np.sqrt(exp) + 3, (k, got, exp)
    assert np.abs(flat).max() < 7.46   # sqrt(2 * 40 * log 2)
    c = lambda a, b: float(np.mean(a * b))
    lim = 5 / np.sqrt(steps * parts)
    assert abs(c(z[..., 0], z[..., 1])) < lim and abs(c(z[..., 2], z[..., 3])) < lim          # inside a pair
    assert abs(c(z[..., 0], z[..., 2])) < lim and abs(c(z[..., 1], z[..., 3])) < lim          # the two pairs of a block
    assert abs(c(z[..., 0], z[..., 4])) < lim and abs(c(z[..., 3], z[..., 7])) < lim          # consecutive blocks
    assert abs(c(z[:, :-1, 0], z[:, 1:, 0])) < lim and abs(c(z[:-1, :, 5], z[1:, :, 5])) < lim  # neighbouring particles / steps
    assert abs(c(z[..., 0] ** 2, z[..., 1] ** 2) - 1) < 5 * 2 / np.sqrt(steps * parts)        # radius shared, angle independent
    # the shared resampling uniforms of the same stream: 24 M words, uniform and uncorrelated with the slot index
    u = np.array([L.orc_rng_uniform(C.byref(rng), 1, 3, i, cnt) for i in range(200000)])
    assert 0 <= u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 5 / np.sqrt(12 * u.size)
    assert abs(np.corrcoef(u[:-1], u[1:])[0, 1]) < 5 / np.sqrt(u.size)
    assert stats.kstest(u, "uniform").pvalue > 1e-4


def test_stratified_sample_hand_case(oracle):
    # src/stratified_sample.cpp:9-28 on a case checkable by hand
    p = np.array([0.1, 0.2, 0.3, 0.4]); r = np.array([0.5, 0.5, 0.5, 0.5])
    idx, cdf = oracle.stratified_sample(p, r)
    # targets (r+j)/4 = .125 .375 .625 .875 ; cdf = .1 .3 .6 1
    assert list(idx) == [1, 2, 3, 3]
    assert cdf[-1] == 1.0
    idx, _ = oracle.stratified_sample(np.array([0, 0, 1.0, 0]), np.array([0.01, 0.99, 0.3, 0.7]))
    assert list(idx) == [2, 2, 2, 2]
    # uniform weights + r = 0.5 => identity
    idx, _ = oracle.stratified_sample(np.full(8, 1 / 8), np.full(8, 0.5))
    assert list(idx) == list(range(8))


def test_distribution_constants_closed_form(oracle):
    # full log-density at signal 0 must equal scipy's: logpmf = kernel(s) + const
    y = np.array([3.0, 0.0, 7.0]); u = np.array([2.0, 1.0, 3.0])
    arr = M.model_arrays(M.ssm_ung(y, Z=1.0, T=1.0, R=1.0, P1=1.0, distribution="poisson", u=u))
    kern = np.sum(y * 0.0 - u * np.exp(0.0))
    assert np.isclose(kern + oracle.const_term(arr), stats.poisson.logpmf(y, u).sum(), rtol=1e-12)
    arr = M.model_arrays(M.ssm_ung(y, Z=1.0, T=1.0, R=1.0, P1=1.0, distribution="negative binomial", phi=2.5, u=u))
    kern = np.sum(y * 0.0 - (y + 2.5) * np.log(2.5 + u))
    ref = stats.nbinom.logpmf(y, 2.5, 2.5 / (2.5 + u)).sum()
    assert np.isclose(kern + oracle.const_term(arr), ref, rtol=1e-12)
    yb = np.array([1.0, 0.0, 4.0]); ub = np.array([3.0, 2.0, 5.0])
    arr = M.model_arrays(M.ssm_ung(yb, Z=1.0, T=1.0, R=1.0, P1=1.0, distribution="binomial", u=ub))
    kern = np.sum(yb * 0.0 - ub * np.log1p(1.0))
    assert np.isclose(kern + oracle.const_term(arr), stats.binom.logpmf(yb, ub, 0.5).sum(), rtol=1e-12)
    yg = np.array([1.3, 0.4, 2.2])
    arr = M.model_arrays(M.ssm_ung(yg, Z=1.0, T=1.0, R=1.0, P1=1.0, distribution="gamma", phi=3.0, u=u))
    kern = np.sum(-3.0 * (0.0 + yg / u))
    assert np.isclose(kern + oracle.const_term(arr), stats.gamma.logpdf(yg, 3.0, scale=u / 3.0).sum(), rtol=1e-12)


def _dense_gaussian_loglik(arr):
    n, m = arr["n"], arr["m"]
    T, R, Z = arr["T"][:, :, 0], arr["R"][:, :, 0], arr["Z"][:, 0]
    a, P = arr["a1"].copy(), arr["P1"].copy()
    means, covs = [], np.zeros((n * m, n * m))
    blocks = []
    for t in range(n):
        means.append(a.copy()); blocks.append(P.copy())
        a = arr["C"][:, 0] + T @ a; P = T @ P @ T.T + R @ R.T
    # joint covariance of states
    for t in range(n):
        covs[t * m:(t + 1) * m, t * m:(t + 1) * m] = blocks[t]
        A = np.eye(m)
        for s in range(t + 1, n):
            A = T @ A
            covs[s * m:(s + 1) * m, t * m:(t + 1) * m] = A @ blocks[t]
            covs[t * m:(t + 1) * m, s * m:(s + 1) * m] = (A @ blocks[t]).T
    Zb = np.kron(np.eye(n), Z.reshape(1, -1))
    mu = Zb @ np.concatenate(means) + arr["D"][0]
    S = Zb @ covs @ Zb.T + np.eye(n) * arr["H"][0] ** 2
    return stats.multivariate_normal.logpdf(arr["y"], mu, S)


def test_kalman_loglik_equals_dense_mvn(oracle):
    rng = np.random.default_rng(0)
    y = rng.normal(size=8)
    model = M.ssm_ulg(y, Z=[1.0, 0.5], H=0.7, T=np.array([[0.9, 0.1], [0.0, 0.8]]), R=np.diag([0.3, 0.2]),
                      a1=[0.2, -0.1], P1=np.diag([0.5, 0.4]), D=0.3, C=np.array([[0.05], [0.0]]))
    arr = M.model_arrays(model)
    assert np.isclose(oracle.ulg_loglik(arr), _dense_gaussian_loglik(arr), rtol=1e-11)


def test_pf_equals_kalman_smoother_on_lgssm(oracle):
    """tests/testthat/test_particle_smoother.R:25-31 (tolerance 1e-2, N = 1e5, seed 1)"""
    model = M.bsm_lg(np.ones(5), sd_level=0.05, sd_slope=0.01, sd_y=0.01, a1=[1, 0], P1=np.diag([0.01, 0.01]))
    arr = M.model_arrays(model)
    at, Pt, _ = oracle.ulg_smoother(arr)
    out = oracle.bsf_filter(arr, 100000, seed=1)
    sm = oracle.filter_smoother(out["alpha"], out["indices"])
    mean, cov = oracle.summary(sm)
    assert np.allclose(mean[:, :5], at[:, :5], atol=1e-2)
    assert np.allclose(cov[:, :, :5], Pt[:, :, :5], atol=1e-2)
    assert abs(out["logLik"] - oracle.ulg_loglik(arr)) < 0.1
    # and the filtered means equal the Kalman filter's within MC error
    fat, fatt, fPt, fPtt = oracle.filter_summary(out["alpha"], out["weights"])
    assert np.all(np.isfinite(fatt)) and np.all(np.isfinite(fPtt))


def test_psi_equals_bsf_within_mc_error(oracle):
    """tests/testthat/test_particle_smoother.R:33-45 recipe on a non-Gaussian model (tol 1e-2 .. MC error)"""
    model = small_zoo()["poisson_ll"]
    arr = M.model_arrays(model)
    n = model.n
    psi = oracle.psi_filter(arr, 2000, seed=1)
    bsf = oracle.bsf_filter(arr, 20000, seed=1)
    m1, _ = oracle.summary(oracle.filter_smoother(psi["alpha"], psi["indices"]))
    m2, _ = oracle.summary(oracle.filter_smoother(bsf["alpha"], bsf["indices"]))
    assert np.max(np.abs(m1[:, :n] - m2[:, :n])) < 0.05
    # both estimate the same likelihood: psi has far lower variance
    lls_psi = [oracle.psi_filter(arr, 100, seed=s)["logLik"] for s in range(20)]
    lls_bsf = [oracle.bsf_filter(arr, 2000, seed=s)["logLik"] for s in range(20)]
    assert abs(np.mean(lls_psi) - np.mean(lls_bsf)) < 4 * (np.std(lls_bsf) / np.sqrt(20) + np.std(lls_psi) / np.sqrt(20)) + 0.05
    assert np.std(lls_psi) < np.std(lls_bsf) * 2


def test_psi_setup_properties(oracle):
    """V_t V_t' + C_t V_{t-1}V_{t-1}' ... : the conditional factors reproduce the smoothing covariances
    (src/conditional_dist.cpp:4-31): Var(a_t) = C_t Var(a_{t-1}) C_t' + V_t V_t'."""
    model = small_zoo()["poisson_slope_missing"]
    arr = M.model_arrays(model)
    su = oracle.psi_setup(arr)
    g = dict(arr); g["y"] = su["approx_y"]; g["H"] = su["approx_H"]
    at, Pt, cc = oracle.ulg_smoother(g)
    assert np.allclose(at, su["alphahat"], atol=1e-12)
    V, Cm = su["Vt"], su["Ct"]
    assert np.allclose(V[:, :, 0] @ V[:, :, 0].T, Pt[:, :, 0], atol=1e-10)
    for t in range(1, arr["n"] + 1):
        recon = Cm[:, :, t] @ Pt[:, :, t - 1] @ Cm[:, :, t].T + V[:, :, t] @ V[:, :, t].T
        assert np.allclose(recon, Pt[:, :, t], atol=1e-9)


REF_TEST_MODELS = {
    # tests/testthat/test_bootstrap_filter.R:48-107
    "poisson": lambda: M.bsm_ng(np.arange(1, 11.0), sd_level=2, sd_slope=2, P1=np.diag([2.0, 2.0]), distribution="poisson"),
    "binomial": lambda: M.ar1_ng(np.array([1, 0, 1, 1, 1, 0, 0, 0.0]), rho=M.uniform(0.9, 0, 1), sigma=M.gamma(1, 2, 2),
                                 mu=M.normal(1, 0, 1), xreg=np.arange(1, 9.0), beta=M.normal(0, 0, 0.1),
                                 distribution="binomial"),
    "negbin": lambda: M.bsm_ng(np.array([1, 0, 1, 1, 1, 0, 0, 0.0]), sd_level=2, sd_slope=2, P1=np.diag([2.0, 2.0]),
                               distribution="negative binomial", phi=0.1, u=2),
    "svm": lambda: M.svm(np.loadtxt(os.path.join(GOLD, "exchange.txt")), rho=M.uniform(0.98, -0.999, 0.999),
                         sd_ar=M.halfnormal(0.2, 5), sigma=M.halfnormal(1, 2)),
    "ulg": lambda: M.ssm_ulg(np.arange(1, 11.0), Z=[1, 0], H=2, T=np.array([[1, 1], [0, 1.0]]), R=np.diag([2.0, 2.0]),
                             a1=[0, 0], P1=np.diag([2.0, 2.0])),
    "ar1_lg": lambda: M.ar1_lg(np.arange(1, 11.0), rho=M.tnormal(0.6, 0, 0.5, -1, 1), sigma=M.gamma(1, 2, 2), sd_y=0.1, mu=1),
}


@pytest.mark.parametrize("name", sorted(REF_TEST_MODELS))
def test_reference_invariants(oracle, name):
    """weights in [0, inf), finite logLik / att / Ptt: tests/testthat/test_bootstrap_filter.R:30-34 etc."""
    arr = M.model_arrays(REF_TEST_MODELS[name]())
    out = oracle.bsf_filter(arr, 10, seed=1)
    assert out["weights"].min() >= 0 and np.isfinite(out["weights"].max())
    assert np.isfinite(out["logLik"])
    at, att, Pt, Ptt = oracle.filter_summary(out["alpha"], out["weights"])
    assert np.isfinite(att.sum()) and np.isfinite(Ptt.sum())
    assert np.all(out["weights"][:, -1] == 1.0)
    assert out["indices"].max() < 10
    again = oracle.bsf_filter(arr, 10, seed=1)           # determinism given (model, N, seed)
    assert again["logLik"] == out["logLik"] and np.array_equal(again["alpha"], out["alpha"])


def test_missing_observation_columns_are_ones(oracle):
    model = small_zoo()["poisson_slope_missing"]
    out = oracle.bsf_filter(M.model_arrays(model), 32, seed=3)
    for t in (3, 4, 17):
        assert np.all(out["weights"][:, t] == 1.0)


def test_injected_randoms_reproduce_philox_mode(oracle):
    """The injected-array mode (reference consumption order) and the counter mode agree when fed the same numbers."""
    import ctypes as C
    model = small_zoo()["negbin_m4"]
    arr = M.model_arrays(model)
    N, n, m, k = 12, arr["n"], arr["m"], arr["k"]
    L = oracle.lib()
    rng, _ = oracle.make_rng(seed=9, stream=1)

    def normals(purpose, t, i, cnt):   # the stream's rule: a draw of <= 2 normals is WIDE, of more PACKED (bssm_oracle.c)
        out = (C.c_double * cnt)()
        L.orc_rng_normals(C.byref(rng), purpose, t, i, cnt, out)
        return list(out)
    init = np.array([normals(0, 0, i, m) for i in range(N)])
    unif = np.array([[L.orc_rng_uniform(C.byref(rng), 1, t, i, k) for i in range(N)] for t in range(n)])  # k = normals of the step draw
    nor = np.array([[normals(2, t, i, k) for i in range(N)] for t in range(n)])
    a = oracle.bsf_filter(arr, N, seed=9, stream=1)
    b = oracle.bsf_filter(arr, N, injected=(init, unif, nor))
    assert a["logLik"] == b["logLik"] and np.array_equal(a["alpha"], b["alpha"])


def test_golden_vectors_regression(oracle):
    gold = np.load(os.path.join(GOLD, "golden_filters.npz"))
    zoo = small_zoo()
    for name in sorted(zoo):
        arr = M.model_arrays(zoo[name])
        out = oracle.bsf_filter(arr, 24, seed=17)
        assert np.array_equal(out["indices"], gold[f"{name}/bsf/indices"])
        assert np.allclose(out["alpha"], gold[f"{name}/bsf/alpha"], rtol=1e-12, atol=1e-13)
        assert np.isclose(out["logLik"], float(gold[f"{name}/bsf/logLik"]), rtol=1e-12)
        if not name.startswith("gaussian"):
            psi = oracle.psi_filter(arr, 24, seed=17)
            assert np.array_equal(psi["indices"], gold[f"{name}/psi/indices"])
            assert np.isclose(psi["logLik"], float(gold[f"{name}/psi/logLik"]), rtol=1e-12)
    from models_zoo import mng_zoo
    for name, mdl in sorted(mng_zoo().items()):
        arr = M.model_arrays(mdl)
        for tag, fn in (("bsf", oracle.mng_bsf_filter), ("psi", oracle.mng_psi_filter)):
            res = fn(arr, 24, seed=17)
            assert np.array_equal(res["indices"], gold[f"mng/{name}/{tag}/indices"])
            assert np.isclose(res["logLik"], float(gold[f"mng/{name}/{tag}/logLik"]), rtol=1e-12)


def test_is_correction_weights_and_merge(oracle):
    """src/approx_mcmc.cpp:447-518: weight_i = exp(loglik_i - approx_loglik_i); summary merge is the weighted mean."""
    base = small_zoo()["poisson_ll"]
    arrs, al = [], []
    for i, sd in enumerate([0.08, 0.1, 0.12, 0.15]):
        mdl = M.bsm_ng(base["y"], sd_level=sd, distribution="poisson")
        arrs.append(M.model_arrays(mdl))
        al.append(oracle.psi_setup(arrs[-1])["approx_loglik"])
    counts = np.array([2, 1, 3, 1], dtype=np.uint32)
    res = oracle.is_correction(arrs, 50, 7, method=2, counts=counts, approx_loglik=np.array(al), output_type=2, n_threads=2)
    for i in range(4):
        ref = oracle.bsf_filter(arrs[i], 50, seed=7, stream=i)
        assert np.isclose(res["loglik"][i], ref["logLik"], rtol=1e-13)
        assert np.isclose(res["weights"][i], np.exp(ref["logLik"] - al[i]), rtol=1e-12)
    # merged alphahat == weight*count weighted mean of per-draw smoothed means
    means = []
    for i in range(4):
        ref = oracle.bsf_filter(arrs[i], 50, seed=7, stream=i)
        mean, _ = oracle.summary(oracle.filter_smoother(ref["alpha"], ref["indices"]))
        means.append(mean)
    w = res["weights"] * counts
    expect = sum(wi * mi for wi, mi in zip(w, means)) / w.sum()
    assert np.allclose(res["alphahat"], expect, rtol=1e-10, atol=1e-12)
    # psi variant: weights = exp(loglik) with approx_loglik = 0 (src/approx_mcmc.cpp:385)
    modes = np.stack([oracle.psi_setup(a)["mode"] for a in arrs], axis=1)
    res2 = oracle.is_correction(arrs, 20, 7, method=1, counts=counts, modes=modes, output_type=1)
    assert np.all(np.isfinite(res2["weights"])) and res2["weights"].max() > 0
    assert res2["alpha"].shape == (base.n + 1, 1, 4) and np.all(np.isfinite(res2["alpha"]))


# ---------------------------------------------------------------- ssm_mng / functor-model parts of the oracle
def test_mlg_loglik_equals_dense_mvn(oracle):
    """multivariate Kalman likelihood (src/model_ssm_mlg.cpp:90-137) vs a dense MVN evaluation"""
    import ctypes as C
    from models_zoo import mng_zoo
    rng = np.random.default_rng(3)
    n, p, m = 6, 2, 2
    Z = rng.normal(size=(p, m)); T = np.array([[0.8, 0.1], [0.0, 0.6]]); R = np.diag([0.4, 0.3])
    y = rng.normal(size=(n, p))
    model = M.ssm_mng(y, Z, T, R, a1=[0.1, -0.2], P1=np.diag([0.5, 0.3]), distribution="gaussian", phi=[0.7, 0.9],
                      D=np.array([[0.2], [-0.1]]), C=np.array([[0.05], [0.0]]))
    arr = M.model_arrays(model)
    mod, keep = oracle.pack_mng(arr)
    gy = np.asfortranarray(y.T); gH = np.asfortranarray(np.tile(np.array([[0.7], [0.9]]), (1, n)))
    L = oracle._mng_lib()
    ll = L.orc_mlg_loglik(C.byref(mod), oracle._p(gy), oracle._p(gH))
    # dense: stack observations
    a, P = arr["a1"].copy(), arr["P1"].copy()
    means, blocks = [], []
    for t in range(n):
        means.append(a.copy()); blocks.append(P.copy())
        a = arr["C"][:, 0] + T @ a; P = T @ P @ T.T + R @ R.T
    cov = np.zeros((n * m, n * m))
    for t in range(n):
        cov[t*m:(t+1)*m, t*m:(t+1)*m] = blocks[t]
        A = np.eye(m)
        for s in range(t + 1, n):
            A = T @ A
            cov[s*m:(s+1)*m, t*m:(t+1)*m] = A @ blocks[t]; cov[t*m:(t+1)*m, s*m:(s+1)*m] = (A @ blocks[t]).T
    Zb = np.kron(np.eye(n), Z)
    mu = Zb @ np.concatenate(means) + np.tile(arr["D"][:, 0], n)
    S = Zb @ cov @ Zb.T + np.kron(np.eye(n), np.diag([0.49, 0.81]))
    assert np.isclose(ll, stats.multivariate_normal.logpdf(y.reshape(-1), mu, S), rtol=1e-11)


def test_mng_gaussian_pf_equals_kalman_and_psi_equals_bsf(oracle):
    """an all-gaussian ssm_mng is an LGSSM: bootstrap log-likelihood = Kalman (analytic pin), and for a mixed
    model the psi-APF and the bootstrap filter estimate the same likelihood (psi with far lower variance)."""
    import ctypes as C
    rng = np.random.default_rng(5)
    n, p, m = 12, 2, 2
    Z = rng.normal(size=(p, m)); T = np.array([[0.8, 0.1], [0.0, 0.6]]); R = np.diag([0.4, 0.3])
    y = rng.normal(size=(n, p)); y[3, 1] = np.nan
    model = M.ssm_mng(y, Z, T, R, a1=[0.0, 0.0], P1=np.diag([0.5, 0.3]), distribution="gaussian", phi=[0.7, 0.9])
    arr = M.model_arrays(model)
    mod, keep = oracle.pack_mng(arr)
    gH = np.asfortranarray(np.tile(np.array([[0.7], [0.9]]), (1, n)))
    kal = oracle._mng_lib().orc_mlg_loglik(C.byref(mod), oracle._p(np.asfortranarray(y.T)), oracle._p(gH))
    lls = [oracle.mng_bsf_filter(arr, 20000, seed=s)["logLik"] for s in range(6)]
    assert abs(np.mean(lls) - kal) < 4 * np.std(lls) / np.sqrt(6) + 0.02
    from models_zoo import mng_zoo
    arr2 = M.model_arrays(mng_zoo()["mixed_m4_p3"])
    su = oracle.mng_psi_setup(arr2)
    lp = [oracle.mng_psi_filter(arr2, 50, seed=s, setup=su)["logLik"] for s in range(16)]
    lb = [oracle.mng_bsf_filter(arr2, 4000, seed=s)["logLik"] for s in range(16)]
    assert abs(np.mean(lp) - np.mean(lb)) < 4 * (np.std(lp) + np.std(lb)) / 4 + 0.05
    assert np.std(lp) < np.std(lb) * 1.5


def test_functor_models_analytic_pins(oracle):
    """nlg_linear_gaussian is an exact LGSSM (PF = Kalman); sde_poisson_ou with tiny diffusion follows the ODE."""
    rng = np.random.default_rng(12)
    y = np.cumsum(rng.normal(0, 1, 30)) + rng.normal(0, 0.5, 30)
    mdl = M.ssm_nlg(y, "nlg_linear_gaussian", theta=[np.log(0.5)])
    kal = oracle.ulg_loglik(M.model_arrays(M.ssm_ulg(y, Z=1.0, H=0.5, T=1.0, R=1.0, a1=0.0, P1=1.0)))
    lls = [oracle.fn_bsf_filter(mdl, 20000, seed=s)["logLik"] for s in range(6)]
    assert abs(np.mean(lls) - kal) < 4 * np.std(lls) / np.sqrt(6) + 0.02
    # OU mean reversion: dx = e^{th0}(th1 - x) dt with sigma -> 0: x(t) = th1 + (x0 - th1) e^{-kappa t}
    ou = M.ssm_sde(np.full(5, np.nan), "sde_poisson_ou", theta=[np.log(0.5), 2.0, np.log(1e-9)], x0=0.0, positive=False)
    out = oracle.fn_bsf_filter({**ou, "L": 10}, 4, seed=1)
    exact = 2.0 + (0.0 - 2.0) * np.exp(-0.5 * np.arange(1, 7))
    assert np.allclose(out["alpha"][0, :, 0], exact, atol=2e-3)   # Euler/Milstein bias O(dt) at 2^-10
    # GBM with positive = True stays positive
    g = M.ssm_sde(rng.normal(0, 1, 10), "sde_gbm", theta=[0.05, 0.3, 1.0], x0=1.0, positive=True)
    assert np.all(oracle.fn_bsf_filter({**g, "L": 3}, 50, seed=2)["alpha"] > 0)


def test_general_nlg_bootstrap_equals_builtin_functor_path(oracle):
    """orc_nlg_general_bsf_filter (function-pointer models, state-dependent H / R allowed) given the shipped growth
    model's functions reproduces the built-in-functor restatement (same ancestors, values to the last places); on a linear-Gaussian model written as
    functions it agrees with the exact Kalman log-likelihood within Monte-Carlo error."""
    rng = np.random.default_rng(3)
    n, K, dT = 30, 500.0, 0.1
    y = 50 + np.cumsum(rng.normal(0.5, 1, n)); y[4] = np.nan
    th = np.array([0.0, np.log(0.05), 0.0]); kp = np.array([dT, K, 0.0, 50.0, 1.0, 100.0])
    model = M.ssm_nlg(y, "nlg_growth", theta=th, known_params=kp)
    ref = oracle.fn_bsf_filter({**model, "L": 0}, 200, seed=7)

    def T(t, a, th_, kn):
        r = np.exp(a[0]) / (1.0 + np.exp(a[0]))
        return np.array([a[0], kn[1] * a[1] * np.exp(r * kn[0]) / (kn[1] + a[1] * (np.exp(r * kn[0]) - 1))])
    got = oracle.nlg_general_bsf_filter(y, 2, 2, th, kp, kp[2:4], np.diag(kp[4:6]), lambda t, a, th_, kn: a[1],
                                        lambda t, a, th_, kn: np.exp(th_[0]), T,
                                        lambda t, a, th_, kn: np.diag([np.exp(th_[1]), np.exp(th_[2])]), 200, seed=7)
    assert np.array_equal(got["indices"], ref["indices"])
    # (numpy's exp and libm's differ in the last place inside T_fn)
    assert abs(got["logLik"] - ref["logLik"]) <= 1e-12 * n and np.allclose(got["alpha"], ref["alpha"], rtol=1e-12, atol=1e-12)
    # linear-Gaussian: local level + slope with a full 2 x 1 matrix R (one noise term drives both states)
    n = 40
    yl = np.cumsum(rng.normal(0.1, 0.3, n)) + rng.normal(0, 0.5, n)
    Tm = np.array([[1.0, 1.0], [0.0, 1.0]]); Rm = np.array([[0.3], [0.05]])
    lls = [oracle.nlg_general_bsf_filter(yl, 2, 1, [0.0], [], [0.0, 0.1], np.diag([1.0, 0.01]), lambda t, a, *_: a[0],
                                         lambda *_: 0.5, lambda t, a, *_: Tm @ a, lambda *_: Rm, 400, seed=s)["logLik"] for s in range(1, 9)]
    arr = M.model_arrays(M.ssm_ulg(yl, Z=np.array([1.0, 0.0]), H=0.5, T=Tm, R=Rm, a1=np.array([0.0, 0.1]), P1=np.diag([1.0, 0.01])))
    exact = oracle.ulg_loglik(arr)
    assert abs(np.mean(lls) - exact) < 4 * np.std(lls) / np.sqrt(len(lls)) + 0.05


def test_tie_following_adopts_only_explained_ties(oracle):
    """oracle.following (the instrument behind the full-size GPU parity tests): following its own table changes nothing;
    a table entry moved to the adjacent particle far from a tie is NOT adopted (counted as unexplained); with a tolerance
    wide enough to call it a tie it is adopted and the run continues along that path (layouts N x n and n x N alike)."""
    from bssm_amd import models as M
    from models_zoo import negbin_m4_informative
    arr = M.model_arrays(negbin_m4_informative(40))
    N = 1024
    base = oracle.bsf_filter(arr, N, seed=3)
    with oracle.following(base["indices"], N) as f:
        again = oracle.bsf_filter(arr, N, seed=3)
    assert (f.ties, f.unexplained) == (0, 0) and again["logLik"] == base["logLik"]
    assert np.array_equal(again["alpha"], base["alpha"])
    moved = base["indices"].copy(order="F")
    moved[100, 20] += 1
    with oracle.following(moved, N) as f:
        kept = oracle.bsf_filter(arr, N, seed=3)
    assert (f.ties, f.unexplained) == (0, 1)
    assert np.array_equal(kept["indices"], base["indices"]) and kept["logLik"] == base["logLik"]
    assert f.tol == oracle.tie_tolerance(N) and 0 < f.tol < 1e-9
    with oracle.following(np.ascontiguousarray(moved.T), N, tol=2.0) as f:     # the device's (n, N) layout
        taken = oracle.bsf_filter(arr, N, seed=3)
    assert f.ties >= 1 and f.first_tie == (20, 100) and taken["indices"][100, 20] == moved[100, 20]
    assert taken["logLik"] != base["logLik"]
    # outside the with-block nothing is followed
    assert oracle.bsf_filter(arr, N, seed=3)["logLik"] == base["logLik"]


def test_mlg_loglik_zero_tolerance_line(oracle):
    """src/model_ssm_mlg.cpp:93 (and model_ssm_ulg.cpp:109): `if(arma::accu(H) + arma::accu(R) < zero_tol) logLik = -inf` -- the sums of the ENTRIES.
    The multivariate restatement lacked the line until round 6; R = -6 I sums to -18 against standard deviations of ~0.6 per series and time point (12 of them)."""
    rng = np.random.default_rng(78)
    n, m = 6, 3
    y = np.column_stack([rng.poisson(3, n), rng.negative_binomial(5, 0.5, n)]).astype(float)
    mk = lambda R: M.model_arrays(M.ssm_mng(y, rng.normal(0, 0.4, (2, m)), 0.9 * np.eye(m), R, a1=np.zeros(m), P1=0.2 * np.eye(m),
                                            distribution=["poisson", "negative binomial"], phi=[1.0, 1.5], u=np.full((n, 2), 1.2)))
    neg, pos = oracle.mng_psi_setup(mk(-6.0 * np.eye(m))), oracle.mng_psi_setup(mk(0.9 * np.eye(m)))
    assert np.isneginf(neg["approx_loglik"]) and np.isfinite(pos["approx_loglik"])
    assert 0.0 < np.sum(neg["approx_H"][np.isfinite(neg["approx_H"])]) < 18.0   # (the line did decide: sum H + sum R < 0 only because of R)

