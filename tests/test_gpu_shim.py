"""The R-side binding EXECUTED: shim/bssm_amd_shim.c compiled against the mock of R's C API (shim/mockr/) and linked with
libbssm_amd.so, its `.Call` entry points run with model lists built field for field as the reference's R constructors build
them (/root/reference/R/models.R:351-357, 1103-1113, 1226-1237, 1393-1399), and every output compared BIT FOR BIT with the
same call through the ctypes path (bssm_amd/filters.py, post_correct.py).  The mock checks what R would: argument counts
against the registration, the PROTECT stack (depth back to zero, nothing used after a collection it was not protected
across -- every allocation is a gctorture collection), Rf_error as a non-local exit.  Real R is still absent; what this pins is
argument order, list-field lookup, dims and layouts of the outputs (src/R_bootstrap_filter.cpp:132-155,
src/R_postcorrection.cpp:226-245), the parset staging through update_fn and the error paths."""
import numpy as np
import pytest

from bssm_amd import filters as F
from bssm_amd import models as M
from bssm_amd import post_correct as PC
from mockr import MockR, RError, r_model_list
from models_zoo import mng_zoo, small_zoo

pytestmark = pytest.mark.gpu

ZOO = small_zoo()


@pytest.fixture(scope="module")
def R():
    r = MockR()
    yield r
    r.reset()


def _after(R):
    assert R.L.mockr_protect_depth() == 0 and R.L.mockr_poisoned_touches() == 0


@pytest.mark.parametrize("name", ["poisson_ll", "poisson_slope_missing", "binomial_ar1_xreg", "negbin_m4", "gamma_ll", "svm", "poisson_tv_m3"])
def test_bsf_and_smoothers_equal_the_ctypes_path(R, name):
    """.Call('_bssm_bsf' / '_bssm_bsf_smoother' / '_bssm_psi_smoother' / '_bssm_nongaussian_loglik')"""
    mdl = ZOO[name]
    lst = r_model_list(mdl)
    if name == "svm":
        lst["u"] = 1.0   # R/models.R:1231
    mt, N, seed, n, m = M.model_type(mdl), 300, 11, mdl.n, mdl.m
    out = R.call("_bssm_bsf", lst, N, seed, False, mt)
    _after(R)
    assert list(out) == ["at", "att", "Pt", "Ptt", "weights", "logLik", "alpha"]     # src/R_bootstrap_filter.cpp:151-155
    assert out["at"].shape == (n + 1, m) and out["att"].shape == (n, m) and out["Pt"].shape == (m, m, n + 1)
    assert out["Ptt"].shape == (m, m, n) and out["weights"].shape == (N, n + 1) and out["alpha"].shape == (m, n + 1, N)
    ref = F.bootstrap_filter(mdl, N, seed=seed)
    for k in ("at", "att", "Pt", "Ptt", "weights"):
        assert np.array_equal(out[k], ref[k]), k
    assert out["logLik"][0] == ref["logLik"]
    assert np.array_equal(out["alpha"], np.transpose(ref["alpha"], (1, 0, 2)))      # R then does aperm(alpha, c(2, 1, 3))
    for call, method in (("_bssm_bsf_smoother", "bsf"), ("_bssm_psi_smoother", "psi")):
        sm = R.call(call, lst, N, seed, False, mt) if method == "bsf" else R.call(call, lst, N, seed, mt)
        _after(R)
        assert list(sm) == ["alphahat", "Vt", "weights", "logLik", "alpha"]           # :374-377
        rs = F.particle_smoother(mdl, N, method=method, seed=seed)
        assert sm["alphahat"].shape == (n + 1, m) and sm["Vt"].shape == (m, m, n + 1)
        assert np.array_equal(sm["alphahat"][:n], rs["alphahat"]) and np.array_equal(sm["Vt"][:, :, :n], rs["Vt"])
        assert np.array_equal(sm["weights"], rs["weights"]) and sm["logLik"][0] == rs["logLik"]
        assert np.array_equal(sm["alpha"][:, :n, :], np.transpose(rs["alpha"], (1, 0, 2)))   # R drops the (n+1)-th point
        ll = R.call("_bssm_nongaussian_loglik", lst, N, 1 if method == "psi" else 2, seed, mt)
        assert ll[0] == F.logLik(mdl, N, method=method, seed=seed)
    _after(R)


def test_gaussian_and_multivariate_models_through_the_shim(R):
    for name, zoo in (("gaussian_ar1", ZOO), ("gaussian_bsm", ZOO)):
        mdl = zoo[name]
        out = R.call("_bssm_bsf", r_model_list(mdl), 200, 3, True, M.model_type(mdl))
        ref = F.bootstrap_filter(mdl, 200, seed=3)
        assert np.array_equal(out["att"], ref["att"]) and out["logLik"][0] == ref["logLik"]
    for name, mdl in mng_zoo().items():
        lst = r_model_list(mdl)
        out = R.call("_bssm_bsf", lst, 150, 5, False, M.model_type(mdl))
        ref = F.bootstrap_filter(mdl, 150, seed=5)
        assert np.array_equal(out["weights"], ref["weights"]) and out["logLik"][0] == ref["logLik"], name
        assert np.array_equal(out["alpha"], np.transpose(ref["alpha"], (1, 0, 2))), name
        sm = R.call("_bssm_psi_smoother", lst, 150, 5, M.model_type(mdl))
        rs = F.particle_smoother(mdl, 150, method="psi", seed=5)
        assert np.array_equal(sm["alphahat"][:mdl.n], rs["alphahat"]) and sm["logLik"][0] == rs["logLik"], name
    _after(R)


def _mcmc_output(mdl, ns, output_type, seed=3):
    rng = np.random.default_rng(seed)
    th0 = np.asarray(mdl["theta"], dtype=float)
    theta = th0[None, :] * np.exp(rng.normal(0, 0.05, (ns, th0.size))) if th0.size else np.zeros((ns, 0))
    mode = F.psi_setup(mdl)["mode"].reshape(-1, order="F")
    return dict(theta=theta, counts=rng.integers(1, 4, ns).astype(np.uint32), modes=np.repeat(mode[:, None], ns, axis=1),
                posterior=np.zeros(ns), output_type=output_type)


@pytest.mark.parametrize("output_type", [1, 2, 3])
@pytest.mark.parametrize("is_type", [1, 2, 3])
def test_postcorrection_nongaussian_equals_post_correct(R, output_type, is_type):
    """.Call('_bssm_postcorrection_nongaussian', model_, model_type, output_type, nsim, seed, n_threads, is_type, counts,
    theta, modes) as R/post_correction.R:266-270 calls it (theta transposed to n_par x n_stored, modes p x n x n_stored),
    built-in update_model (bsm_ng): weights, posterior and the state outputs of src/R_postcorrection.cpp:226-245"""
    y = np.asarray(ZOO["poisson_slope_missing"]["y"])
    mdl = M.bsm_ng(y, sd_level=M.halfnormal(0.3, 1), sd_slope=M.halfnormal(0.05, 0.5), distribution="poisson", P1=np.diag([1.0, 0.1]), a1=[1.0, 0.0])
    ns, nsim, seed = 9, 40, 21
    mo = _mcmc_output(mdl, ns, output_type)
    ref = PC.post_correct(mdl, mo, nsim, threads=2, is_type="is%d" % is_type, seed=seed, devices="all")
    n, m = mdl.n, mdl.m
    out = R.call("_bssm_postcorrection_nongaussian", r_model_list(mdl), M.model_type(mdl), output_type, nsim, seed, 2, is_type,
                 mo["counts"].astype(np.float64), mo["theta"].T, mo["modes"].reshape(1, n, ns, order="F"))
    _after(R)
    want = {1: ["alpha", "weights", "posterior"], 2: ["alphahat", "Vt", "weights", "posterior"], 3: ["weights", "posterior"]}[output_type]
    assert list(out) == want
    assert np.array_equal(out["weights"], ref["weights"])
    assert np.array_equal(out["posterior"], ref["posterior"])
    if output_type == 1:
        assert out["alpha"].shape == (n + 1, m, ref["weights"].size) and np.array_equal(out["alpha"], ref["alpha"])
    if output_type == 2:
        assert out["alphahat"].shape == (n + 1, m) and out["Vt"].shape == (m, m, n + 1)
        assert np.array_equal(out["alphahat"], ref["alphahat"]) and np.array_equal(out["Vt"], ref["Vt"])


def test_postcorrection_through_the_r_closure_and_suggest_n(R):
    """ssm_ung with update_fn / prior_fn closures (parset staging, src/parset_ng.cpp:5-72): the shim evaluates the R closure
    once per stored theta -- here a Python function behind the mock's function object -- and copies the components it returns"""
    base = ZOO["negbin_m4"]
    R0, phi0 = np.array(base["R"], dtype=float).reshape(4, 4), float(base["phi"])
    calls = []

    def update_fn(theta):
        calls.append(np.array(theta))
        return dict(R=(R0 * np.exp(theta[0])).reshape(4, 4, 1), phi=phi0 * np.exp(theta[1]))

    mdl = M.BssmModel(base.classes, **{**base, "update_fn": update_fn, "prior_fn": (lambda th: -0.5 * float(th @ th)), "theta": np.array([0.1, -0.2])})
    ns, nsim, seed = 6, 32, 8
    rng = np.random.default_rng(1)
    theta = rng.normal(0, 0.1, (ns, 2))
    mode = F.psi_setup(mdl)["mode"].reshape(-1, order="F")
    mo = dict(theta=theta, counts=np.ones(ns, dtype=np.uint32), modes=np.repeat(mode[:, None], ns, axis=1), posterior=np.zeros(ns), output_type=3)
    ref = PC.post_correct(mdl, mo, nsim, threads=1, is_type="is2", seed=seed, devices="all")
    calls.clear()
    out = R.call("_bssm_postcorrection_nongaussian", r_model_list(mdl), M.model_type(mdl), 3, nsim, seed, 1, 2,
                 np.ones(ns), theta.T, mo["modes"].reshape(1, mdl.n, ns, order="F"))
    _after(R)
    assert len(calls) == ns and all(np.array_equal(c, t) for c, t in zip(calls, theta))   # serial, in draw order, before any filtering
    assert np.array_equal(out["weights"], ref["weights"]) and np.array_equal(out["posterior"], ref["posterior"])
    # a closure that returns a component of the wrong length is an R error, with the protect stack unwound
    bad = M.BssmModel(base.classes, **{**mdl, "update_fn": (lambda th: dict(R=np.zeros(5)))})
    with pytest.raises(RError, match="update_fn returned 'R' of length 5, expected 16"):
        R.call("_bssm_postcorrection_nongaussian", r_model_list(bad), M.model_type(bad), 3, nsim, seed, 1, 2, np.ones(ns), theta.T,
               mo["modes"].reshape(1, mdl.n, ns, order="F"))
    boom = M.BssmModel(base.classes, **{**mdl, "update_fn": (lambda th: 1 / 0)})
    with pytest.raises(RError, match="ZeroDivisionError"):
        R.call("_bssm_postcorrection_nongaussian", r_model_list(boom), M.model_type(boom), 3, nsim, seed, 1, 2, np.ones(ns), theta.T,
               mo["modes"].reshape(1, mdl.n, ns, order="F"))
    assert R.L.mockr_protect_depth() == 0
    # .Call('_bssm_suggest_n_nongaussian', model_, theta, candidates, replications, seed, model_type): R/post_correction.R:118
    cand = [10, 20, 40]
    sds = R.call("_bssm_suggest_n_nongaussian", r_model_list(mdl), theta[0], np.asarray(cand, dtype=np.float64), 12, 5, M.model_type(mdl))
    _after(R)
    want = PC.suggest_N(mdl, theta[0], candidates=cand, replications=12, seed=5, devices="all")
    assert np.array_equal(sds, want["results"]["sd"])


def test_stratified_sample_and_the_functor_models_through_the_shim(R):
    rng = np.random.default_rng(0)
    p = rng.random(50); p /= p.sum()
    r = rng.random(20)
    got = R.call("_bssm_stratified_sample", p, r, 20)
    _after(R)
    assert np.array_equal(got, F.stratified_sample(p, r, 20)[0].astype(float))          # arma::uvec wraps to numeric
    # ssm_nlg: the 16 leading arguments of every stub (src/R_bootstrap_filter.cpp:461-466); the model identity rides on the first
    # function pointer; the reference hands t(y)
    n = 40
    y = np.cumsum(rng.normal(0, 0.3, n)) + 50.0
    th = np.array([0.0, np.log(0.05), 0.0])
    kn = np.array([0.1, 500.0, 0.0, 50.0, 1.0, 100.0])
    mdl = M.ssm_nlg(y, "nlg_growth", theta=th, known_params=kn)
    Z = R.extptr(bssm_amd_model_id=np.asarray([mdl["model_id"]], dtype=np.int32))
    ptr = R.extptr
    args = [y.reshape(1, n), Z, ptr(), ptr(), ptr(), ptr(), ptr(), ptr(), ptr(), th, ptr(), kn, np.full((1, 1), np.nan), 2, 2, 0]
    out = R.call("_bssm_bsf_nlg", *args, 100, 4)
    _after(R)
    ref = F.bootstrap_filter(mdl, 100, seed=4)
    assert np.array_equal(out["weights"], ref["weights"]) and out["logLik"][0] == ref["logLik"]
    assert np.array_equal(out["alpha"], np.transpose(ref["alpha"], (1, 0, 2)))
    sm = R.call("_bssm_psi_smoother_nlg", *args, 100, 4, 100, 1e-8, 0)
    rs = F.particle_smoother(mdl, 100, method="psi", seed=4)
    assert np.array_equal(sm["alphahat"][:n], rs["alphahat"]) and sm["logLik"][0] == rs["logLik"]
    with pytest.raises(RError, match="bssm_amd_model_id"):       # a pointer without the identity attributes
        R.call("_bssm_bsf_nlg", *([args[0], ptr()] + args[2:]), 100, 4)
    assert R.L.mockr_protect_depth() == 0


def test_degenerate_filter_warns_like_the_reference(R):
    """weights that underflow to zero: logLik = -Inf and Rcpp::warning (src/R_bootstrap_filter.cpp:140-141), not an error"""
    rng = np.random.default_rng(2)
    n = 12
    D = np.zeros(n); D[5] = 800.0  # exp(signal) overflows at t = 5: every log-weight is -Inf there
    mdl = M.ssm_ung(rng.poisson(3.0, n).astype(float), Z=1.0, T=1.0, R=0.3, a1=0.0, P1=1.0, distribution="poisson", D=D)
    out = R.call("_bssm_bsf", r_model_list(mdl), 20, 1, False, M.model_type(mdl))
    _after(R)
    assert out["logLik"][0] == -np.inf
    assert "stopped prematurely" in R.last_warning()


# ---------------------------------------------------------------- the remaining registered entry points
def _nlg_args(R, mdl, theta=None, prior=None):
    """the 16 leading arguments of every ssm_nlg stub (R/bootstrap_filter.R:150-160 passes t(model$y), the nine pointers, theta,
    known_params, known_tv_params, n_states, n_etas, as.integer(time_varying)); identity attributes on the first pointer"""
    n = np.asarray(mdl["y"]).size
    Z = R.extptr(bssm_amd_model_id=np.asarray([mdl["model_id"]], dtype=np.int32))
    lp = R.extptr(bssm_amd_prior_fn=prior) if prior is not None else R.extptr()
    m = int(mdl.get("n_states", 1))
    return [np.asarray(mdl["y"], dtype=float).reshape(1, n), Z, R.extptr(), R.extptr(), R.extptr(), R.extptr(), R.extptr(), R.extptr(),
            R.extptr(), np.asarray(mdl["theta"] if theta is None else theta, dtype=float), lp,
            np.asarray(mdl.get("known_params", np.zeros(0)), dtype=float), np.full((1, 1), np.nan), m, m, np.zeros(4, dtype=np.int32)]


def _growth(n=60, seed=3):
    rng = np.random.default_rng(seed)
    y = 50.0 + np.cumsum(rng.normal(0.5, 1.0, n))
    return M.ssm_nlg(y, "nlg_growth", theta=np.array([0.0, np.log(0.05), 0.0]), known_params=[0.1, 500.0, 0.0, 50.0, 1.0, 100.0])


def test_gaussian_psi_smoother_ekpf_and_nonlinear_loglik_through_the_shim(R):
    mdl = ZOO["gaussian_bsm"]
    cube = R.call("_bssm_gaussian_psi_smoother", r_model_list(mdl), 64, 9, M.model_type(mdl))
    _after(R)
    ref = F.particle_smoother(mdl, 64, method="psi", seed=9)
    assert cube.shape == (mdl.m, mdl.n + 1, 64)                                  # src/R_psi.cpp:14-50: the cube m x (n+1) x nsim
    ra = np.transpose(ref["alpha"], (1, 0, 2))
    assert np.array_equal(cube[:, :ra.shape[1], :], ra)
    g = _growth()
    args = _nlg_args(R, g)
    n = g.n
    for call, method in (("_bssm_ekpf", None), ("_bssm_ekpf_smoother", "ekf"), ("_bssm_bsf_smoother_nlg", "bsf")):
        out = R.call(call, *args, 80, 6)
        _after(R)
        if method is None:
            import bssm_amd.filters as FF
            ref = FF._functor_filter(g, 80, 6, False, None, False, ekpf=True)
            assert np.array_equal(out["att"], ref["att"]) and out["logLik"][0] == ref["logLik"]
        else:
            ref = F.particle_smoother(g, 80, method=method, seed=6)
            assert np.array_equal(out["alphahat"][:n], ref["alphahat"]) and out["logLik"][0] == ref["logLik"]
    # .Call('_bssm_nonlinear_loglik', <16>, nsim, seed, max_iter, conv_tol, iekf_iter, method): R/loglik.R:131
    for method, name in ((1, "psi"), (2, "bsf")):
        ll = R.call("_bssm_nonlinear_loglik", *args, 120, 4, 100, 1e-8, 0, method)
        assert ll[0] == F.logLik(g, 120, method=name, seed=4)
    ll0 = R.call("_bssm_nonlinear_loglik", *args, 0, 4, 100, 1e-8, 0, 1)            # nsim = 0: the approximate log-likelihood
    assert ll0[0] == F.psi_setup(g)["approx_loglik"]
    _after(R)


@pytest.mark.parametrize("output_type", [1, 2, 3])
def test_postcorrection_and_suggest_n_nonlinear_through_the_shim(R, output_type):
    g = _growth()
    ns, nsim, seed = 7, 64, 12
    rng = np.random.default_rng(4)
    theta = np.asarray(g["theta"])[None, :] + rng.normal(0, 0.05, (ns, 3))
    mode = F.psi_setup(g)["mode"].reshape(-1, order="F")
    mo = dict(theta=theta, counts=rng.integers(1, 3, ns).astype(np.uint32), modes=np.repeat(mode[:, None], ns, axis=1),
              posterior=np.zeros(ns), output_type=output_type)
    ref = PC.post_correct(g, mo, nsim, threads=1, is_type="is2", seed=seed, devices="all")
    out = R.call("_bssm_postcorrection_nonlinear", *_nlg_args(R, g), output_type, nsim, seed, 1, 2, mo["counts"].astype(float), theta.T,
                 mo["modes"].reshape(2, g.n, ns, order="F"))
    _after(R)
    assert np.array_equal(out["weights"], ref["weights"]) and np.array_equal(out["posterior"], ref["posterior"])
    if output_type == 1:
        assert np.array_equal(out["alpha"], ref["alpha"])
    if output_type == 2:
        assert np.array_equal(out["alphahat"], ref["alphahat"]) and np.array_equal(out["Vt"], ref["Vt"])
    if output_type == 3:
        cand = [16, 32]
        sds = R.call("_bssm_suggest_n_nonlinear", *_nlg_args(R, g), theta[0], np.asarray(cand, dtype=float), 10, 3)
        want = PC.suggest_N(g, theta[0], candidates=cand, replications=10, seed=3, devices="all")
        assert np.array_equal(sds, want["results"]["sd"])
        _after(R)


def test_sde_entry_points_through_the_shim(R):
    """ssm_sde: y, x0, positive, the five pointers (identity on `drift`), theta, nsim, L, seed (src/R_sde.cpp:10-130)"""
    rng = np.random.default_rng(8)
    n = 30
    y = np.exp(np.cumsum(rng.normal(0.01, 0.05, n)))
    mdl = M.ssm_sde(y, "sde_gbm", theta=np.array([0.02, 0.2, 0.1]), x0=1.0, positive=True)
    drift = R.extptr(bssm_amd_model_id=np.asarray([mdl["model_id"]], dtype=np.int32))
    args = [y, 1.0, True, drift, R.extptr(), R.extptr(), R.extptr(), R.extptr(), np.asarray(mdl["theta"], dtype=float)]
    out = R.call("_bssm_bsf_sde", *args, 100, 3, 5)
    _after(R)
    ref = F.bootstrap_filter(mdl, 100, seed=5, L=3)
    assert np.array_equal(out["weights"], ref["weights"]) and out["logLik"][0] == ref["logLik"]
    assert np.array_equal(out["alpha"], np.transpose(ref["alpha"], (1, 0, 2)))
    sm = R.call("_bssm_bsf_smoother_sde", *args, 100, 3, 5)
    rs = F.particle_smoother(mdl, 100, method="bsf", seed=5, L=3)
    assert np.array_equal(sm["alphahat"][:n], rs["alphahat"]) and sm["logLik"][0] == rs["logLik"]
    ll = R.call("_bssm_loglik_sde", *args, 100, 3, 5)
    assert ll[0] == F.logLik(mdl, 100, method="bsf", seed=5, L=3)
    _after(R)


def test_nongaussian_is_mcmc_through_the_shim(R):
    """.Call('_bssm_nongaussian_is_mcmc', model_, output_type, nsim, iter, burnin, thin, gamma, target_acceptance, S, seed,
    end_ram, n_threads, sampling_method, is_type, model_type, approx, verbose) as R/run_mcmc.R:497-522 calls it -- the approximate
    chain on the host, then the psi-APF correction on the device -- against run_mcmc() of the Python mirror (same library
    calls, same seeds): theta on the sampling scale, counts, weights, posterior, the adapted S, the acceptance rate"""
    from bssm_amd import mcmc
    y = np.asarray(ZOO["poisson_slope_missing"]["y"])
    mdl = M.bsm_ng(y, sd_level=M.halfnormal(0.3, 1), sd_slope=M.halfnormal(0.05, 0.5), distribution="poisson", P1=np.diag([1.0, 0.1]), a1=[1.0, 0.0])
    it, burn, nsim, seed = 400, 200, 16, 77
    ref = mcmc.run_mcmc(mdl, iter=it, particles=nsim, output_type="theta", mcmc_type="is2", sampling_method="psi", burnin=burn, seed=seed)
    lst = r_model_list(mdl)
    th0 = np.log(np.maximum(0.001, np.asarray(mdl["theta"], dtype=float)))            # R/run_mcmc.R:471-489: sds on the log scale
    lst["theta"] = th0
    S0 = np.diag(0.1 * np.maximum(0.1, np.abs(th0)))
    out = R.call("_bssm_nongaussian_is_mcmc", lst, 3, nsim, it, burn, 1, 2 / 3, 0.234, S0, seed, False, 1, 1, 2, M.model_type(mdl), False, False)
    _after(R)
    assert list(out) == ["theta", "weights", "counts", "acceptance_rate", "S", "posterior", "modes"]      # src/R_mcmc.cpp:440-469
    assert np.array_equal(np.exp(out["theta"]), ref["theta"])
    assert np.array_equal(out["counts"], ref["counts"].astype(float))
    assert np.array_equal(out["weights"], ref["weights"]) and np.array_equal(out["posterior"], ref["posterior"])
    assert out["acceptance_rate"][0] == ref["acceptance_rate"] and np.array_equal(np.tril(out["S"]), np.tril(ref["S"]))
    assert out["modes"].shape == (1, mdl.n, out["counts"].size)


def test_nonlinear_and_sde_is_mcmc_and_the_sde_state_sampler_through_the_shim(R):
    """The last three registered stubs.  .Call('_bssm_nonlinear_is_mcmc', <34>) (src/R_mcmc.cpp:736-832),
    .Call('_bssm_sde_is_mcmc', <24>) (src/R_sde.cpp:221-290) and .Call('_bssm_sde_state_sampler_bsf_is2', <13>)
    (src/R_sde_state_sampler.cpp:10-50) against run_mcmc / is_correction of the Python mirror: the same library calls with the
    same seeds, so theta, counts, weights, posterior, S and the acceptance rate must be equal bit for bit.  The log prior is
    an R function here (attr(log_prior_pdf, 'bssm_amd_prior_fn'): INTEGRATION.md) -- a Python callable behind the mock."""
    from bssm_amd import mcmc
    g = _growth(n=40)
    prior = M.example_log_prior("nlg_growth")
    it, burn, nsim, seed = 300, 150, 16, 31
    ref = mcmc.run_mcmc(g, iter=it, particles=nsim, output_type="theta", mcmc_type="is2", sampling_method="psi", burnin=burn, seed=seed)
    th0 = np.asarray(g["theta"], dtype=float)
    S0 = np.diag(0.1 * np.maximum(0.1, np.abs(th0)))
    a = _nlg_args(R, g, prior=lambda th: float(prior(th)))
    # (y, Z, H, T, R, Zg, Tg, a1, P1, theta, log_prior_pdf, known_params, known_tv_params, time_varying, n_states, n_etas, seed, nsim, iter,
    #  burnin, thin, gamma, target_acceptance, S, end_ram, n_threads, is_type, sampling_method, max_iter, conv_tol, iekf_iter, output_type,
    #  approx, verbose)
    out = R.call("_bssm_nonlinear_is_mcmc", *a[:13], a[15], a[13], a[14], seed, nsim, it, burn, 1, 2 / 3, 0.234, S0, False, 1, 2, 1, 100,
                 1e-8, 0, 3, False, False)
    _after(R)
    assert list(out) == ["theta", "weights", "counts", "acceptance_rate", "S", "posterior", "modes"]
    assert np.array_equal(out["theta"], ref["theta"]) and np.array_equal(out["counts"], ref["counts"].astype(float))
    assert np.array_equal(out["weights"], ref["weights"]) and np.array_equal(out["posterior"], ref["posterior"])
    assert out["acceptance_rate"][0] == ref["acceptance_rate"] and np.array_equal(np.tril(out["S"]), np.tril(ref["S"]))
    assert out["modes"].shape == (2, g.n, out["counts"].size)

    rng = np.random.default_rng(8)
    n = 24
    y = np.exp(np.cumsum(rng.normal(0.01, 0.05, n)))
    sde = M.ssm_sde(y, "sde_gbm", theta=np.array([0.02, 0.2, 0.6]), x0=1.0, positive=True)
    sprior = M.example_log_prior("sde_gbm")
    drift = R.extptr(bssm_amd_model_id=np.asarray([sde["model_id"]], dtype=np.int32))
    lp = R.extptr(bssm_amd_prior_fn=lambda th: float(sprior(th)))
    sargs = [y, 1.0, True, drift, R.extptr(), R.extptr(), lp, R.extptr()]
    th0 = np.asarray(sde["theta"], dtype=float)
    S0 = np.diag(0.1 * np.maximum(0.1, np.abs(th0)))
    it, burn, nsim, seed = 200, 100, 32, 5
    ref = mcmc.run_mcmc(sde, iter=it, particles=nsim, output_type="theta", mcmc_type="is2", burnin=burn, seed=seed, L_c=2, L_f=4)
    # (<8>, theta, nsim, L_c, L_f, seed, iter, burnin, thin, gamma, target_acceptance, S, end_ram, is_type, n_threads, type, verbose)
    out = R.call("_bssm_sde_is_mcmc", *sargs, th0, nsim, 2, 4, seed, it, burn, 1, 2 / 3, 0.234, S0, False, 2, 1, 3, False)
    _after(R)
    assert list(out) == ["theta", "weights", "counts", "acceptance_rate", "S", "posterior"]          # no `modes` (src/R_sde.cpp:262-285)
    assert np.array_equal(out["theta"], ref["theta"]) and np.array_equal(out["counts"], ref["counts"].astype(float))
    assert np.array_equal(out["weights"], ref["weights"]) and np.array_equal(out["posterior"], ref["posterior"])
    assert out["acceptance_rate"][0] == ref["acceptance_rate"]
    # the state sampler: bootstrap IS2 correction with one trajectory per stored theta
    ns = ref["theta"].shape[0]
    fine = M.BssmModel(sde.classes, **{**sde, "L": 4})
    r = PC.is_correction(fine, ref["theta"], np.ones(ns, dtype=np.uint32), nsim, method="bsf", approx_loglik=ref["approx_loglik"], is_type=2,
                         output_type=1, seed=seed, threads=1)
    # (<8>, nsim, L_f, seed, approx_loglik_storage, theta n_par x n_stored)
    st = R.call("_bssm_sde_state_sampler_bsf_is2", *sargs, nsim, 4, seed, ref["approx_loglik"], ref["theta"].T)
    _after(R)
    assert list(st) == ["alpha", "weights"]
    assert np.array_equal(st["weights"], r["weights"]) and np.array_equal(st["alpha"], r["alpha"])


# ---------------------------------------------------------------- round 6: the pseudo-marginal / delayed-acceptance chains and R_milstein
def _chain_equal(out, ref, output_type, log_theta=False):
    want = {1: ["alpha", "theta", "counts", "acceptance_rate", "S", "posterior"],
            2: ["alphahat", "Vt", "theta", "counts", "acceptance_rate", "S", "posterior"],
            3: ["theta", "counts", "acceptance_rate", "S", "posterior"]}[output_type]
    assert list(out) == want                                                        # src/R_mcmc.cpp:207-232 (no weights / modes)
    assert np.array_equal(np.exp(out["theta"]) if log_theta else out["theta"], ref["theta"])
    assert np.array_equal(out["counts"], ref["counts"].astype(float)) and np.array_equal(out["posterior"], ref["posterior"])
    assert out["acceptance_rate"][0] == ref["acceptance_rate"] and np.array_equal(np.tril(out["S"]), np.tril(ref["S"]))
    if output_type == 1:
        assert out["alpha"].shape == ref["alpha"].shape and np.array_equal(out["alpha"], ref["alpha"])
    if output_type == 2:
        assert np.array_equal(out["alphahat"], ref["alphahat"]) and np.array_equal(out["Vt"], ref["Vt"])


@pytest.mark.parametrize("mcmc_type", ["pm", "da"])
@pytest.mark.parametrize("output_type,method", [(1, "psi"), (2, "bsf"), (3, "psi")])
def test_nongaussian_pm_and_da_mcmc_through_the_shim(R, mcmc_type, output_type, method):
    """.Call('_bssm_nongaussian_pm_mcmc' / '_bssm_nongaussian_da_mcmc', model_, output_type, nsim, iter, burnin, thin, gamma,
    target_acceptance, S, seed, end_ram, n_threads, sampling_method, model_type, verbose) as R/run_mcmc.R:497-522 calls them
    (src/R_mcmc.cpp:167-327): the device-resident plan filters once per iteration.  Against run_mcmc() of the Python mirror -- the
    same bssm_run_mcmc call with the same seed: theta, counts, posterior, S, acceptance rate and the state outputs bit for bit."""
    from bssm_amd import mcmc
    y = np.asarray(ZOO["poisson_slope_missing"]["y"])
    mdl = M.bsm_ng(y, sd_level=M.halfnormal(0.3, 1), sd_slope=M.halfnormal(0.05, 0.5), distribution="poisson", P1=np.diag([1.0, 0.1]), a1=[1.0, 0.0])
    it, burn, nsim, seed = 120, 60, 32, 41
    ot = {1: "full", 2: "summary", 3: "theta"}[output_type]
    ref = mcmc.run_mcmc(mdl, iter=it, particles=nsim, output_type=ot, mcmc_type=mcmc_type, sampling_method=method, burnin=burn, seed=seed)
    lst = r_model_list(mdl)
    th0 = np.log(np.maximum(0.001, np.asarray(mdl["theta"], dtype=float)))            # R/run_mcmc.R:471-489
    lst["theta"] = th0
    S0 = np.diag(0.1 * np.maximum(0.1, np.abs(th0)))
    out = R.call("_bssm_nongaussian_%s_mcmc" % mcmc_type, lst, output_type, nsim, it, burn, 1, 2 / 3, 0.234, S0, seed, False, 1,
                 1 if method == "psi" else 2, M.model_type(mdl), False)
    _after(R)
    _chain_equal(out, ref, output_type, log_theta=True)


def test_ssm_ung_closures_in_the_pm_chain_through_the_shim(R):
    """ssm_ung with update_fn / prior_fn (model_type 1, src/R_mcmc.cpp:195-199): the chain calls the R closures once per
    iteration -- Python functions behind the mock -- through bssm_run_mcmc's callbacks"""
    from bssm_amd import mcmc
    base = ZOO["negbin_m4"]
    R0, phi0 = np.array(base["R"], dtype=float).reshape(4, 4), float(base["phi"])
    mdl = M.BssmModel(base.classes, **{**base, "update_fn": (lambda th: dict(R=(R0 * np.exp(th[0])).reshape(4, 4, 1), phi=phi0 * np.exp(th[1]))),
                                       "prior_fn": (lambda th: -0.5 * float(th @ th)), "theta": np.array([0.1, -0.2])})
    it, burn, nsim, seed = 60, 30, 32, 9
    ref = mcmc.run_mcmc(mdl, iter=it, particles=nsim, output_type="theta", mcmc_type="pm", sampling_method="bsf", burnin=burn, seed=seed)
    th0 = np.asarray(mdl["theta"], dtype=float)
    S0 = np.diag(0.1 * np.maximum(0.1, np.abs(th0)))
    out = R.call("_bssm_nongaussian_pm_mcmc", r_model_list(mdl), 3, nsim, it, burn, 1, 2 / 3, 0.234, S0, seed, False, 1, 2, M.model_type(mdl), False)
    _after(R)
    _chain_equal(out, ref, 3)


@pytest.mark.parametrize("mcmc_type,method,output_type", [("pm", "psi", 1), ("pm", "bsf", 2), ("pm", "ekf", 3), ("da", "psi", 3), ("da", "bsf", 1)])
def test_nonlinear_pm_and_da_mcmc_through_the_shim(R, mcmc_type, method, output_type):
    """.Call('_bssm_nonlinear_pm_mcmc' / '_da_mcmc', y, Z, H, T, R, Zg, Tg, a1, P1, theta, log_prior_pdf, known_params,
    known_tv_params, time_varying, n_states, n_etas, seed, nsim, iter, burnin, thin, gamma, target_acceptance, S, end_ram,
    n_threads, max_iter, conv_tol, sampling_method, iekf_iter, output_type, verbose): src/R_mcmc.cpp:550-668"""
    from bssm_amd import mcmc
    g = _growth(n=30)
    prior = M.example_log_prior("nlg_growth")
    it, burn, nsim, seed = 80, 40, 32, 17
    ot = {1: "full", 2: "summary", 3: "theta"}[output_type]
    ref = mcmc.run_mcmc(g, iter=it, particles=nsim, output_type=ot, mcmc_type=mcmc_type, sampling_method=method, burnin=burn, seed=seed)
    th0 = np.asarray(g["theta"], dtype=float)
    S0 = np.diag(0.1 * np.maximum(0.1, np.abs(th0)))
    a = _nlg_args(R, g, prior=lambda th: float(prior(th)))
    out = R.call("_bssm_nonlinear_%s_mcmc" % mcmc_type, *a[:13], a[15], a[13], a[14], seed, nsim, it, burn, 1, 2 / 3, 0.234, S0, False, 1,
                 100, 1e-8, {"psi": 1, "bsf": 2, "ekf": 4}[method], 0, output_type, False)
    _after(R)
    _chain_equal(out, ref, output_type)


def test_sde_pm_da_mcmc_and_milstein_through_the_shim(R):
    """.Call('_bssm_sde_pm_mcmc', <8>, theta, nsim, L, seed, iter, burnin, thin, gamma, target_acceptance, S, end_ram, type, verbose)
    (src/R_sde.cpp:134-178), '_bssm_sde_da_mcmc' with (L_c, L_f) (:180-219) and '_bssm_R_milstein' (src/R_milstein.cpp:4-22)"""
    from bssm_amd import mcmc
    from oracle import oracle as O
    rng = np.random.default_rng(8)
    n = 20
    y = np.exp(np.cumsum(rng.normal(0.01, 0.05, n)))
    sde = M.ssm_sde(y, "sde_gbm", theta=np.array([0.02, 0.2, 0.6]), x0=1.0, positive=True)
    sprior = M.example_log_prior("sde_gbm")
    drift = R.extptr(bssm_amd_model_id=np.asarray([sde["model_id"]], dtype=np.int32))
    lp = R.extptr(bssm_amd_prior_fn=lambda th: float(sprior(th)))
    sargs = [y, 1.0, True, drift, R.extptr(), R.extptr(), lp, R.extptr()]
    th0 = np.asarray(sde["theta"], dtype=float)
    S0 = np.diag(0.1 * np.maximum(0.1, np.abs(th0)))
    it, burn, nsim, seed = 80, 40, 32, 5
    ref = mcmc.run_mcmc(sde, iter=it, particles=nsim, output_type="full", mcmc_type="pm", burnin=burn, seed=seed, L_f=3)
    out = R.call("_bssm_sde_pm_mcmc", *sargs, th0, nsim, 3, seed, it, burn, 1, 2 / 3, 0.234, S0, False, 1, False)
    _after(R)
    _chain_equal(out, ref, 1)
    ref = mcmc.run_mcmc(sde, iter=it, particles=nsim, output_type="summary", mcmc_type="da", burnin=burn, seed=seed, L_c=2, L_f=4)
    out = R.call("_bssm_sde_da_mcmc", *sargs, th0, nsim, 2, 4, seed, it, burn, 1, 2 / 3, 0.234, S0, False, 2, False)
    _after(R)
    _chain_equal(out, ref, 2)
    # R_milstein(x0, L, t, theta, drift, diffusion, ddiffusion, positive, seed): the device path against the oracle's restatement of
    # milstein() (src/milstein.cpp:18-48) on the same stream, for the filters' unit interval and for another terminal time
    for model_name, th, pos in (("sde_gbm", np.array([0.05, 0.3, 0.1]), True), ("sde_poisson_ou", np.array([-0.5, 1.0, -1.0]), False)):
        mdl = M.ssm_sde(y, model_name, theta=th, x0=1.3, positive=pos)
        dr = R.extptr(bssm_amd_model_id=np.asarray([mdl["model_id"]], dtype=np.int32))
        for L_, t_end, sd in ((3, 1.0, 7), (5, 2.5, 8), (0, 0.25, 9)):
            x = R.call("_bssm_R_milstein", 1.3, L_, t_end, th, dr, R.extptr(), R.extptr(), pos, sd)
            _after(R)
            assert x[0] == F.milstein(mdl, L_, t=t_end, seed=sd)
            want = O.milstein({**mdl, "L": L_}, t_end, seed=sd)
            assert abs(x[0] - want) <= 1e-12 * max(1.0, abs(want)), (model_name, L_, t_end, x[0], want)
    # t = 1 is the prior draw of particle 0 of the bootstrap filter with that seed (src/model_ssm_sde.cpp:25-28)
    bf = F.bootstrap_filter(sde, 64, seed=7, L=3)
    assert F.milstein(sde, 3, t=1.0, seed=7) == bf["alpha"][0, 0, 0]
