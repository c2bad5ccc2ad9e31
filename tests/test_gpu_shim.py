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
