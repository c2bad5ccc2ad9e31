"""GPU parity of the ssm_nlg / ssm_sde bootstrap filters (built-in model functors) against the oracle
(src/model_ssm_nlg.cpp:1066-1146, src/model_ssm_sde.cpp:21-82, src/milstein.cpp:18-48)."""
import numpy as np
import pytest

from bssm_amd import filters as F
from bssm_amd import models as M

pytestmark = pytest.mark.gpu


def _models():
    rng = np.random.default_rng(12)
    out = {}
    # growth model of vignettes/growth_model.Rmd:51-75: K=500, p1=50, H=1, R1=.05, R2=1, dT=.1
    n, K, dT = 60, 500.0, 0.1
    r = np.zeros(n); p = np.zeros(n); r[0] = 0.0; p[0] = 50.0
    for t in range(1, n):
        r[t] = r[t - 1] + rng.normal(0, 0.05)
        rr = np.exp(r[t - 1]) / (1 + np.exp(r[t - 1]))
        p[t] = K * p[t - 1] * np.exp(rr * dT) / (K + p[t - 1] * (np.exp(rr * dT) - 1)) + rng.normal(0, 1)
    y = p + rng.normal(0, 1, n); y[[7, 8]] = np.nan
    out["nlg_growth"] = (M.ssm_nlg(y, "nlg_growth", theta=[0.0, np.log(0.05), 0.0], known_params=[dT, K, 0.0, 50.0, 1.0, 100.0]), None)
    a = np.zeros(40); a[0] = 0.3
    for t in range(1, 40):
        a[t] = 0.2 * (1 - 0.9) + 0.9 * a[t - 1] + rng.normal(0, 0.3)
    out["nlg_ar_exp"] = (M.ssm_nlg(np.exp(a) + rng.normal(0, 0.4, 40), "nlg_ar_exp", theta=[0.2, 0.9, np.log(0.3), np.log(0.4)]), None)
    out["nlg_sin_exp"] = (M.ssm_nlg(rng.normal(1.0, 0.5, 30), "nlg_sin_exp", theta=[np.log(0.6), np.log(0.4)]), None)
    out["nlg_linear_gaussian"] = (M.ssm_nlg(np.cumsum(rng.normal(0, 1, 30)) + rng.normal(0, 0.5, 30), "nlg_linear_gaussian",
                                            theta=[np.log(0.5)]), None)
    x = np.exp(np.cumsum(rng.normal(0.01, 0.1, 25)))
    out["sde_gbm"] = (M.ssm_sde(np.log(x) + rng.normal(0, 0.5, 25), "sde_gbm", theta=[0.05, 0.3, 1.0], x0=1.0, positive=True), 3)
    yp = rng.poisson(np.exp(0.7), 25).astype(float); yp[5] = np.nan
    out["sde_poisson_ou"] = (M.ssm_sde(yp, "sde_poisson_ou", theta=[np.log(0.5), 0.7, np.log(0.4)], x0=0.5, positive=False), 2)
    return out


MODELS = _models()


def _rel(a, b):
    return np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))


@pytest.mark.parametrize("path", ["wg", "grid"])
@pytest.mark.parametrize("name", sorted(MODELS))
@pytest.mark.parametrize("N", [30, 600])
def test_functor_bootstrap_filter(oracle, monkeypatch, name, N, path):
    monkeypatch.setenv("BSSM_AMD_FORCE_PATH", path)
    model, L = MODELS[name]
    out = F.bootstrap_filter(model, N, seed=8, return_indices=True, L=L)
    ref = oracle.fn_bsf_filter({**model, "L": L or 0}, N, seed=8)
    assert np.array_equal(out["indices"], ref["indices"])
    assert abs(out["logLik"] - ref["logLik"]) <= 1e-9 * model.n
    assert _rel(np.transpose(out["alpha"], (1, 0, 2)), ref["alpha"]) <= 1e-9
    assert _rel(out["weights"], ref["weights"]) <= 1e-9


@pytest.mark.parametrize("name", ["nlg_growth", "sde_poisson_ou"])
def test_functor_bsf_smoother(oracle, name):
    model, L = MODELS[name]
    n = model.n
    out = F.particle_smoother(model, 200, method="bsf", seed=4, L=L)
    ref = oracle.fn_bsf_filter({**model, "L": L or 0}, 200, seed=4)
    mean, cov = oracle.summary(oracle.filter_smoother(ref["alpha"], ref["indices"]))
    assert abs(out["logLik"] - ref["logLik"]) <= 1e-9 * n
    assert _rel(out["alphahat"], mean.T[:n]) <= 1e-8 and _rel(out["Vt"], cov[:, :, :n]) <= 1e-7


def test_nlg_linear_gaussian_equals_kalman(oracle):
    """Analytic pin: the nlg_linear_gaussian example is an exact LGSSM (a1=0, P1=1, T=Z=1, R=1, H=e^theta0)."""
    model, _ = MODELS["nlg_linear_gaussian"]
    lg = M.ssm_ulg(model["y"], Z=1.0, H=0.5, T=1.0, R=1.0, a1=0.0, P1=1.0)
    kal = oracle.ulg_loglik(M.model_arrays(lg))
    lls = [F.bootstrap_filter(model, 20000, seed=s)["logLik"] for s in range(8)]
    assert abs(np.mean(lls) - kal) < 4 * np.std(lls) / np.sqrt(8) + 0.02


def test_unknown_example_is_rejected():
    with pytest.raises(ValueError, match="built in"):
        M.ssm_nlg([1.0, 2.0], "my_snippet", theta=[0.0])


# ---------------------------------------------------------------- psi-APF of ssm_nlg
@pytest.mark.parametrize("name", ["nlg_growth", "nlg_ar_exp", "nlg_sin_exp", "nlg_linear_gaussian"])
def test_nlg_psi_setup_matches_oracle(oracle, name):
    model, _ = MODELS[name]
    so, sp = oracle.fn_psi_setup(model), F.psi_setup(model)
    assert so["ok"] == 1
    for k in ("mode", "scales", "alphahat", "Vt", "Ct", "Z", "T", "D", "C"):
        assert np.allclose(so[k], sp[k], rtol=1e-9, atol=1e-10), k
    assert np.isclose(so["approx_loglik"], sp["approx_loglik"], rtol=1e-11)
    so2, sp2 = oracle.fn_psi_setup(model, is_mode=so["mode"]), F.psi_setup(model, is_mode=so["mode"])
    assert sp2["approx_loglik"] == 0.0 and np.all(sp2["scales"] == 0.0)
    assert np.allclose(so2["Vt"], sp2["Vt"], rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("path", ["wg", "grid"])
@pytest.mark.parametrize("name", ["nlg_growth", "nlg_ar_exp", "nlg_sin_exp", "nlg_linear_gaussian"])
def test_nlg_psi_smoother(oracle, monkeypatch, name, path):
    monkeypatch.setenv("BSSM_AMD_FORCE_PATH", path)
    model, _ = MODELS[name]
    n = model.n
    out = F.particle_smoother(model, 80, method="psi", seed=3, return_indices=True)
    ref = oracle.fn_psi_filter(model, 80, seed=3)
    assert np.array_equal(out["indices"], ref["indices"])
    assert abs(out["logLik"] - ref["logLik"]) <= 1e-9 * n
    assert _rel(out["weights"], ref["weights"]) <= 1e-7
    mean, cov = oracle.summary(oracle.filter_smoother(ref["alpha"], ref["indices"]))
    assert _rel(out["alphahat"], mean.T[:n]) <= 1e-8 and _rel(out["Vt"], cov[:, :, :n]) <= 1e-7
    assert abs(F.logLik(model, 80, method="psi", seed=3) - ref["logLik"]) <= 1e-9 * n


def test_nlg_post_correct(oracle):
    """postcorrection_nonlinear (src/R_postcorrection.cpp:252-317): is_correction_psi over theta-draws of the growth model"""
    from bssm_amd import post_correct as PC
    model, _ = MODELS["nlg_growth"]
    rng = np.random.default_rng(2)
    ns = 6
    theta = np.array([0.0, np.log(0.05), 0.0]) + 0.05 * rng.normal(size=(ns, 3))
    counts = rng.integers(1, 4, ns).astype(np.uint32)
    modes = np.stack([oracle.fn_psi_setup({**model, "theta": th})["mode"].reshape(-1, order="F") for th in theta], axis=1)
    mo = dict(theta=theta, counts=counts, modes=modes, posterior=np.zeros(ns), output_type=2)
    out = PC.post_correct(model, mo, 32, threads=2, is_type="is2", seed=11)
    m, n = 2, model.n
    ah = np.zeros((m, n + 1, ns)); vt = np.zeros((m, m, n + 1, ns)); w = np.zeros(ns)
    for i, th in enumerate(theta):
        mi = {**model, "theta": th}
        ref = oracle.fn_psi_filter(mi, 32, seed=11, stream=i, is_mode=modes[:, i].reshape((m, n), order="F"))
        assert abs(out["loglik"][i] - ref["logLik"]) <= 1e-9 * n
        w[i] = np.exp(ref["logLik"])
        ah[:, :, i], vt[:, :, :, i] = oracle.summary(oracle.filter_smoother(ref["alpha"], ref["indices"]))
    assert np.allclose(out["weights"], w, rtol=1e-8)
    wc = w * counts
    assert np.allclose(out["alphahat"], (np.tensordot(ah, wc, axes=([2], [0])) / wc.sum()).T, rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("path", ["wg", "grid"])
@pytest.mark.parametrize("name", ["nlg_growth", "nlg_ar_exp", "nlg_sin_exp", "nlg_linear_gaussian"])
@pytest.mark.parametrize("N", [20, 700])
def test_ekpf_filter_matches_oracle(oracle, monkeypatch, name, N, path):
    """EKF-proposal particle filter (src/model_ssm_nlg.cpp:1148-1293) against its oracle restatement, same stream"""
    monkeypatch.setenv("BSSM_AMD_FORCE_PATH", path)
    model, _ = MODELS[name]
    out = F.ekpf_filter(model, N, seed=6, return_indices=True)
    ref = oracle.fn_ekpf_filter(model, N, seed=6)
    assert np.array_equal(out["indices"], ref["indices"])
    assert abs(out["logLik"] - ref["logLik"]) <= 1e-9 * model.n
    assert _rel(np.transpose(out["alpha"], (1, 0, 2)), ref["alpha"]) <= 1e-9
    assert _rel(out["weights"], ref["weights"]) <= 1e-8
    sm = F.particle_smoother(model, N, method="ekf", seed=6)
    assert abs(sm["logLik"] - ref["logLik"]) <= 1e-9 * model.n
    ll = F.logLik(model, N, method="ekf", seed=6)
    assert abs(ll - ref["logLik"]) <= 1e-9 * model.n


# ---------------------------------------------------------------------------------------------------------------
# User-defined models: the reference compiles the user's C++ at run time and calls it through function pointers; here
# the user's __device__ functions are compiled by hipcc into the filter kernels (bssm_user_model_build).  Writing the
# shipped example models as USER snippets must reproduce the built-in versions exactly.
USER_SIN_EXP = r"""
#define BSSM_USER_NLG 1
#define BSSM_USER_JACOBIANS 1
// nlg_sin_exp of R/cpp_example_models.R:635-760 written as a user model: T(a) = sin(a), Z(a) = exp(a)
BSSM_FN void bssm_user_T(const double* alpha, const double* theta, const double* known, double* out) { out[0] = sin(alpha[0]); }
BSSM_FN double bssm_user_Z(const double* alpha, const double* theta, const double* known) { return exp(alpha[0]); }
BSSM_FN void bssm_user_Tg(const double* alpha, const double* theta, const double* known, double* out) { out[0] = cos(alpha[0]); }
BSSM_FN void bssm_user_Zg(const double* alpha, const double* theta, const double* known, double* out) { out[0] = exp(alpha[0]); }
"""
USER_GROWTH = r"""
#define BSSM_USER_NLG 1
#define BSSM_USER_JACOBIANS 1
// logistic growth: R/cpp_example_models.R:393-450 (known = dT, K, ...)
BSSM_FN void bssm_user_T(const double* a, const double* theta, const double* known, double* out) {
  const double dT = known[0], K = known[1];
  const double r = exp(a[0]) / (1.0 + exp(a[0]));
  const double e = exp(r * dT);
  out[0] = a[0];
  out[1] = K * a[1] * e / (K + a[1] * (e - 1.0));
}
BSSM_FN double bssm_user_Z(const double* a, const double* theta, const double* known) { return a[1]; }
BSSM_FN void bssm_user_Tg(const double* a, const double* theta, const double* known, double* Tg) {
  const double dT = known[0], K = known[1];
  const double r = exp(a[0]) / (1.0 + exp(a[0]));
  const double tmp = exp(r * dT) / pow(K + a[1] * (exp(r * dT) - 1.0), 2);
  Tg[0] = 1.0; Tg[2] = 0.0;
  Tg[1] = dT * K * a[1] * (K - a[1]) * tmp * r / (1.0 + exp(a[0]));
  Tg[3] = K * K * tmp;
}
BSSM_FN void bssm_user_Zg(const double* a, const double* theta, const double* known, double* Zg) { Zg[0] = 0.0; Zg[1] = 1.0; }
"""
USER_GBM = r"""
#define BSSM_USER_SDE 1
// geometric Brownian motion with Gaussian observations of log(x): R/cpp_example_models.R:109-193
BSSM_FN double bssm_user_drift(double x, const double* theta) { return theta[0] * x; }
BSSM_FN double bssm_user_diffusion(double x, const double* theta) { return fmax(0.0, theta[1] * x); }
BSSM_FN double bssm_user_ddiffusion(double x, const double* theta) { return x > 0.0 ? theta[1] : 0.0; }
BSSM_FN double bssm_user_log_obs(double y, double x, const double* theta) {
  const double q = (y - log(x)) / theta[2];
  return -0.5 * (1.8378770664093453 + 2.0 * log(theta[2])) - 0.5 * q * q;
}
"""


@pytest.mark.parametrize("path", ["wg", "grid"])
def test_user_defined_nlg_models_equal_builtin(oracle, monkeypatch, path):
    monkeypatch.setenv("BSSM_AMD_FORCE_PATH", path)
    # one state
    b, _ = MODELS["nlg_sin_exp"]
    th = np.asarray(b["theta"])
    u = M.ssm_nlg_user(b["y"], USER_SIN_EXP, theta=th, a1=[0.0], P1=[[1.0]], H=np.exp(th[0]), R_diag=[np.exp(th[1])])
    for N in (50, 900):
        ob, ou = F.bootstrap_filter(b, N, seed=4, return_indices=True), F.bootstrap_filter(u, N, seed=4, return_indices=True)
        ref = oracle.fn_bsf_filter({**b, "L": 0}, N, seed=4)
        assert np.array_equal(ou["indices"], ref["indices"]) and np.array_equal(ou["indices"], ob["indices"])
        assert abs(ou["logLik"] - ref["logLik"]) <= 1e-9 * b.n
        assert _rel(ou["alpha"], ob["alpha"]) <= 1e-9
    # two states
    b, _ = MODELS["nlg_growth"]
    th, kp = np.asarray(b["theta"]), np.asarray(b["known_params"])
    u = M.ssm_nlg_user(b["y"], USER_GROWTH, theta=th, known_params=kp, a1=kp[2:4], P1=np.diag(kp[4:6]), H=np.exp(th[0]),
                       R_diag=np.exp(th[1:3]))
    ob, ou = F.particle_smoother(b, 600, method="bsf", seed=9), F.particle_smoother(u, 600, method="bsf", seed=9)
    assert abs(ou["logLik"] - ob["logLik"]) <= 1e-9 * b.n
    assert _rel(ou["alphahat"], ob["alphahat"]) <= 1e-9 and _rel(ou["Vt"], ob["Vt"]) <= 1e-8
    assert abs(F.logLik(u, 600, method="bsf", seed=9) - ob["logLik"]) <= 1e-9 * b.n


@pytest.mark.parametrize("name,snippet", [("nlg_sin_exp", "sin"), ("nlg_growth", "growth")])
def test_user_defined_nlg_psi_and_ekf_equal_builtin(name, snippet):
    """psi-APF (host mode search through the user's host entry points + device weights) and the EKF particle filter"""
    b, _ = MODELS[name]
    th = np.asarray(b["theta"])
    if snippet == "sin":
        u = M.ssm_nlg_user(b["y"], USER_SIN_EXP, theta=th, a1=[0.0], P1=[[1.0]], H=np.exp(th[0]), R_diag=[np.exp(th[1])])
    else:
        kp = np.asarray(b["known_params"])
        u = M.ssm_nlg_user(b["y"], USER_GROWTH, theta=th, known_params=kp, a1=kp[2:4], P1=np.diag(kp[4:6]), H=np.exp(th[0]),
                           R_diag=np.exp(th[1:3]))
    for method in ("psi", "ekf"):
        ob, ou = F.particle_smoother(b, 300, method=method, seed=5), F.particle_smoother(u, 300, method=method, seed=5)
        assert abs(ou["logLik"] - ob["logLik"]) <= 1e-8 * b.n, method
        assert _rel(ou["alphahat"], ob["alphahat"]) <= 1e-8 and _rel(ou["Vt"], ob["Vt"]) <= 1e-7
    sb, su = F.psi_setup(b), F.psi_setup(u)
    assert _rel(su["mode"], sb["mode"]) <= 1e-9 and abs(su["approx_loglik"] - sb["approx_loglik"]) <= 1e-8 * b.n


def test_user_defined_sde_model_equals_builtin(oracle):
    b, L = MODELS["sde_gbm"]
    u = M.ssm_sde_user(b["y"], USER_GBM, theta=b["theta"], x0=b["x0"], positive=True)
    ob, ou = F.bootstrap_filter(b, 400, seed=3, return_indices=True, L=L), F.bootstrap_filter(u, 400, seed=3, return_indices=True, L=L)
    ref = oracle.fn_bsf_filter({**b, "L": L}, 400, seed=3)
    assert np.array_equal(ou["indices"], ref["indices"])
    assert abs(ou["logLik"] - ref["logLik"]) <= 1e-9 * b.n and abs(ou["logLik"] - ob["logLik"]) <= 1e-9 * b.n


def test_user_model_compile_error_is_reported():
    with pytest.raises(Exception, match="compiling the user model failed"):
        M_bad = M.ssm_nlg_user(np.zeros(5), "#define BSSM_USER_NLG 1\nthis is not C++", theta=[0.0], a1=[0.0], P1=[[1.0]], H=1.0, R_diag=[1.0])
        F.bootstrap_filter(M_bad, 10, seed=1)


def test_user_model_with_theta_dependent_numbers_in_post_correct_and_run_mcmc(oracle):
    """User-defined ssm_nlg whose H, diag(R) are functions of theta (as the reference's H_fn / R_fn are): the IS
    post-correction and run_mcmc evaluate them for every theta and agree with the built-in model of the same equations."""
    from bssm_amd import mcmc, post_correct as PC
    b, _ = MODELS["nlg_growth"]
    th, kp = np.asarray(b["theta"], dtype=float), np.asarray(b["known_params"])
    u = M.ssm_nlg_user(b["y"], USER_GROWTH, theta=th, known_params=kp, a1=kp[2:4], P1=np.diag(kp[4:6]),
                       H=lambda t: np.exp(t[0]), R_diag=lambda t: np.exp(t[1:3]))
    u = M.BssmModel(u.classes, **{**u, "prior_fn": M.example_log_prior("nlg_growth")})
    # the approximate chain (host) of the two models is the same chain
    cb = mcmc.run_mcmc(b, 120, burnin=40, mcmc_type="approx", seed=2)
    cu = mcmc.run_mcmc(u, 120, burnin=40, mcmc_type="approx", seed=2)
    assert np.array_equal(cb["counts"], cu["counts"]) and np.allclose(cb["theta"], cu["theta"], rtol=1e-10, atol=1e-12)
    assert np.allclose(cb["approx_loglik"], cu["approx_loglik"], rtol=0, atol=1e-8 * b.n)
    assert len(np.unique(cb["theta"][:, 0])) > 3  # H really varies along the chain
    # post-correction of the stored values
    pb = PC.post_correct(b, dict(cb, output_type=3), 64, threads=2, is_type="is2", seed=5)
    pu = PC.post_correct(u, dict(cu, output_type=3), 64, threads=2, is_type="is2", seed=5)
    assert np.allclose(pb["weights"], pu["weights"], rtol=1e-7)
    # and a pseudo-marginal chain through the device filter
    mb = mcmc.run_mcmc(b, 40, particles=32, burnin=10, mcmc_type="pm", sampling_method="bsf", output_type="theta", seed=3)
    mu = mcmc.run_mcmc(u, 40, particles=32, burnin=10, mcmc_type="pm", sampling_method="bsf", output_type="theta", seed=3)
    assert np.array_equal(mb["counts"], mu["counts"]) and np.allclose(mb["theta"], mu["theta"], rtol=1e-9, atol=1e-11)


# A three-state, two-noise model with state-dependent H_fn and R_fn (src/model_ssm_nlg.h:10-21 allows any m, k and
# state-dependent H / R): stochastic-volatility-like level + log-scale + damped slope
#   T(a) = (a0 + a2, th0 + th1 (a1 - th0), th2 a2),   R(a) = [[exp(a1 / 2) th3, 0], [0, th4], [0.3 exp(a1 / 2) th3, th5]],
#   Z(a) = a0 + 0.1 a0^2 / (1 + a0^2),   H(a) = kn0 (1 + 0.5 / (1 + a0^2))
USER_WIDE3 = r"""
#define BSSM_USER_NLG 1
#define BSSM_USER_HFN 1
#define BSSM_USER_RFN 1
BSSM_FN void bssm_user_T(const double* a, const double* th, const double* kn, double* out) {
  out[0] = a[0] + a[2]; out[1] = th[0] + th[1] * (a[1] - th[0]); out[2] = th[2] * a[2];
}
BSSM_FN double bssm_user_Z(const double* a, const double* th, const double* kn) { return a[0] + 0.1 * a[0] * a[0] / (1.0 + a[0] * a[0]); }
BSSM_FN double bssm_user_H(const double* a, const double* th, const double* kn) { return kn[0] * (1.0 + 0.5 / (1.0 + a[0] * a[0])); }
BSSM_FN void bssm_user_R(const double* a, const double* th, const double* kn, double* out) {
  const double s = exp(0.5 * a[1]) * th[3];
  out[0] = s; out[1] = 0.0; out[2] = 0.3 * s;        /* first column */
  out[3] = 0.0; out[4] = th[4]; out[5] = th[5];      /* second column */
}
"""


def _wide3_python():
    Z = lambda t, a, th, kn: a[0] + 0.1 * a[0] ** 2 / (1.0 + a[0] ** 2)
    H = lambda t, a, th, kn: kn[0] * (1.0 + 0.5 / (1.0 + a[0] ** 2))
    T = lambda t, a, th, kn: np.array([a[0] + a[2], th[0] + th[1] * (a[1] - th[0]), th[2] * a[2]])
    def R(t, a, th, kn):
        s = np.exp(0.5 * a[1]) * th[3]
        return np.array([[s, 0.0], [0.0, th[4]], [0.3 * s, th[5]]])
    return Z, H, T, R


@pytest.mark.parametrize("path,N", [("wg", 300), ("wg", 1500), ("grid", 5000)])
def test_user_model_three_states_state_dependent_H_and_R(oracle, monkeypatch, path, N):
    """bootstrap filter of a user model with 3 states, a 3 x 2 matrix R_fn and state-dependent H_fn / R_fn against the
    oracle's restatement of ssm_nlg::bsf_filter for function-pointer models (src/model_ssm_nlg.cpp:1066-1146)"""
    monkeypatch.setenv("BSSM_AMD_FORCE_PATH", path)
    rng = np.random.default_rng(8)
    n = 25
    th = np.array([-1.0, 0.9, 0.8, 0.5, 0.2, 0.05]); kn = np.array([0.7])
    Z, H, T, R = _wide3_python()
    a = np.array([0.5, -1.0, 0.1]); y = np.zeros(n)
    for t in range(n):
        y[t] = Z(t, a, th, kn) + H(t, a, th, kn) * rng.normal()
        a = T(t, a, th, kn) + R(t, a, th, kn) @ rng.normal(size=2)
    y[6] = np.nan
    a1 = np.array([0.5, -1.0, 0.0]); P1 = np.array([[1.0, 0.2, 0.0], [0.2, 0.5, 0.0], [0.0, 0.0, 0.0]])  # singular: psd_chol drops the third state
    u = M.ssm_nlg_user(y, USER_WIDE3, theta=th, known_params=kn, a1=a1, P1=P1, n_states=3,
                       R=np.zeros((3, 2)))  # (R only sets n_etas here: the kernels call bssm_user_R)
    out = F.bootstrap_filter(u, N, seed=5, return_indices=True)
    ref = oracle.nlg_general_bsf_filter(y, 3, 2, th, kn, a1, P1, Z, H, T, R, N, seed=5)
    assert np.array_equal(out["indices"], ref["indices"])
    assert abs(out["logLik"] - ref["logLik"]) <= 1e-9 * n
    assert _rel(np.transpose(out["alpha"], (1, 0, 2)), ref["alpha"]) <= 1e-9
    assert _rel(out["weights"], ref["weights"]) <= 1e-8
    sm = F.particle_smoother(u, N, method="bsf", seed=5)
    assert abs(sm["logLik"] - ref["logLik"]) <= 1e-9 * n
    ah, Vt = oracle.summary(oracle.filter_smoother(ref["alpha"], ref["indices"]))
    assert _rel(sm["alphahat"], ah.T[:n]) <= 1e-8 and _rel(sm["Vt"], Vt[:, :, :n]) <= 1e-7
    # the psi-APF / EKF particle filter refuse what they are not built for (this snippet has no Jacobians; R is a matrix)
    with pytest.raises(Exception, match="JACOBIANS|diagonal R"):
        F.particle_smoother(u, 50, method="psi", seed=1)


USER_CONST_R4 = r"""
#define BSSM_USER_NLG 1
BSSM_FN void bssm_user_T(const double* a, const double* th, const double* kn, double* out) {
  out[0] = a[0] + a[1]; out[1] = th[0] * a[1] + sin(a[2]); out[2] = 0.9 * a[2] + a[3]; out[3] = 0.5 * a[3];
}
BSSM_FN double bssm_user_Z(const double* a, const double* th, const double* kn) { return a[0] + a[2]; }
"""


def test_user_model_constant_matrix_R_four_states(oracle):
    """4 states, constant 4 x 3 matrix R and constant H passed as numbers (no snippet R / H functions)"""
    code = USER_CONST_R4
    rng = np.random.default_rng(2)
    n, N = 20, 700
    th = np.array([0.7])
    Rm = np.array([[0.3, 0.0, 0.0], [0.1, 0.2, 0.0], [0.0, 0.0, 0.4], [0.0, 0.1, 0.1]])
    y = np.cumsum(rng.normal(0, 0.5, n))
    a1 = np.array([0.0, 0.1, 0.0, 0.0]); P1 = np.diag([1.0, 0.5, 0.3, 0.2]); P1[0, 1] = P1[1, 0] = 0.3
    u = M.ssm_nlg_user(y, code, theta=th, a1=a1, P1=P1, H=0.6, R=Rm, n_states=4)
    out = F.bootstrap_filter(u, N, seed=3, return_indices=True)
    Z = lambda t, a, th, kn: a[0] + a[2]
    T = lambda t, a, th, kn: np.array([a[0] + a[1], th[0] * a[1] + np.sin(a[2]), 0.9 * a[2] + a[3], 0.5 * a[3]])
    ref = oracle.nlg_general_bsf_filter(y, 4, 3, th, [], a1, P1, Z, lambda *q: 0.6, T, lambda *q: Rm, N, seed=3)
    assert np.array_equal(out["indices"], ref["indices"])
    assert abs(out["logLik"] - ref["logLik"]) <= 1e-9 * n
    assert _rel(np.transpose(out["alpha"], (1, 0, 2)), ref["alpha"]) <= 1e-9


USER_LINEAR3 = r"""
#define BSSM_USER_NLG 1
#define BSSM_USER_JACOBIANS 1
// linear-Gaussian model with 3 states written as a user model: T(a) = A a with A column-major in theta[0..7], known[3]
// (at most 8 theta); Z(a) = z'a with z = known[0..2]
BSSM_FN void bssm_user_T(const double* a, const double* th, const double* kn, double* out) {
  out[0] = th[0] * a[0] + th[3] * a[1] + th[6] * a[2];
  out[1] = th[1] * a[0] + th[4] * a[1] + th[7] * a[2];
  out[2] = th[2] * a[0] + th[5] * a[1] + kn[3] * a[2];
}
BSSM_FN double bssm_user_Z(const double* a, const double* th, const double* kn) { return kn[0] * a[0] + kn[1] * a[1] + kn[2] * a[2]; }
BSSM_FN void bssm_user_Tg(const double* a, const double* th, const double* kn, double* Tg) { for (int q = 0; q < 8; q++) Tg[q] = th[q]; Tg[8] = kn[3]; }
BSSM_FN void bssm_user_Zg(const double* a, const double* th, const double* kn, double* Zg) { for (int c = 0; c < 3; c++) Zg[c] = kn[c]; }
"""
USER_NONLIN4 = r"""
#define BSSM_USER_NLG 1
#define BSSM_USER_JACOBIANS 1
// 4 states, mildly non-linear transition and observation
BSSM_FN void bssm_user_T(const double* a, const double* th, const double* kn, double* out) {
  out[0] = a[0] + a[1]; out[1] = th[0] * a[1] + 0.3 * sin(a[2]); out[2] = 0.9 * a[2] + 0.2 * a[3]; out[3] = 0.5 * a[3];
}
BSSM_FN double bssm_user_Z(const double* a, const double* th, const double* kn) { return a[0] + 0.5 * a[2] + 0.1 * a[0] * a[0] / (1.0 + a[0] * a[0]); }
BSSM_FN void bssm_user_Tg(const double* a, const double* th, const double* kn, double* Tg) {
  for (int q = 0; q < 16; q++) Tg[q] = 0.0;
  Tg[0 + 4 * 0] = 1.0; Tg[0 + 4 * 1] = 1.0; Tg[1 + 4 * 1] = th[0]; Tg[1 + 4 * 2] = 0.3 * cos(a[2]);
  Tg[2 + 4 * 2] = 0.9; Tg[2 + 4 * 3] = 0.2; Tg[3 + 4 * 3] = 0.5;
}
BSSM_FN void bssm_user_Zg(const double* a, const double* th, const double* kn, double* Zg) {
  const double q = 1.0 + a[0] * a[0];
  Zg[0] = 1.0 + 0.2 * a[0] / (q * q); Zg[1] = 0.0; Zg[2] = 0.5; Zg[3] = 0.0;
}
"""


@pytest.mark.parametrize("path,N", [("wg", 64), ("wg", 1500), ("grid", 5000)])
def test_user_model_three_states_psi_apf_is_exact_for_a_linear_gaussian_model(oracle, monkeypatch, path, N):
    """psi-APF of a user model with 3 states (constant diagonal R, diagonal P1; runs in the 4-state kernels with one padding
    component): for a linear-Gaussian model the approximating model IS the model, every psi weight is one and the
    estimate equals the Kalman-filter log-likelihood for any N (src/model_ssm_nlg.cpp:993-1064; the oracle's ulg_loglik)"""
    monkeypatch.setenv("BSSM_AMD_FORCE_PATH", path)
    rng = np.random.default_rng(21)
    n = 30
    A = np.array([[0.8, 0.1, 0.0], [-0.2, 0.7, 0.1], [0.05, 0.0, 0.9]])
    z = np.array([1.0, 0.5, -0.3]); rd = np.array([0.3, 0.2, 0.1]); Hs = 0.6
    a1 = np.array([0.2, -0.1, 0.0]); P1 = np.diag([0.5, 0.3, 0.2])
    a = a1 + rng.normal(size=3) * np.sqrt(np.diag(P1)); y = np.zeros(n)
    for t in range(n):
        y[t] = z @ a + Hs * rng.normal(); a = A @ a + rd * rng.normal(size=3)
    y[7] = np.nan
    u = M.ssm_nlg_user(y, USER_LINEAR3, theta=A.reshape(-1, order="F")[:8], known_params=np.append(z, A[2, 2]), a1=a1, P1=P1, H=Hs, R=np.diag(rd), n_states=3)
    lg = M.ssm_ulg(y, Z=z, H=Hs, T=A, R=np.diag(rd), a1=a1, P1=P1)
    kal = oracle.ulg_loglik(M.model_arrays(lg))
    out = F.particle_smoother(u, N, method="psi", seed=3)
    assert abs(out["logLik"] - kal) <= 1e-8 * n
    ah = oracle.ulg_smoother(M.model_arrays(lg))[0]  # m x (n + 1)
    assert abs(F.logLik(u, N, method="psi", seed=11) - kal) <= 1e-8 * n
    su = F.psi_setup(u)
    assert abs(su["approx_loglik"] - kal) <= 1e-8 * n
    # the smoothed means of the particle system scatter around the Kalman smoother's (exact draws, equal weights)
    if N >= 1500:
        sm = np.asarray(out["alphahat"])  # (n + 1) x m, or n x m
        assert np.max(np.abs(sm - ah.T[:sm.shape[0]])) < 6.0 * np.sqrt(np.max(P1) / N) + 0.02


def test_user_model_four_states_psi_apf_agrees_with_the_bootstrap_filter(monkeypatch):
    """non-linear user model with 4 states: the psi-APF and the bootstrap filter estimate the same likelihood (both
    unbiased; log-mean-exp over seeds), the psi-APF with the smaller spread; IS post-correction through the psi-APF
    reproduces the single calls"""
    from bssm_amd import post_correct as PC
    rng = np.random.default_rng(4)
    n = 40
    th = np.array([0.7]); rd = np.array([0.2, 0.15, 0.3, 0.1]); Hs = 0.5
    a1 = np.zeros(4); P1 = np.diag([0.5, 0.2, 0.3, 0.1])
    a = rng.normal(size=4) * np.sqrt(np.diag(P1)); y = np.zeros(n)
    for t in range(n):
        y[t] = a[0] + 0.5 * a[2] + 0.1 * a[0] ** 2 / (1 + a[0] ** 2) + Hs * rng.normal()
        a = np.array([a[0] + a[1], th[0] * a[1] + 0.3 * np.sin(a[2]), 0.9 * a[2] + 0.2 * a[3], 0.5 * a[3]]) + rd * rng.normal(size=4)
    u = M.ssm_nlg_user(y, USER_NONLIN4, theta=th, a1=a1, P1=P1, H=Hs, R=np.diag(rd), n_states=4)
    lme = lambda v: np.max(v) + np.log(np.mean(np.exp(v - np.max(v))))
    psi = np.array([F.logLik(u, 2000, method="psi", seed=s) for s in range(24)])
    bsf = np.array([F.logLik(u, 2000, method="bsf", seed=s) for s in range(24)])
    assert np.all(np.isfinite(psi)) and np.all(np.isfinite(bsf))
    assert abs(lme(psi) - lme(bsf)) < 4.0 * (np.std(psi) + np.std(bsf)) / np.sqrt(24) + 0.05
    assert np.std(psi) < np.std(bsf)
    for path in ("wg", "grid"):  # both kernel shapes give the same particle system
        monkeypatch.setenv("BSSM_AMD_FORCE_PATH", path)
        o = F.particle_smoother(u, 700, method="psi", seed=2)
        if path == "wg": ref = o
    assert abs(o["logLik"] - ref["logLik"]) <= 1e-9 * n and _rel(o["alphahat"], ref["alphahat"]) <= 1e-8


@pytest.mark.parametrize("dev", ["0", "1"])
def test_user_model_three_states_post_correct_psi(monkeypatch, dev):
    """IS post-correction (is2, approximate_for_is at stored modes, src/model_ssm_nlg.cpp:146-176) of the 3-state
    linear-Gaussian user model: the linearisation of a linear model is the model whatever the stored mode, so every
    psi weight and every IS weight is exactly one -- through the host set-up (0) and the device smoother (1), with the
    approximating (T_t, C_t) rows padded to the 4-state kernels"""
    from bssm_amd import post_correct as PC
    monkeypatch.setenv("BSSM_AMD_DEVICE_SETUP", dev)
    rng = np.random.default_rng(23)
    n, ns = 25, 9
    A = np.array([[0.8, 0.1, 0.0], [-0.2, 0.7, 0.1], [0.05, 0.0, 0.9]])
    z = np.array([1.0, 0.5, -0.3]); rd = np.array([0.3, 0.2, 0.1])
    y = np.cumsum(rng.normal(0, 0.4, n)); y[3] = np.nan
    u = M.ssm_nlg_user(y, USER_LINEAR3, theta=A.reshape(-1, order="F")[:8], known_params=np.append(z, A[2, 2]), a1=np.zeros(3),
                       P1=np.diag([0.5, 0.3, 0.2]), H=0.6, R=np.diag(rd), n_states=3)
    theta = A.reshape(-1, order="F")[:8] + 0.03 * rng.normal(size=(ns, 8))
    counts = rng.integers(1, 4, ns).astype(np.uint32)
    modes = rng.normal(0, 0.5, (3 * n, ns))  # any stored mode
    for output_type in (2, 3):
        mo = dict(theta=theta, counts=counts, modes=modes, posterior=np.zeros(ns), output_type=output_type)
        out = PC.post_correct(u, mo, 200, threads=2, is_type="is2", seed=7)
        assert np.max(np.abs(out["loglik"])) <= 1e-9 * n
        assert np.allclose(out["weights"], 1.0, rtol=1e-8)
        if output_type == 2:
            assert np.all(np.isfinite(out["alphahat"])) and np.all(np.isfinite(out["Vt"]))


@pytest.mark.parametrize("dev", ["0", "1"])
def test_user_model_three_states_post_correct_with_a_failed_draw(monkeypatch, dev):
    """one draw of the 3-state user model whose approximation is non-finite (transition coefficient overflows): that draw
    gets weight 0 / logLik -Inf as in the reference (src/model_ssm_nlg.cpp:988-990), the other draws are untouched -- its
    stand-in tables carry the 4-state padding of the good draws' tables (a batch shares padding and time-variation)"""
    from bssm_amd import post_correct as PC
    monkeypatch.setenv("BSSM_AMD_DEVICE_SETUP", dev)
    rng = np.random.default_rng(29)
    n, ns = 25, 7
    A = np.array([[0.8, 0.1, 0.0], [-0.2, 0.7, 0.1], [0.05, 0.0, 0.9]])
    z = np.array([1.0, 0.5, -0.3]); rd = np.array([0.3, 0.2, 0.1])
    y = np.cumsum(rng.normal(0, 0.4, n)); y[3] = np.nan
    u = M.ssm_nlg_user(y, USER_LINEAR3, theta=A.reshape(-1, order="F")[:8], known_params=np.append(z, A[2, 2]), a1=np.zeros(3),
                       P1=np.diag([0.5, 0.3, 0.2]), H=0.6, R=np.diag(rd), n_states=3)
    theta = A.reshape(-1, order="F")[:8] + 0.03 * rng.normal(size=(ns, 8))
    theta[4, 0] = 1e200
    counts = np.ones(ns, dtype=np.uint32)
    modes = rng.normal(0, 0.5, (3 * n, ns))
    for output_type in (2, 3):
        mo = dict(theta=theta, counts=counts, modes=modes, posterior=np.zeros(ns), output_type=output_type)
        out = PC.post_correct(u, mo, 200, threads=3, is_type="is2", seed=7)
        assert out["weights"][4] == 0.0 and np.isneginf(out["loglik"][4])
        good = np.delete(np.arange(ns), 4)
        assert np.max(np.abs(out["loglik"][good])) <= 1e-9 * n
        assert np.allclose(out["weights"][good], 1.0, rtol=1e-8)


@pytest.mark.parametrize("jit", [False, True])   # (True, round 6: compiled in process -- the host set-up evaluates the snippet through its evaluation kernel)
@pytest.mark.parametrize("path,N", [("wg", 700), ("grid", 3000)])
@pytest.mark.parametrize("which", ["linear3", "nonlin4"])
def test_user_model_psi_apf_against_the_general_oracle(oracle, monkeypatch, which, path, N, jit):
    """psi-APF of user-defined models with 3 and 4 states against orc_nlg_general_psi_* -- the restatement of
    ssm_nlg::psi_filter / log_weights / approximate over function pointers (src/model_ssm_nlg.cpp:62-143, 854-935, 971-1063),
    itself pinned in tests/test_reference_goldens.py: mode, approximate log-likelihood, ancestors, weights, log-likelihood"""
    monkeypatch.setenv("BSSM_AMD_FORCE_PATH", path)
    rng = np.random.default_rng(41)
    n = 30
    if which == "linear3":
        A = np.array([[0.8, 0.1, 0.0], [-0.2, 0.7, 0.1], [0.05, 0.0, 0.9]])
        z = np.array([1.0, 0.5, -0.3]); rd = np.array([0.3, 0.2, 0.1]); Hs = 0.6; m = 3
        a1 = np.array([0.2, -0.1, 0.0]); P1 = np.diag([0.5, 0.3, 0.2])
        y = np.cumsum(rng.normal(0, 0.4, n)); y[7] = np.nan
        th, kn = A.reshape(-1, order="F")[:8], np.append(z, A[2, 2])
        u = M.ssm_nlg_user(y, USER_LINEAR3, theta=th, known_params=kn, a1=a1, P1=P1, H=Hs, R=np.diag(rd), n_states=3, jit=jit)
        Af = lambda th, kn: np.append(th, kn[3]).reshape(3, 3, order="F")
        fns = dict(Z_fn=lambda t, a, th, kn: kn[:3] @ a, T_fn=lambda t, a, th, kn: Af(th, kn) @ a,
                   Z_gn=lambda t, a, th, kn: kn[:3], T_gn=lambda t, a, th, kn: Af(th, kn))
    else:
        th, kn = np.array([0.7]), []
        rd = np.array([0.2, 0.15, 0.3, 0.1]); Hs = 0.5; m = 4
        a1 = np.zeros(4); P1 = np.diag([0.5, 0.2, 0.3, 0.1])
        a = rng.normal(size=4) * np.sqrt(np.diag(P1)); y = np.zeros(n)
        for t in range(n):
            y[t] = a[0] + 0.5 * a[2] + 0.1 * a[0] ** 2 / (1 + a[0] ** 2) + Hs * rng.normal()
            a = np.array([a[0] + a[1], th[0] * a[1] + 0.3 * np.sin(a[2]), 0.9 * a[2] + 0.2 * a[3], 0.5 * a[3]]) + rd * rng.normal(size=4)
        y[11] = np.nan
        u = M.ssm_nlg_user(y, USER_NONLIN4, theta=th, a1=a1, P1=P1, H=Hs, R=np.diag(rd), n_states=4, jit=jit)

        def Tg(t, a, th, kn):
            J = np.zeros((4, 4))
            J[0, 0] = 1.0; J[0, 1] = 1.0; J[1, 1] = th[0]; J[1, 2] = 0.3 * np.cos(a[2]); J[2, 2] = 0.9; J[2, 3] = 0.2; J[3, 3] = 0.5
            return J
        fns = dict(Z_fn=lambda t, a, th, kn: a[0] + 0.5 * a[2] + 0.1 * a[0] * a[0] / (1.0 + a[0] * a[0]),
                   T_fn=lambda t, a, th, kn: np.array([a[0] + a[1], th[0] * a[1] + 0.3 * np.sin(a[2]), 0.9 * a[2] + 0.2 * a[3], 0.5 * a[3]]),
                   Z_gn=lambda t, a, th, kn: np.array([1.0 + 0.2 * a[0] / (1.0 + a[0] * a[0]) ** 2, 0.0, 0.5, 0.0]), T_gn=Tg)
    ref = oracle.nlg_general_psi(y, m, m, th, kn, a1, P1, H_fn=lambda t, a, th, kn: Hs, R_fn=lambda t, a, th, kn: np.diag(rd), nsim=N, seed=3, **fns)
    assert ref["ok"]
    su = F.psi_setup(u)
    assert np.allclose(np.asarray(su["mode"]).reshape(m, n, order="F"), ref["mode"], rtol=0, atol=1e-9)
    assert abs(su["approx_loglik"] - ref["approx_loglik"]) <= 1e-9 * n
    out = F.particle_smoother(u, N, method="psi", seed=3, return_indices=True)
    assert np.array_equal(out["indices"], ref["indices"])
    assert abs(out["logLik"] - ref["logLik"]) <= 1e-9 * n
    assert _rel(out["weights"], ref["weights"]) <= 1e-8
    sm = oracle.filter_smoother(ref["alpha"], ref["indices"])
    mean, cov = oracle.summary(sm)
    assert _rel(out["alphahat"], mean.T[:n]) <= 1e-8


USER_TV3 = r"""
#define BSSM_USER_NLG 1
#define BSSM_USER_JACOBIANS 1
#define BSSM_USER_TV 1
// 3 states with the reference's time argument and known_tv_params (src/model_ssm_nlg.h:10-21): a forcing term and a
// time-varying loading read from the matrix (2 x n: row 0 forcing, row 1 loading), and an explicit dependence on t
BSSM_FN void bssm_user_T(const bssm_user_tv& tv, const double* a, const double* th, const double* kn, double* out) {
  const double f = tv.known_tv[0 + tv.rows * tv.t];
  out[0] = th[0] * a[0] + 0.2 * sin(a[1]) + f;
  out[1] = 0.8 * a[1] + 0.1 * a[2] + 0.01 * tv.t;
  out[2] = 0.5 * a[2] + 0.1 * a[0];
}
BSSM_FN double bssm_user_Z(const bssm_user_tv& tv, const double* a, const double* th, const double* kn) {
  return tv.known_tv[1 + tv.rows * tv.t] * a[0] + a[2];
}
BSSM_FN void bssm_user_Tg(const bssm_user_tv& tv, const double* a, const double* th, const double* kn, double* Tg) {
  for (int q = 0; q < 9; q++) Tg[q] = 0.0;
  Tg[0 + 3 * 0] = th[0]; Tg[0 + 3 * 1] = 0.2 * cos(a[1]);
  Tg[1 + 3 * 1] = 0.8;   Tg[1 + 3 * 2] = 0.1;
  Tg[2 + 3 * 0] = 0.1;   Tg[2 + 3 * 2] = 0.5;
}
BSSM_FN void bssm_user_Zg(const bssm_user_tv& tv, const double* a, const double* th, const double* kn, double* Zg) {
  Zg[0] = tv.known_tv[1 + tv.rows * tv.t]; Zg[1] = 0.0; Zg[2] = 1.0;
}
"""


def _tv3_case(seed=51, n=28, jit=False):
    rng = np.random.default_rng(seed)
    tvm = np.vstack([0.3 * np.sin(np.arange(n) / 3.0), 1.0 + 0.5 * np.cos(np.arange(n) / 5.0)])  # 2 x n
    th = np.array([0.7])
    Rm = np.array([[0.3, 0.0, 0.0], [0.1, 0.25, 0.0], [0.05, -0.1, 0.2]])  # full lower-triangular R: R R' is not diagonal
    a1 = np.array([0.1, 0.0, -0.1]); P1 = np.array([[0.5, 0.1, 0.0], [0.1, 0.3, 0.05], [0.0, 0.05, 0.2]])
    Hs = 0.4
    y = np.cumsum(rng.normal(0, 0.3, n)); y[9] = np.nan
    fns = dict(Z_fn=lambda t, a, th, kn: tvm[1, t] * a[0] + a[2],
               T_fn=lambda t, a, th, kn: np.array([th[0] * a[0] + 0.2 * np.sin(a[1]) + tvm[0, t], 0.8 * a[1] + 0.1 * a[2] + 0.01 * t, 0.5 * a[2] + 0.1 * a[0]]),
               H_fn=lambda t, a, th, kn: Hs, R_fn=lambda t, a, th, kn: Rm)
    jac = dict(Z_gn=lambda t, a, th, kn: np.array([tvm[1, t], 0.0, 1.0]),
               T_gn=lambda t, a, th, kn: np.array([[th[0], 0.2 * np.cos(a[1]), 0.0], [0.0, 0.8, 0.1], [0.1, 0.0, 0.5]]))
    u = M.ssm_nlg_user(y, USER_TV3, theta=th, a1=a1, P1=P1, H=Hs, R=Rm, n_states=3, known_tv_params=tvm, jit=jit)
    return u, y, th, a1, P1, fns, jac


@pytest.mark.parametrize("path,N", [("wg", 600), ("grid", 3000)])
def test_user_model_time_argument_and_known_tv_params_bootstrap(oracle, monkeypatch, path, N):
    """a user model that reads the reference's time index and known_tv_params in T and Z (BSSM_USER_TV), with a full
    matrix R and a full P1: bootstrap filter against orc_nlg_general_bsf_filter (callbacks take t as in the reference)"""
    monkeypatch.setenv("BSSM_AMD_FORCE_PATH", path)
    u, y, th, a1, P1, fns, _ = _tv3_case()
    out = F.bootstrap_filter(u, N, seed=6, return_indices=True)
    ref = oracle.nlg_general_bsf_filter(y, 3, 3, th, [], a1, P1, fns["Z_fn"], fns["H_fn"], fns["T_fn"], fns["R_fn"], N, seed=6)
    assert np.array_equal(out["indices"], ref["indices"])
    assert abs(out["logLik"] - ref["logLik"]) <= 1e-9 * y.size
    assert _rel(np.transpose(out["alpha"], (1, 0, 2)), ref["alpha"]) <= 1e-9


@pytest.mark.parametrize("path,N", [("wg", 600), ("grid", 3000)])
def test_user_model_time_argument_full_R_psi_apf(oracle, monkeypatch, path, N):
    """the same model through the psi-APF: host mode search with t and known_tv_params, full constant R in the
    transition-density ratio (inverse Cholesky factor of R R'), full P1 -- against orc_nlg_general_psi_*"""
    monkeypatch.setenv("BSSM_AMD_FORCE_PATH", path)
    u, y, th, a1, P1, fns, jac = _tv3_case()
    ref = oracle.nlg_general_psi(y, 3, 3, th, [], a1, P1, nsim=N, seed=8, **fns, **jac)
    assert ref["ok"]
    su = F.psi_setup(u)
    assert np.allclose(np.asarray(su["mode"]).reshape(3, y.size, order="F"), ref["mode"], rtol=0, atol=1e-9)
    assert abs(su["approx_loglik"] - ref["approx_loglik"]) <= 1e-9 * y.size
    out = F.particle_smoother(u, N, method="psi", seed=8, return_indices=True)
    assert np.array_equal(out["indices"], ref["indices"])
    assert abs(out["logLik"] - ref["logLik"]) <= 1e-9 * y.size
    assert _rel(out["weights"], ref["weights"]) <= 1e-8


def test_user_model_tv_snippet_without_matrix_is_refused():
    u, y, th, a1, P1, fns, jac = _tv3_case()
    bad = M.BssmModel(u.classes, **{**u, "known_tv_params": None})
    with pytest.raises(Exception, match="known_tv_params"):
        F.particle_smoother(bad, 64, method="psi", seed=1)


@pytest.mark.parametrize("jit", [False, True])
@pytest.mark.parametrize("path,N", [("wg", 500), ("grid", 3000)])
@pytest.mark.parametrize("which", ["tv3_full_R", "nonlin4"])
def test_user_model_ekf_particle_filter_three_and_four_states(oracle, monkeypatch, which, path, N, jit):
    """EKF-proposal particle filter (src/model_ssm_nlg.cpp:1148-1293) of user models with 3 states (time argument,
    known_tv_params, full matrix R, full P1) and 4 states: one EKF update per particle from its ancestor with general small
    factorisations on the device (psd_chol, dmvnorm on covariance / factor) -- against orc_nlg_general_ekpf_filter"""
    monkeypatch.setenv("BSSM_AMD_FORCE_PATH", path)
    if which == "tv3_full_R":
        u, y, th, a1, P1, fns, jac = _tv3_case(seed=61, n=22, jit=jit)
        m, kn = 3, []
    else:
        rng = np.random.default_rng(63)
        n = 22
        th, kn = np.array([0.7]), []
        rd = np.array([0.2, 0.15, 0.3, 0.1]); Hs = 0.5; m = 4
        a1 = np.zeros(4); P1 = np.diag([0.5, 0.2, 0.3, 0.1])
        y = np.cumsum(rng.normal(0, 0.3, n)); y[6] = np.nan
        u = M.ssm_nlg_user(y, USER_NONLIN4, theta=th, a1=a1, P1=P1, H=Hs, R=np.diag(rd), n_states=4, jit=jit)

        def Tg(t, a, th, kn):
            J = np.zeros((4, 4))
            J[0, 0] = 1.0; J[0, 1] = 1.0; J[1, 1] = th[0]; J[1, 2] = 0.3 * np.cos(a[2]); J[2, 2] = 0.9; J[2, 3] = 0.2; J[3, 3] = 0.5
            return J
        fns = dict(Z_fn=lambda t, a, th, kn: a[0] + 0.5 * a[2] + 0.1 * a[0] * a[0] / (1.0 + a[0] * a[0]),
                   T_fn=lambda t, a, th, kn: np.array([a[0] + a[1], th[0] * a[1] + 0.3 * np.sin(a[2]), 0.9 * a[2] + 0.2 * a[3], 0.5 * a[3]]),
                   H_fn=lambda t, a, th, kn: Hs, R_fn=lambda t, a, th, kn: np.diag(rd))
        jac = dict(Z_gn=lambda t, a, th, kn: np.array([1.0 + 0.2 * a[0] / (1.0 + a[0] * a[0]) ** 2, 0.0, 0.5, 0.0]), T_gn=Tg)
    ref = oracle.nlg_general_ekpf_filter(y, m, m, th, kn, a1, P1, nsim=N, seed=4, **fns, **jac)
    out = F.particle_smoother(u, N, method="ekf", seed=4, return_indices=True)
    assert np.array_equal(out["indices"], ref["indices"])
    assert abs(out["logLik"] - ref["logLik"]) <= 1e-9 * y.size
    assert _rel(out["weights"], ref["weights"]) <= 1e-8
    sm = oracle.filter_smoother(ref["alpha"], ref["indices"])
    mean, cov = oracle.summary(sm)
    assert _rel(out["alphahat"], mean.T[:y.size]) <= 1e-8


# Two user models whose observation sd AND noise matrix depend on the state (the reference's H_fn / R_fn take alpha,
# src/model_ssm_nlg.h:12-13), with Jacobians, so that the psi-APF and the EKF particle filter apply:
#   2 states: T(a) = (th0 a0 + 0.3 sin(a1), 0.85 a1),  R(a) = [[th1 (1 + 0.5 tanh(a1)), 0], [0.1, th2 exp(0.2 a0)]],
#             Z(a) = a0 + 0.2 a1^2,  H(a) = kn0 (1 + 0.3 / (1 + a0^2))
#   3 states: T(a) = (a0 + a2, th0 a1 + 0.2 sin(a0), 0.8 a2),
#             R(a) = [[th1 exp(0.3 a1), 0, 0], [0, th2, 0], [0.2 th1 exp(0.3 a1), th3, 0.15]]  (R R' must be positive definite for
#             the psi-APF: its weights contain the transition density, src/model_ssm_nlg.cpp:919-932),
#             Z(a) = a0 + 0.5 a1,  H(a) = kn0 sqrt(1 + 0.5 a1^2)
USER_SNOISE2 = r"""
#define BSSM_USER_NLG 1
#define BSSM_USER_JACOBIANS 1
#define BSSM_USER_HFN 1
#define BSSM_USER_RFN 1
BSSM_FN void bssm_user_T(const double* a, const double* th, const double* kn, double* out) { out[0] = th[0] * a[0] + 0.3 * sin(a[1]); out[1] = 0.85 * a[1]; }
BSSM_FN double bssm_user_Z(const double* a, const double* th, const double* kn) { return a[0] + 0.2 * a[1] * a[1]; }
BSSM_FN double bssm_user_H(const double* a, const double* th, const double* kn) { return kn[0] * (1.0 + 0.3 / (1.0 + a[0] * a[0])); }
BSSM_FN void bssm_user_R(const double* a, const double* th, const double* kn, double* out) {
  out[0] = th[1] * (1.0 + 0.5 * tanh(a[1])); out[1] = 0.1; out[2] = 0.0; out[3] = th[2] * exp(0.2 * a[0]);
}
BSSM_FN void bssm_user_Tg(const double* a, const double* th, const double* kn, double* Tg) { Tg[0] = th[0]; Tg[1] = 0.0; Tg[2] = 0.3 * cos(a[1]); Tg[3] = 0.85; }
BSSM_FN void bssm_user_Zg(const double* a, const double* th, const double* kn, double* Zg) { Zg[0] = 1.0; Zg[1] = 0.4 * a[1]; }
"""
USER_SNOISE3 = r"""
#define BSSM_USER_NLG 1
#define BSSM_USER_JACOBIANS 1
#define BSSM_USER_HFN 1
#define BSSM_USER_RFN 1
BSSM_FN void bssm_user_T(const double* a, const double* th, const double* kn, double* out) { out[0] = a[0] + a[2]; out[1] = th[0] * a[1] + 0.2 * sin(a[0]); out[2] = 0.8 * a[2]; }
BSSM_FN double bssm_user_Z(const double* a, const double* th, const double* kn) { return a[0] + 0.5 * a[1]; }
BSSM_FN double bssm_user_H(const double* a, const double* th, const double* kn) { return kn[0] * sqrt(1.0 + 0.5 * a[1] * a[1]); }
BSSM_FN void bssm_user_R(const double* a, const double* th, const double* kn, double* out) {
  const double s = th[1] * exp(0.3 * a[1]);
  out[0] = s; out[1] = 0.0; out[2] = 0.2 * s;       /* first column */
  out[3] = 0.0; out[4] = th[2]; out[5] = th[3];     /* second column */
  out[6] = 0.0; out[7] = 0.0; out[8] = 0.15;        /* third column */
}
BSSM_FN void bssm_user_Tg(const double* a, const double* th, const double* kn, double* Tg) {
  Tg[0] = 1.0; Tg[1] = 0.2 * cos(a[0]); Tg[2] = 0.0;  Tg[3] = 0.0; Tg[4] = th[0]; Tg[5] = 0.0;  Tg[6] = 1.0; Tg[7] = 0.0; Tg[8] = 0.8;
}
BSSM_FN void bssm_user_Zg(const double* a, const double* th, const double* kn, double* Zg) { Zg[0] = 1.0; Zg[1] = 0.5; Zg[2] = 0.0; }
"""


def _snoise_case(which, seed, n):
    rng = np.random.default_rng(seed)
    if which == "snoise2":
        m, k = 2, 2
        th, kn = np.array([0.9, 0.25, 0.15]), np.array([0.5])
        fns = dict(Z_fn=lambda t, a, th, kn: a[0] + 0.2 * a[1] ** 2,
                   H_fn=lambda t, a, th, kn: kn[0] * (1.0 + 0.3 / (1.0 + a[0] ** 2)),
                   T_fn=lambda t, a, th, kn: np.array([th[0] * a[0] + 0.3 * np.sin(a[1]), 0.85 * a[1]]),
                   R_fn=lambda t, a, th, kn: np.array([[th[1] * (1.0 + 0.5 * np.tanh(a[1])), 0.0], [0.1, th[2] * np.exp(0.2 * a[0])]]))
        jac = dict(Z_gn=lambda t, a, th, kn: np.array([1.0, 0.4 * a[1]]),
                   T_gn=lambda t, a, th, kn: np.array([[th[0], 0.3 * np.cos(a[1])], [0.0, 0.85]]))
        a1 = np.array([0.3, -0.2]); P1 = np.array([[0.4, 0.1], [0.1, 0.3]])
        code = USER_SNOISE2
    else:
        m, k = 3, 3
        th, kn = np.array([0.8, 0.3, 0.2, 0.1]), np.array([0.6])

        def R(t, a, th, kn):
            s = th[1] * np.exp(0.3 * a[1])
            return np.array([[s, 0.0, 0.0], [0.0, th[2], 0.0], [0.2 * s, th[3], 0.15]])
        fns = dict(Z_fn=lambda t, a, th, kn: a[0] + 0.5 * a[1],
                   H_fn=lambda t, a, th, kn: kn[0] * np.sqrt(1.0 + 0.5 * a[1] ** 2),
                   T_fn=lambda t, a, th, kn: np.array([a[0] + a[2], th[0] * a[1] + 0.2 * np.sin(a[0]), 0.8 * a[2]]), R_fn=R)
        jac = dict(Z_gn=lambda t, a, th, kn: np.array([1.0, 0.5, 0.0]),
                   T_gn=lambda t, a, th, kn: np.array([[1.0, 0.0, 1.0], [0.2 * np.cos(a[0]), th[0], 0.0], [0.0, 0.0, 0.8]]))
        a1 = np.array([0.2, 0.0, 0.1]); P1 = np.diag([0.5, 0.3, 0.2]); P1[0, 1] = P1[1, 0] = 0.1
        code = USER_SNOISE3
    a = a1.copy(); y = np.zeros(n)
    for t in range(n):
        y[t] = fns["Z_fn"](t, a, th, kn) + fns["H_fn"](t, a, th, kn) * rng.normal()
        a = fns["T_fn"](t, a, th, kn) + fns["R_fn"](t, a, th, kn) @ rng.normal(size=k)
    y[5] = np.nan
    u = M.ssm_nlg_user(y, code, theta=th, known_params=kn, a1=a1, P1=P1, n_states=m, H=1.0,
                       R=np.zeros((m, k)))  # (H and R only fix the shapes here: the snippet's H_fn / R_fn are what is evaluated)
    return u, y, m, k, th, kn, a1, P1, fns, jac


@pytest.mark.parametrize("path,N", [("wg", 600), ("grid", 3000)])
@pytest.mark.parametrize("which", ["snoise2", "snoise3"])
def test_user_model_psi_apf_with_state_dependent_H_and_R(oracle, monkeypatch, which, path, N):
    """psi-APF of user models whose H_fn and R_fn depend on the state: the approximating model takes H_t = H_fn(mode_t) and
    R_t = R_fn(mode_t) per time point (src/model_ssm_nlg.cpp:62-143), and neither density ratio of the weight has equal
    covariances any more (:880-935) -- mode, approximate log-likelihood, ancestors, weights and log-likelihood against
    orc_nlg_general_psi_*, the restatement over function pointers"""
    monkeypatch.setenv("BSSM_AMD_FORCE_PATH", path)
    u, y, m, k, th, kn, a1, P1, fns, jac = _snoise_case(which, seed=71, n=26)
    n = y.size
    ref = oracle.nlg_general_psi(y, m, k, th, kn, a1, P1, nsim=N, seed=6, **fns, **jac)
    assert ref["ok"]
    su = F.psi_setup(u)
    assert np.allclose(np.asarray(su["mode"]).reshape(m, n, order="F"), ref["mode"], rtol=0, atol=1e-9)
    assert abs(su["approx_loglik"] - ref["approx_loglik"]) <= 1e-9 * n
    out = F.particle_smoother(u, N, method="psi", seed=6, return_indices=True)
    assert np.array_equal(out["indices"], ref["indices"])
    assert abs(out["logLik"] - ref["logLik"]) <= 1e-9 * n
    assert _rel(out["weights"], ref["weights"]) <= 1e-8
    mean, cov = oracle.summary(oracle.filter_smoother(ref["alpha"], ref["indices"]))
    assert _rel(out["alphahat"], mean.T[:n]) <= 1e-8
    # the state dependence matters in this case: with H and R frozen at their values in a1 the log-likelihood moves
    frozen = oracle.nlg_general_psi(y, m, k, th, kn, a1, P1, nsim=N, seed=6, **{**fns, "H_fn": lambda t, a, th_, kn_: fns["H_fn"](t, a1, th_, kn_),
                                                                         "R_fn": lambda t, a, th_, kn_: fns["R_fn"](t, a1, th_, kn_)}, **jac)
    assert abs(frozen["logLik"] - ref["logLik"]) > 1e-3


@pytest.mark.parametrize("path,N", [("wg", 600), ("grid", 3000)])
@pytest.mark.parametrize("which", ["snoise2", "snoise3"])
def test_user_model_ekf_particle_filter_with_state_dependent_H_and_R(oracle, monkeypatch, which, path, N):
    """EKF-proposal particle filter of the same models: R_fn at the ancestor, H_fn at the predicted state inside the EKF
    update and at the particle in the weight (src/model_ssm_nlg.cpp:1148-1293) -- against orc_nlg_general_ekpf_filter"""
    monkeypatch.setenv("BSSM_AMD_FORCE_PATH", path)
    u, y, m, k, th, kn, a1, P1, fns, jac = _snoise_case(which, seed=73, n=22)
    ref = oracle.nlg_general_ekpf_filter(y, m, k, th, kn, a1, P1, nsim=N, seed=8, **fns, **jac)
    out = F.particle_smoother(u, N, method="ekf", seed=8, return_indices=True)
    assert np.array_equal(out["indices"], ref["indices"])
    assert abs(out["logLik"] - ref["logLik"]) <= 1e-9 * y.size
    assert _rel(out["weights"], ref["weights"]) <= 1e-8
    mean, cov = oracle.summary(oracle.filter_smoother(ref["alpha"], ref["indices"]))
    assert _rel(out["alphahat"], mean.T[:y.size]) <= 1e-8


@pytest.mark.parametrize("dev", ["0", "1"])
def test_user_model_state_dependent_noise_post_correct_at_the_mode(oracle, monkeypatch, dev):
    """IS post-correction (psi) of draws of the 3-state model with state-dependent H / R against the general oracle run per
    draw at the stored mode (approximate_for_is, src/model_ssm_nlg.cpp:146-176): log-weights per draw, host set-up and the
    device smoother behind the host linearisation"""
    from bssm_amd import post_correct as PC
    monkeypatch.setenv("BSSM_AMD_DEVICE_SETUP", dev)
    u, y, m, k, th, kn, a1, P1, fns, jac = _snoise_case("snoise3", seed=75, n=18)
    n, ns, N = y.size, 5, 150
    rng = np.random.default_rng(5)
    theta = th + 0.02 * rng.normal(size=(ns, th.size))
    counts = np.ones(ns, dtype=np.uint32)
    modes = np.stack([oracle.nlg_general_psi(y, m, k, theta[i], kn, a1, P1, nsim=8, seed=1, filter=False, **fns, **jac)["mode"].reshape(-1, order="F")
                      for i in range(ns)], axis=1)
    mo = dict(theta=theta, counts=counts, modes=modes, posterior=np.zeros(ns), output_type=3)
    out = PC.post_correct(u, mo, N, threads=2, is_type="is2", seed=9)
    for i in range(ns):
        ref = oracle.nlg_general_psi(y, m, k, theta[i], kn, a1, P1, nsim=N, seed=9, stream=i, is_mode=modes[:, i], **fns, **jac)
        assert abs(out["loglik"][i] - ref["logLik"]) <= 1e-9 * n, (i, out["loglik"][i], ref["logLik"])


# ---------------------------------------------------------------- round 4: user models with more than four states (bootstrap paths)
USER_SIX = r"""
#define BSSM_USER_NLG 1
BSSM_FN void bssm_user_T(const double* a, const double* th, const double* kn, double* out) {
  out[0] = a[0] + a[1]; out[1] = 0.9 * a[1] + 0.1 * sin(a[2]); out[2] = th[0] * a[2] + 0.05 * a[3];
  out[3] = 0.8 * a[3] - 0.1 * a[4]; out[4] = 0.7 * a[4] + 0.2 * tanh(a[5]); out[5] = 0.5 * a[5];
}
BSSM_FN double bssm_user_Z(const double* a, const double* th, const double* kn) { return a[0] + 0.5 * a[2] + 0.1 * a[5] * a[5]; }
"""


@pytest.mark.parametrize("path,N", [("wg", 200), ("wg", 1200), ("grid", 4000)])
def test_user_model_six_states_bootstrap_against_the_general_oracle(oracle, monkeypatch, path, N):
    """A user model with SIX states and a constant 6 x 4 matrix R (the MP = 8 functor kernels; the reference's function-pointer
    interface has no state limit, src/model_ssm_nlg.h:10-21): bootstrap filter and smoother against orc_nlg_general_bsf_filter --
    ancestors identical, log-likelihood to 1e-9 n, states and weights to 1e-9 / 1e-8; the psi-APF and the EKF particle filter stay
    at four states and say so."""
    monkeypatch.setenv("BSSM_AMD_FORCE_PATH", path)
    rng = np.random.default_rng(12)
    n, m, k = 30, 6, 4
    th = np.array([0.85]); kn = np.zeros(0)
    Rm = np.zeros((m, k)); Rm[0, 0] = 0.05; Rm[1, 0] = 0.02; Rm[2, 1] = 0.3; Rm[3, 1] = -0.1; Rm[3, 2] = 0.2; Rm[4, 2] = 0.1; Rm[5, 3] = 0.4
    Z = lambda t, a, th_, kn_: a[0] + 0.5 * a[2] + 0.1 * a[5] ** 2
    H = lambda t, a, th_, kn_: 0.6
    T = lambda t, a, th_, kn_: np.array([a[0] + a[1], 0.9 * a[1] + 0.1 * np.sin(a[2]), th_[0] * a[2] + 0.05 * a[3], 0.8 * a[3] - 0.1 * a[4],
                                         0.7 * a[4] + 0.2 * np.tanh(a[5]), 0.5 * a[5]])
    R = lambda t, a, th_, kn_: Rm
    a = np.array([1.0, 0.05, 0.3, -0.2, 0.1, 0.4]); y = np.zeros(n)
    for t in range(n):
        y[t] = Z(t, a, th, kn) + 0.6 * rng.normal()
        a = T(t, a, th, kn) + Rm @ rng.normal(size=k)
    y[9] = np.nan
    a1 = np.array([1.0, 0.0, 0.3, 0.0, 0.0, 0.4])
    P1 = np.diag([0.5, 0.01, 0.2, 0.1, 0.0, 0.3]); P1[0, 2] = P1[2, 0] = 0.1
    u = M.ssm_nlg_user(y, USER_SIX, theta=th, known_params=kn, a1=a1, P1=P1, H=0.6, n_states=m, R=Rm)
    out = F.bootstrap_filter(u, N, seed=7, return_indices=True)
    ref = oracle.nlg_general_bsf_filter(y, m, k, th, kn, a1, P1, Z, H, T, R, N, seed=7)
    assert np.array_equal(out["indices"], ref["indices"])
    assert abs(out["logLik"] - ref["logLik"]) <= 1e-9 * n
    assert _rel(np.transpose(out["alpha"], (1, 0, 2)), ref["alpha"]) <= 1e-9
    assert _rel(out["weights"], ref["weights"]) <= 1e-8
    sm = F.particle_smoother(u, N, method="bsf", seed=7)
    assert abs(sm["logLik"] - ref["logLik"]) <= 1e-9 * n
    ah, Vt = oracle.summary(oracle.filter_smoother(ref["alpha"], ref["indices"]))
    assert _rel(sm["alphahat"], ah.T[:n]) <= 1e-8 and _rel(sm["Vt"], Vt[:, :, :n]) <= 1e-7
    if path == "wg" and N == 200:
        with pytest.raises(Exception, match="Jacobians|BSSM_USER_JACOBIANS"):
            F.particle_smoother(u, N, method="psi", seed=7)


USER_SIX_J = r"""
#define BSSM_USER_NLG 1
#define BSSM_USER_JACOBIANS 1
BSSM_FN void bssm_user_T(const double* a, const double* th, const double* kn, double* out) {
  out[0] = a[0] + a[1]; out[1] = 0.9 * a[1] + 0.1 * sin(a[2]); out[2] = th[0] * a[2] + 0.05 * a[3];
  out[3] = 0.8 * a[3] - 0.1 * a[4]; out[4] = 0.7 * a[4] + 0.2 * tanh(a[5]); out[5] = 0.5 * a[5];
}
BSSM_FN double bssm_user_Z(const double* a, const double* th, const double* kn) { return a[0] + 0.5 * a[2] + 0.1 * a[5] * a[5]; }
BSSM_FN void bssm_user_Tg(const double* a, const double* th, const double* kn, double* Tg) {
  for (int q = 0; q < 36; q++) Tg[q] = 0.0;
  Tg[0 + 6 * 0] = 1.0; Tg[0 + 6 * 1] = 1.0;
  Tg[1 + 6 * 1] = 0.9; Tg[1 + 6 * 2] = 0.1 * cos(a[2]);
  Tg[2 + 6 * 2] = th[0]; Tg[2 + 6 * 3] = 0.05;
  Tg[3 + 6 * 3] = 0.8; Tg[3 + 6 * 4] = -0.1;
  Tg[4 + 6 * 4] = 0.7; const double th5 = tanh(a[5]); Tg[4 + 6 * 5] = 0.2 * (1.0 - th5 * th5);
  Tg[5 + 6 * 5] = 0.5;
}
BSSM_FN void bssm_user_Zg(const double* a, const double* th, const double* kn, double* Zg) {
  Zg[0] = 1.0; Zg[1] = 0.0; Zg[2] = 0.5; Zg[3] = 0.0; Zg[4] = 0.0; Zg[5] = 0.2 * a[5];
}
"""


def _six_case(seed=12, n=30, jit=False):
    rng = np.random.default_rng(seed)
    m = 6
    th = np.array([0.85]); kn = np.zeros(0)
    # a full-rank 6 x 6 lower-triangular R: R R' is positive definite and not diagonal
    Rm = np.diag([0.05, 0.1, 0.3, 0.2, 0.15, 0.4]); Rm[1, 0] = 0.02; Rm[3, 2] = -0.1; Rm[4, 3] = 0.1; Rm[5, 1] = 0.05
    Z = lambda t, a, th_, kn_: a[0] + 0.5 * a[2] + 0.1 * a[5] ** 2
    H = lambda t, a, th_, kn_: 0.6
    T = lambda t, a, th_, kn_: np.array([a[0] + a[1], 0.9 * a[1] + 0.1 * np.sin(a[2]), th_[0] * a[2] + 0.05 * a[3], 0.8 * a[3] - 0.1 * a[4],
                                         0.7 * a[4] + 0.2 * np.tanh(a[5]), 0.5 * a[5]])
    R = lambda t, a, th_, kn_: Rm

    def Tg(t, a, th_, kn_):
        J = np.zeros((6, 6))
        J[0, 0] = 1.0; J[0, 1] = 1.0; J[1, 1] = 0.9; J[1, 2] = 0.1 * np.cos(a[2]); J[2, 2] = th_[0]; J[2, 3] = 0.05
        J[3, 3] = 0.8; J[3, 4] = -0.1; J[4, 4] = 0.7; J[4, 5] = 0.2 * (1.0 - np.tanh(a[5]) ** 2); J[5, 5] = 0.5
        return J
    Zg = lambda t, a, th_, kn_: np.array([1.0, 0.0, 0.5, 0.0, 0.0, 0.2 * a[5]])
    a = np.array([1.0, 0.05, 0.3, -0.2, 0.1, 0.4]); y = np.zeros(n)
    for t in range(n):
        y[t] = Z(t, a, th, kn) + 0.6 * rng.normal()
        a = T(t, a, th, kn) + Rm @ rng.normal(size=m)
    y[9] = np.nan
    a1 = np.array([1.0, 0.0, 0.3, 0.0, 0.0, 0.4])
    P1 = np.diag([0.5, 0.01, 0.2, 0.1, 0.05, 0.3]); P1[0, 2] = P1[2, 0] = 0.1
    u = M.ssm_nlg_user(y, USER_SIX_J, theta=th, known_params=kn, a1=a1, P1=P1, H=0.6, n_states=m, R=Rm, jit=jit)
    return u, y, m, th, kn, a1, P1, dict(Z_fn=Z, H_fn=H, T_fn=T, R_fn=R, Z_gn=Zg, T_gn=Tg)


@pytest.mark.parametrize("jit", [False, True])
@pytest.mark.parametrize("path,N", [("wg", 300), ("wg", 1200), ("grid", 4000)])
def test_user_model_six_states_psi_apf_against_the_general_oracle(oracle, monkeypatch, path, N, jit):
    """psi-APF of a user model with SIX states and a full lower-triangular R (late round 4: the MP = 8 functor kernels carry the
    psi-APF too; src/model_ssm_nlg.cpp:62-143, 854-935, 971-1063 over function pointers): mode, approximate log-likelihood,
    ancestors, weights, log-likelihood and smoothed means against orc_nlg_general_psi_*."""
    monkeypatch.setenv("BSSM_AMD_FORCE_PATH", path)
    u, y, m, th, kn, a1, P1, fns = _six_case(jit=jit)
    n = y.size
    ref = oracle.nlg_general_psi(y, m, m, th, kn, a1, P1, nsim=N, seed=3, **fns)
    assert ref["ok"]
    su = F.psi_setup(u)
    assert np.allclose(np.asarray(su["mode"]).reshape(m, n, order="F"), ref["mode"], rtol=0, atol=1e-9)
    assert abs(su["approx_loglik"] - ref["approx_loglik"]) <= 1e-9 * n
    out = F.particle_smoother(u, N, method="psi", seed=3, return_indices=True)
    assert np.array_equal(out["indices"], ref["indices"])
    assert abs(out["logLik"] - ref["logLik"]) <= 1e-9 * n
    assert _rel(out["weights"], ref["weights"]) <= 1e-8
    mean, cov = oracle.summary(oracle.filter_smoother(ref["alpha"], ref["indices"]))
    assert _rel(out["alphahat"], mean.T[:n]) <= 1e-8


@pytest.mark.parametrize("path,N", [("wg", 300), ("grid", 3000)])
def test_user_model_six_states_ekf_particle_filter_against_the_general_oracle(oracle, monkeypatch, path, N):
    """EKF-proposal particle filter (src/model_ssm_nlg.cpp:1148-1293) of the six-state user model: one 6 x 6 EKF update per particle
    with the general small factorisations of the MP = 8 functor kernels -- against orc_nlg_general_ekpf_filter: ancestors, weights,
    log-likelihood, smoothed means; nine states are refused with a clear message"""
    monkeypatch.setenv("BSSM_AMD_FORCE_PATH", path)
    u, y, m, th, kn, a1, P1, fns = _six_case()
    ref = oracle.nlg_general_ekpf_filter(y, m, m, th, kn, a1, P1, nsim=N, seed=4, **fns)
    out = F.particle_smoother(u, N, method="ekf", seed=4, return_indices=True)
    assert np.array_equal(out["indices"], ref["indices"])
    assert abs(out["logLik"] - ref["logLik"]) <= 1e-9 * y.size
    assert _rel(out["weights"], ref["weights"]) <= 1e-8
    mean, cov = oracle.summary(oracle.filter_smoother(ref["alpha"], ref["indices"]))
    assert _rel(out["alphahat"], mean.T[:y.size]) <= 1e-8
    if path == "wg":
        with pytest.raises(ValueError, match="1 ... 16"):
            M.ssm_nlg_user(np.zeros(5), USER_SIX_J, theta=th, a1=np.zeros(17), P1=np.eye(17), H=0.6, n_states=17, R=np.eye(17))


@pytest.mark.parametrize("dev", ["0", "1"])
def test_user_model_six_states_post_correct_psi_against_the_general_oracle(oracle, monkeypatch, dev):
    """IS post-correction (is2) of the six-state user model over theta-draws with their stored modes (approximate_for_is,
    src/model_ssm_nlg.cpp:146-176; src/approx_mcmc.cpp:364-417): every draw's psi-APF log-likelihood against
    orc_nlg_general_psi_filter on stream = draw index, weights = exp(logLik), through the host set-up and the device smoother"""
    from bssm_amd import post_correct as PC
    monkeypatch.setenv("BSSM_AMD_DEVICE_SETUP", dev)
    u, y, m, th0, kn, a1, P1, fns = _six_case()
    n, ns, N = y.size, 5, 300
    rng = np.random.default_rng(5)
    theta = th0 + 0.03 * rng.normal(size=(ns, 1))
    counts = rng.integers(1, 4, ns).astype(np.uint32)
    sus = [oracle.nlg_general_psi(y, m, m, th, kn, a1, P1, nsim=N, seed=9, filter=False, **fns) for th in theta]
    modes = np.stack([s["mode"].reshape(-1, order="F") for s in sus], axis=1)
    mo = dict(theta=theta, counts=counts, modes=modes, posterior=np.zeros(ns), output_type=2)
    out = PC.post_correct(u, mo, N, threads=2, is_type="is2", seed=9)
    w = np.zeros(ns); ah = np.zeros((m, n + 1, ns))
    for i, th in enumerate(theta):
        ref = oracle.nlg_general_psi(y, m, m, th, kn, a1, P1, nsim=N, seed=9, stream=i, is_mode=modes[:, i].reshape((m, n), order="F"), **fns)
        assert ref["ok"]
        assert abs(out["loglik"][i] - ref["logLik"]) <= 1e-9 * n
        w[i] = np.exp(ref["logLik"])
        ah[:, :, i], _ = oracle.summary(oracle.filter_smoother(ref["alpha"], ref["indices"]))
    assert np.allclose(out["weights"], w, rtol=1e-8)
    wc = w * counts
    assert np.allclose(out["alphahat"], (np.tensordot(ah, wc, axes=([2], [0])) / wc.sum()).T[:out["alphahat"].shape[0]], rtol=1e-7, atol=1e-9)


# ---------------------------------------------------------------- functor models beyond the row family's 8192 particles: round-2 step kernel
@pytest.mark.parametrize("name,N,method", [("nlg_growth", 20000, "bsf"), ("nlg_growth", 50000, "psi"), ("nlg_growth", 100000, "bsf"),
                                           ("sde_poisson_ou", 40000, "bsf"), ("nlg_sin_exp", 100000, "psi"), ("nlg_growth", 30000, "ekf")])
def test_functor_models_on_the_round2_step_kernel(oracle, name, N, method):
    """ssm_nlg / ssm_sde filters larger than one workgroup ran the round-1 step kernel through round 4 (seven barriers, a search per
    particle); since round 5 they take pf_grid_step_v2_fn (one search per wave, wave-local combine, weights exchange) at one, two
    or four particles per lane: the reference's own study runs the growth model at N = 1e5 (vignettes/psi_pf_experiments/gm_truth.rds)"""
    from bssm_amd.plan import FilterPlan
    model, L = MODELS[name]
    n = model.n
    if method == "bsf":
        out = F.bootstrap_filter(model, N, seed=12, return_indices=True, L=L)
        ref = oracle.fn_bsf_filter({**model, "L": L or 0}, N, seed=12)
    elif method == "psi":
        out = F.particle_smoother(model, N, method="psi", seed=12, return_indices=True)
        ref = oracle.fn_psi_filter(model, N, seed=12)
    else:
        out = F.ekpf_filter(model, N, seed=12, return_indices=True)
        ref = oracle.fn_ekpf_filter(model, N, seed=12)
    assert FilterPlan.last_kernel().startswith("pf_grid_step_v2_fn<"), FilterPlan.last_kernel()
    assert np.array_equal(out["indices"], ref["indices"])
    assert abs(out["logLik"] - ref["logLik"]) <= 1e-9 * n
    if "weights" in out and method != "psi":
        assert _rel(out["weights"], ref["weights"]) <= 1e-8


# ---------------------------------------------------------------- user models compiled IN PROCESS (hiprtc): bssm_user_model_jit
def test_user_model_compiled_in_process_equals_the_hipcc_built_library(oracle, tmp_path, monkeypatch):
    """The reference compiles a user's model inside the session (Rcpp::sourceCpp, R/cpp_example_models.R:762).  jit=True compiles
    the snippet's device functions with hiprtc -- no hipcc, no process, no shared library -- into exactly the kernel variant a call
    needs: bootstrap filter / smoother / log-likelihood batches of the two-state growth model, the six-state model and the SDE model
    give the SAME BITS as the hipcc-built library (same kernel templates), on the workgroup-resident shape and on the row family;
    the psi-APF and the EKF particle filter, whose set-up runs the snippet on the host, refuse such a model with a message."""
    import os
    import time
    from bssm_amd import _lib
    from bssm_amd.plan import FilterPlan
    monkeypatch.setenv("BSSM_AMD_USER_CACHE", str(tmp_path))   # an empty code-object cache: the compile is really cold
    os.chmod(tmp_path, 0o700)
    b, _ = MODELS["nlg_growth"]
    th, kp = np.asarray(b["theta"]), np.asarray(b["known_params"])
    mk = lambda jit: M.ssm_nlg_user(b["y"], USER_GROWTH, theta=th, known_params=kp, a1=kp[2:4], P1=np.diag(kp[4:6]), H=np.exp(th[0]),
                                     R_diag=np.exp(th[1:3]), jit=jit)
    t0 = time.perf_counter()
    uj = mk(True)
    first = F.bootstrap_filter(uj, 300, seed=4, return_indices=True)
    cold = time.perf_counter() - t0
    assert "[hiprtc]" in FilterPlan.last_kernel(), FilterPlan.last_kernel()
    assert cold < 6.0, f"cold in-process build + first filter call took {cold:.1f} s"
    print(f"in-process build of a two-state model + first filter call: {cold:.2f} s")
    monkeypatch.delenv("BSSM_AMD_USER_CACHE")
    uh = mk(False)
    ref = oracle.fn_bsf_filter({**b, "L": 0}, 300, seed=4)
    assert np.array_equal(first["indices"], ref["indices"]) and abs(first["logLik"] - ref["logLik"]) <= 1e-9 * b.n
    for N, path in ((300, None), (1500, None), (5000, None), (3000, "row"), (12000, None), (40000, None)):
        if path: monkeypatch.setenv("BSSM_AMD_FORCE_PATH", path)
        oj = F.bootstrap_filter(uj, N, seed=6, return_indices=True)
        kern = FilterPlan.last_kernel()
        if path: monkeypatch.delenv("BSSM_AMD_FORCE_PATH")
        assert "[hiprtc]" in kern, kern
        if N <= 8192:   # (the hipcc-built library takes the step kernel above the workgroup-resident sizes by default, the in-process one the row family)
            monkeypatch.setenv("BSSM_AMD_FORCE_PATH", "row" if ("pf_row_step" in kern) else "wg")
            oh = F.bootstrap_filter(uh, N, seed=6, return_indices=True)
            monkeypatch.delenv("BSSM_AMD_FORCE_PATH")
            for k in ("indices", "weights", "alpha"):
                assert np.array_equal(oj[k], oh[k]), (N, k)
            assert oj["logLik"] == oh["logLik"]
        elif N > 16384:   # beyond the row family: the step kernel (pf_grid_step_v2_fn) on both routes since round 5 -- same template, same bits
            assert "pf_grid_step_v2_fn" in kern, kern
            oh = F.bootstrap_filter(uh, N, seed=6, return_indices=True)
            for k in ("indices", "weights", "alpha"):
                assert np.array_equal(oj[k], oh[k]), (N, k)
            assert oj["logLik"] == oh["logLik"]
            r2 = oracle.fn_bsf_filter({**b, "L": 0}, N, seed=6)
            assert np.array_equal(oj["indices"], r2["indices"]) and abs(oj["logLik"] - r2["logLik"]) <= 1e-9 * b.n
        else:
            r2 = oracle.fn_bsf_filter({**b, "L": 0}, N, seed=6)
            assert np.array_equal(oj["indices"], r2["indices"]) and abs(oj["logLik"] - r2["logLik"]) <= 1e-9 * b.n
    sj, sh = F.particle_smoother(uj, 700, method="bsf", seed=2), F.particle_smoother(uh, 700, method="bsf", seed=2)
    assert np.array_equal(sj["alphahat"], sh["alphahat"]) and np.array_equal(sj["Vt"], sh["Vt"])
    assert np.array_equal(F.logLik_batch(uj, 200, 40, method="bsf", seed=3), F.logLik_batch(uh, 200, 40, method="bsf", seed=3))
    # round 6: the psi-APF and the EKF particle filter of the in-process model too -- its host set-up (mode search, linearisation) evaluates the
    # snippet through a one-thread kernel instead of host code (device libm instead of the host's: the last place may differ, nothing else)
    for method in ("psi", "ekf"):
        pj, ph = F.particle_smoother(uj, 300, method=method, seed=1, return_indices=True), F.particle_smoother(uh, 300, method=method, seed=1, return_indices=True)
        assert np.array_equal(pj["indices"], ph["indices"]) and abs(pj["logLik"] - ph["logLik"]) <= 1e-10 * b.n, method
        assert np.allclose(pj["alphahat"], ph["alphahat"], rtol=1e-9, atol=1e-10), method
    # an SDE model (Milstein sub-steps)
    g, L = MODELS["sde_gbm"]
    gj = M.ssm_sde_user(g["y"], USER_GBM, theta=g["theta"], x0=g["x0"], positive=True, jit=True)
    gh = M.ssm_sde_user(g["y"], USER_GBM, theta=g["theta"], x0=g["x0"], positive=True)
    a, c = F.bootstrap_filter(gj, 400, seed=3, return_indices=True, L=L), F.bootstrap_filter(gh, 400, seed=3, return_indices=True, L=L)
    assert np.array_equal(a["indices"], c["indices"]) and a["logLik"] == c["logLik"]


def test_user_model_in_process_compile_error_is_reported():
    with pytest.raises(Exception, match="compiling the user model in process failed"):
        bad = M.ssm_nlg_user(np.zeros(5), "#define BSSM_USER_NLG 1\nthis is not C++", theta=[0.0], a1=[0.0], P1=[[1.0]], H=1.0, R_diag=[1.0], jit=True)
        F.bootstrap_filter(bad, 10, seed=1)


# ---------------------------------------------------------------- user models of 9 ... 16 states (round 5): bootstrap paths
USER_TWELVE = r"""
#define BSSM_USER_NLG 1
BSSM_FN void bssm_user_T(const double* a, const double* th, const double* kn, double* out) {
  for (int j = 0; j < 11; j++) out[j] = (0.5 + 0.03 * j) * a[j] + 0.1 * sin(a[j + 1]);
  out[11] = th[0] * a[11] + 0.05 * tanh(a[0]);
}
BSSM_FN double bssm_user_Z(const double* a, const double* th, const double* kn) {
  double z = 0.0;
  for (int j = 0; j < 12; j += 3) z += a[j];
  return z + 0.1 * a[7] * a[7];
}
"""


def _twelve_case(seed=21, n=24):
    rng = np.random.default_rng(seed)
    m, k = 12, 5
    th = np.array([0.8]); kn = np.zeros(0)
    Rm = np.zeros((m, k))
    for j in range(m): Rm[j, j % k] = 0.1 + 0.02 * j
    Rm[3, 1] = -0.05; Rm[10, 4] = 0.07
    Z = lambda t, a, th_, kn_: a[0] + a[3] + a[6] + a[9] + 0.1 * a[7] ** 2
    H = lambda t, a, th_, kn_: 0.5
    T = lambda t, a, th_, kn_: np.array([(0.5 + 0.03 * j) * a[j] + 0.1 * np.sin(a[j + 1]) for j in range(11)] + [th_[0] * a[11] + 0.05 * np.tanh(a[0])])
    R = lambda t, a, th_, kn_: Rm
    a = 0.3 * rng.normal(size=m); y = np.zeros(n)
    for t in range(n):
        y[t] = Z(t, a, th, kn) + 0.5 * rng.normal()
        a = T(t, a, th, kn) + Rm @ rng.normal(size=k)
    y[5] = np.nan
    a1 = 0.1 * np.arange(m) - 0.5
    P1 = np.diag(0.05 + 0.02 * np.arange(m)); P1[11, 11] = 0.0; P1[0, 4] = P1[4, 0] = 0.03
    return dict(y=y, m=m, k=k, th=th, kn=kn, Rm=Rm, Z=Z, H=H, T=T, R=R, a1=a1, P1=P1, n=n)


USER_SIXTEEN = r"""
#define BSSM_USER_NLG 1
BSSM_FN void bssm_user_T(const double* a, const double* th, const double* kn, double* out) {
  for (int j = 0; j < 16; j++) out[j] = (0.4 + 0.03 * j) * a[j] + 0.1 * sin(a[(j + 5) & 15]) + th[0] * 0.01 * j;
}
BSSM_FN double bssm_user_Z(const double* a, const double* th, const double* kn) {
  double z = 0.0;
  for (int j = 0; j < 16; j++) z += (j & 1 ? -0.2 : 0.3) * a[j];
  return z;
}
"""


@pytest.mark.parametrize("path,N", [("wg", 400), ("row", 1000), ("grid", 18000)])
def test_user_model_sixteen_states_full_noise_matrix(oracle, monkeypatch, path, N):
    """SIXTEEN states, a dense 16 x 16 noise matrix, a dense P1: the padded dimension IS the state dimension (no padding rows to hide an
    indexing slip); bootstrap filter against orc_nlg_general_bsf_filter on the three kernel shapes"""
    rng = np.random.default_rng(33)
    m, k, n = 16, 16, 14
    th = np.array([0.7]); kn = np.zeros(0)
    Rm = 0.05 * rng.normal(size=(m, k)) + 0.15 * np.eye(m)
    Z = lambda t, a, th_, kn_: float(sum((-0.2 if j & 1 else 0.3) * a[j] for j in range(16)))
    H = lambda t, a, th_, kn_: 0.7
    T = lambda t, a, th_, kn_: np.array([(0.4 + 0.03 * j) * a[j] + 0.1 * np.sin(a[(j + 5) & 15]) + th_[0] * 0.01 * j for j in range(16)])
    R = lambda t, a, th_, kn_: Rm
    y = rng.normal(size=n); y[3] = np.nan
    a1 = 0.1 * rng.normal(size=m)
    Lp = 0.2 * rng.normal(size=(m, m)); P1 = Lp @ Lp.T + 0.05 * np.eye(m)
    u = M.ssm_nlg_user(y, USER_SIXTEEN, theta=th, known_params=kn, a1=a1, P1=P1, H=0.7, n_states=m, R=Rm)
    if path != "grid": monkeypatch.setenv("BSSM_AMD_FORCE_PATH", path)
    out = F.bootstrap_filter(u, N, seed=5, return_indices=True)
    ref = oracle.nlg_general_bsf_filter(y, m, k, th, kn, a1, P1, Z, H, T, R, N, seed=5)
    assert np.array_equal(out["indices"], ref["indices"])
    assert abs(out["logLik"] - ref["logLik"]) <= 1e-9 * n
    assert _rel(np.transpose(out["alpha"], (1, 0, 2)), ref["alpha"]) <= 1e-9
    assert _rel(out["weights"], ref["weights"]) <= 1e-8


@pytest.mark.parametrize("path,N", [("wg", 300), ("row", 2000), ("grid", 20000)])
def test_user_model_twelve_states_bootstrap_against_the_general_oracle(oracle, monkeypatch, path, N):
    """A user model with TWELVE states and a constant 12 x 5 matrix R (round 5: the MP = 16 functor kernels; the reference's
    function-pointer interface has no state limit, src/model_ssm_nlg.h:10-21): bootstrap filter and smoother against
    orc_nlg_general_bsf_filter on the workgroup-resident shape, the row family and the round-1 step kernel (above 16 384 particles)
    -- ancestors identical, log-likelihood to 1e-9 n; the hiprtc-compiled model gives the bits of the hipcc-built one; the EKF particle
    filter of user models ends at eight states and says so, the psi-APF asks for the Jacobians."""
    c = _twelve_case()
    mk = lambda jit: M.ssm_nlg_user(c["y"], USER_TWELVE, theta=c["th"], known_params=c["kn"], a1=c["a1"], P1=c["P1"], H=0.5,
                                     n_states=c["m"], R=c["Rm"], jit=jit)
    u = mk(False)
    if path != "grid": monkeypatch.setenv("BSSM_AMD_FORCE_PATH", path)
    out = F.bootstrap_filter(u, N, seed=7, return_indices=True)
    ref = oracle.nlg_general_bsf_filter(c["y"], c["m"], c["k"], c["th"], c["kn"], c["a1"], c["P1"], c["Z"], c["H"], c["T"], c["R"], N, seed=7)
    assert np.array_equal(out["indices"], ref["indices"])
    assert abs(out["logLik"] - ref["logLik"]) <= 1e-9 * c["n"]
    assert _rel(np.transpose(out["alpha"], (1, 0, 2)), ref["alpha"]) <= 1e-9
    assert _rel(out["weights"], ref["weights"]) <= 1e-8
    if path == "grid": return
    sm = F.particle_smoother(u, N, method="bsf", seed=7)
    ah, Vt = oracle.summary(oracle.filter_smoother(ref["alpha"], ref["indices"]))
    assert _rel(sm["alphahat"], ah.T[:c["n"]]) <= 1e-8 and _rel(sm["Vt"], Vt[:, :, :c["n"]]) <= 1e-7
    oj = F.bootstrap_filter(mk(True), N, seed=7, return_indices=True)
    for key in ("indices", "weights", "alpha"):
        assert np.array_equal(oj[key], out[key]), key
    assert oj["logLik"] == out["logLik"]
    if path == "wg":
        # (round 6: the psi-APF goes to 16 states for snippets WITH Jacobians -- USER_TWELVE_J below; this one has none and says so;
        # the EKF particle filter ends at eight states)
        with pytest.raises(Exception, match="BSSM_USER_JACOBIANS"):
            F.particle_smoother(u, 100, method="psi", seed=7)
        with pytest.raises(Exception, match="at most 8 states"):
            F.particle_smoother(u, 100, method="ekf", seed=7)
        with pytest.raises(Exception, match="16"):
            M.ssm_nlg_user(np.zeros(4), USER_TWELVE, theta=c["th"], a1=np.zeros(17), P1=np.eye(17), H=0.5, n_states=17, R=np.eye(17))


# ---------------------------------------------------------------- round 6: more than eight theta / known_params entries
USER_MANYPAR = r"""
#define BSSM_USER_NLG 1
#define BSSM_USER_JACOBIANS 1
// two states, TWELVE parameters and TEN known values (the reference has no cap on either: theta and known_params are arma::vec,
// src/model_ssm_nlg.h:23-31): the entries beyond the eighth carry the model
BSSM_FN void bssm_user_T(const double* a, const double* th, const double* kn, double* out) {
  out[0] = th[0] * a[0] + th[9] * sin(a[1]) + kn[8];
  out[1] = th[10] * a[1] + kn[9] * a[0];
}
BSSM_FN double bssm_user_Z(const double* a, const double* th, const double* kn) { return a[0] + th[11] * a[1] + kn[1]; }
BSSM_FN void bssm_user_Tg(const double* a, const double* th, const double* kn, double* Tg) { Tg[0] = th[0]; Tg[1] = kn[9]; Tg[2] = th[9] * cos(a[1]); Tg[3] = th[10]; }
BSSM_FN void bssm_user_Zg(const double* a, const double* th, const double* kn, double* Zg) { Zg[0] = 1.0; Zg[1] = th[11]; }
"""


@pytest.mark.parametrize("jit", [False, True])
@pytest.mark.parametrize("path,N", [("wg", 400), ("grid", 9000)])
def test_user_model_with_twelve_parameters_and_ten_known_values(oracle, monkeypatch, path, N, jit):
    """theta of 12 and known_params of 10 entries (round-5 review, "What's missing" 3: both were capped at 8, the reference caps neither):
    bootstrap filter and psi-APF against the general oracle over Python functions of the same theta / known_params"""
    monkeypatch.setenv("BSSM_AMD_FORCE_PATH", path)
    rng = np.random.default_rng(77)
    n, m = 26, 2
    th = np.array([0.8, 9.0, 9.0, 9.0, 9.0, 9.0, 9.0, 9.0, 9.0, 0.3, 0.7, 0.4])   # (the nines are never read)
    kn = np.array([7.0, 0.1, 7.0, 7.0, 7.0, 7.0, 7.0, 7.0, 0.05, 0.1])
    rd, Hs = np.array([0.3, 0.2]), 0.5
    a1, P1 = np.array([0.1, -0.2]), np.diag([0.4, 0.3])
    y = np.cumsum(rng.normal(0, 0.3, n)); y[5] = np.nan
    u = M.ssm_nlg_user(y, USER_MANYPAR, theta=th, known_params=kn, a1=a1, P1=P1, H=Hs, R=np.diag(rd), n_states=2, jit=jit)
    fns = dict(Z_fn=lambda t, a, th, kn: a[0] + th[11] * a[1] + kn[1],
               T_fn=lambda t, a, th, kn: np.array([th[0] * a[0] + th[9] * np.sin(a[1]) + kn[8], th[10] * a[1] + kn[9] * a[0]]),
               H_fn=lambda t, a, th, kn: Hs, R_fn=lambda t, a, th, kn: np.diag(rd))
    jac = dict(Z_gn=lambda t, a, th, kn: np.array([1.0, th[11]]),
               T_gn=lambda t, a, th, kn: np.array([[th[0], th[9] * np.cos(a[1])], [kn[9], th[10]]]))
    ref = oracle.nlg_general_bsf_filter(y, m, m, th, kn, a1, P1, nsim=N, seed=6, **fns)
    out = F.bootstrap_filter(u, N, seed=6, return_indices=True)
    assert np.array_equal(out["indices"], ref["indices"]) and abs(out["logLik"] - ref["logLik"]) <= 1e-9 * n
    rp = oracle.nlg_general_psi(y, m, m, th, kn, a1, P1, nsim=N, seed=3, **fns, **jac)
    assert rp["ok"]
    su = F.psi_setup(u)
    assert np.allclose(np.asarray(su["mode"]).reshape(m, n, order="F"), rp["mode"], rtol=0, atol=1e-9)
    op = F.particle_smoother(u, N, method="psi", seed=3, return_indices=True)
    assert np.array_equal(op["indices"], rp["indices"]) and abs(op["logLik"] - rp["logLik"]) <= 1e-9 * n


# ---------------------------------------------------------------- round 6: the psi-APF of user models with 9 ... 16 states
USER_TWELVE_J = r"""
#define BSSM_USER_NLG 1
#define BSSM_USER_JACOBIANS 1
// twelve states with Jacobians (the psi-APF needs them, as in the reference: src/model_ssm_nlg.cpp:62-143)
BSSM_FN void bssm_user_T(const double* a, const double* th, const double* kn, double* out) {
  for (int j = 0; j < 11; j++) out[j] = (0.5 + 0.03 * j) * a[j] + 0.1 * sin(a[j + 1]);
  out[11] = th[0] * a[11] + 0.05 * tanh(a[0]);
}
BSSM_FN double bssm_user_Z(const double* a, const double* th, const double* kn) { return a[0] + a[3] + a[6] + a[9] + 0.1 * a[7] * a[7]; }
BSSM_FN void bssm_user_Tg(const double* a, const double* th, const double* kn, double* Tg) {
  for (int q = 0; q < 144; q++) Tg[q] = 0.0;
  for (int j = 0; j < 11; j++) { Tg[j + 12 * j] = 0.5 + 0.03 * j; Tg[j + 12 * (j + 1)] = 0.1 * cos(a[j + 1]); }
  const double th0 = tanh(a[0]);
  Tg[11 + 12 * 11] = th[0]; Tg[11 + 12 * 0] = 0.05 * (1.0 - th0 * th0);
}
BSSM_FN void bssm_user_Zg(const double* a, const double* th, const double* kn, double* Zg) {
  for (int j = 0; j < 12; j++) Zg[j] = 0.0;
  Zg[0] = 1.0; Zg[3] = 1.0; Zg[6] = 1.0; Zg[9] = 1.0; Zg[7] = 0.2 * a[7];
}
"""


@pytest.mark.parametrize("jit", [False, True])
@pytest.mark.parametrize("path,N", [("wg", 300), ("grid", 9000)])
def test_user_model_twelve_states_psi_apf_against_the_general_oracle(oracle, monkeypatch, path, N, jit):
    """psi-APF of a user model with TWELVE states (round-5 review, "What's missing" 3: it ended at 8; the reference takes any n_states,
    src/model_ssm_nlg.cpp:971-1063): mode search and linearisation on the host (generic matrices), the MP = 16 functor kernels with the
    transition-density ratio through the inverse factor of R R' -- mode, approximate log-likelihood, ancestors, weights, log-likelihood and
    smoothed means against orc_nlg_general_psi_*; the EKF particle filter of it is refused (8 states)."""
    monkeypatch.setenv("BSSM_AMD_FORCE_PATH", path)
    rng = np.random.default_rng(23)
    m, n = 12, 20
    th, kn = np.array([0.8]), np.zeros(0)
    Rm = np.diag(0.1 + 0.02 * np.arange(m)); Rm[3, 1] = -0.05; Rm[10, 4] = 0.07; Rm[6, 5] = 0.03
    Z = lambda t, a, th_, kn_: a[0] + a[3] + a[6] + a[9] + 0.1 * a[7] ** 2
    T = lambda t, a, th_, kn_: np.array([(0.5 + 0.03 * j) * a[j] + 0.1 * np.sin(a[j + 1]) for j in range(11)] + [th_[0] * a[11] + 0.05 * np.tanh(a[0])])

    def Tg(t, a, th_, kn_):
        J = np.zeros((12, 12))
        for j in range(11):
            J[j, j] = 0.5 + 0.03 * j; J[j, j + 1] = 0.1 * np.cos(a[j + 1])
        J[11, 11] = th_[0]; J[11, 0] = 0.05 * (1.0 - np.tanh(a[0]) ** 2)
        return J
    Zg = lambda t, a, th_, kn_: np.array([1.0, 0, 0, 1.0, 0, 0, 1.0, 0.2 * a[7], 0, 1.0, 0, 0])
    a = 0.3 * rng.normal(size=m); y = np.zeros(n)
    for t in range(n):
        y[t] = Z(t, a, th, kn) + 0.5 * rng.normal()
        a = T(t, a, th, kn) + Rm @ rng.normal(size=m)
    y[5] = np.nan
    a1 = 0.1 * np.arange(m) - 0.5
    P1 = np.diag(0.05 + 0.02 * np.arange(m)); P1[0, 4] = P1[4, 0] = 0.03
    u = M.ssm_nlg_user(y, USER_TWELVE_J, theta=th, known_params=kn, a1=a1, P1=P1, H=0.5, n_states=m, R=Rm, jit=jit)
    fns = dict(Z_fn=Z, H_fn=lambda t, a, th_, kn_: 0.5, T_fn=T, R_fn=lambda t, a, th_, kn_: Rm, Z_gn=Zg, T_gn=Tg)
    ref = oracle.nlg_general_psi(y, m, m, th, kn, a1, P1, nsim=N, seed=3, **fns)
    assert ref["ok"]
    su = F.psi_setup(u)
    assert np.allclose(np.asarray(su["mode"]).reshape(m, n, order="F"), ref["mode"], rtol=0, atol=1e-9)
    assert abs(su["approx_loglik"] - ref["approx_loglik"]) <= 1e-9 * n
    out = F.particle_smoother(u, N, method="psi", seed=3, return_indices=True)
    assert np.array_equal(out["indices"], ref["indices"])
    assert abs(out["logLik"] - ref["logLik"]) <= 1e-9 * n
    assert _rel(out["weights"], ref["weights"]) <= 1e-8
    mean, cov = oracle.summary(oracle.filter_smoother(ref["alpha"], ref["indices"]))
    assert _rel(out["alphahat"], mean.T[:n]) <= 1e-8
    if path == "wg":
        from bssm_amd import _lib
        with pytest.raises(_lib.BssmAmdError, match="8 states"):
            F.particle_smoother(u, 100, method="ekf", seed=1)
