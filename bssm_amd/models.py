"""Host-side mirror of bssm's R model constructors (the named lists its C++ constructors read).

Only the fields the particle-filter path needs are produced; shapes, names and defaults follow
the reference R code so tests read like the reference's own:

* ``ssm_ung``  R/models.R:309-358      * ``bsm_ng``  R/models.R:926-1114
* ``svm``      R/models.R:1183-1238    * ``ar1_ng``  R/models.R:1285-1400
* ``ssm_ulg`` / ``bsm_lg`` / ``ar1_lg`` (gaussian observation, `bootstrap_filter` with gaussian=TRUE)
* ``init_mode`` R/init_mode.R:3-31

Arrays are numpy float64, Fortran (column-major) order, with R's extents:
Z m x (1|n), T m x m x (1|n), R m x k x (1|n), a1 m, P1 m x m, D (1|n), C m x (1|n).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

DISTRIBUTIONS = ("svm", "poisson", "binomial", "negative binomial", "gamma", "gaussian")


@dataclass
class Prior:
    """bssm's prior objects (R/priors.R): ``init`` is the parameter value the filters use; ``kind`` / ``params`` feed the
    chains of mcmc.run_mcmc (prior tables as R/priors.R:246-261 builds them)."""
    init: float
    kind: str = "fixed"
    params: tuple = ()


def normal(init, mean=0.0, sd=1.0): return Prior(float(init), "normal", (mean, sd))
def halfnormal(init, sd=1.0): return Prior(float(init), "halfnormal", (sd,))
def uniform(init, min=0.0, max=1.0): return Prior(float(init), "uniform", (min, max))
def tnormal(init, mean=0.0, sd=1.0, min=-np.inf, max=np.inf): return Prior(float(init), "tnormal", (mean, sd, min, max))
def gamma(init, shape=1.0, rate=1.0): return Prior(float(init), "gamma", (shape, rate))


_PRIOR_CODES = {"uniform": 0, "halfnormal": 1, "normal": 2, "tnormal": 3, "gamma": 4}


def prior_tables(priors):
    """(prior_distributions int32[n_par], prior_parameters 4 x n_par column-major) -- combine_priors, R/priors.R:246-261"""
    pd = np.zeros(len(priors), dtype=np.int32)
    pp = np.full((4, len(priors)), np.nan, order="F")
    for i, pr in enumerate(priors):
        if not isinstance(pr, Prior) or pr.kind not in _PRIOR_CODES:
            raise ValueError("every estimated parameter needs a prior (uniform, halfnormal, normal, tnormal or gamma)")
        pd[i] = _PRIOR_CODES[pr.kind]
        pp[:len(pr.params), i] = pr.params
    return pd, pp


def _beta_priors(beta):
    if beta is None:
        return [], []
    bl = list(beta) if isinstance(beta, (list, tuple)) else [beta]
    if all(_is_prior(b) for b in bl):
        return bl, ["coef_%d" % (i + 1) for i in range(len(bl))]
    return [], []


def _val(x):
    return x.init if isinstance(x, Prior) else float(x)


def _is_prior(x):
    return isinstance(x, Prior)


class BssmModel(dict):
    """A named list with S3-like classes, as the R constructors return."""

    def __init__(self, classes: Sequence[str], **fields):
        super().__init__(**fields)
        self.classes = tuple(classes)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def inherits(self, cls: str) -> bool:
        return cls in self.classes

    @property
    def n(self): return int(np.asarray(self["y"]).shape[0])

    @property
    def p(self): return int(np.asarray(self["y"]).shape[1]) if np.asarray(self["y"]).ndim == 2 else 1

    @property
    def m(self): return int(np.asarray(self["a1"]).shape[0])


def model_type(model: BssmModel) -> int:
    """R/model_type.R:1-16"""
    if model.inherits("lineargaussian"):
        return {"ssm_mlg": 0, "ssm_ulg": 1, "bsm_lg": 2, "ar1_lg": 3}[model.classes[0]]
    return {"ssm_mng": 0, "ssm_ung": 1, "bsm_ng": 2, "svm": 3, "ar1_ng": 4}[model.classes[0]]


def init_mode(y, u, distribution):
    """R/init_mode.R:3-31"""
    y = np.array(y, dtype=float, copy=True)
    u = np.asarray(u, dtype=float)
    nan = np.isnan(y)
    if distribution == "poisson":
        y = y / u
        y[(y < 0.1) | nan] = 0.1
        return np.log(y)
    if distribution == "binomial":
        p = (np.where(nan, 0.5, y) + 0.5) / (u + 1)
        return np.log(p / (1 - p))
    if distribution == "gamma":
        y = y / u
        y[nan | (y < 1)] = 1
        return np.log(y)
    if distribution == "negative binomial":
        y = y / u
        y[nan | (y < 1 / 6)] = 1 / 6
        return np.log(y)
    if distribution == "gaussian":
        return y
    raise ValueError("Argument distribution must be 'poisson', 'binomial', 'gamma', 'gaussian', or 'negative binomial'.")


def _f(a, shape=None):
    a = np.asarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape, order="F")
    return np.asfortranarray(a)


def _check_Z(Z, n):
    Z = np.asarray(Z, dtype=float)
    if Z.ndim == 0:
        Z = Z.reshape(1, 1)
    elif Z.ndim == 1:
        Z = Z.reshape(-1, 1)
    if Z.shape[1] not in (1, n):
        raise ValueError("'Z' must be a vector of length m or an m x n matrix")
    return _f(Z)


def _check_cube(X, m, cols, n, name):
    X = np.asarray(X, dtype=float)
    if X.ndim == 0:
        X = X.reshape(1, 1, 1)
    elif X.ndim == 1:
        X = X.reshape(m, -1, 1) if X.size == m * cols else X.reshape(1, 1, -1)
    elif X.ndim == 2:
        X = X.reshape(X.shape[0], X.shape[1], 1)
    if X.shape[0] != m or X.shape[2] not in (1, n):
        raise ValueError(f"'{name}' has wrong dimensions {X.shape}")
    return _f(X)


def _check_a1P1(a1, P1, m):
    a1 = np.zeros(m) if a1 is None else np.asarray(a1, dtype=float).reshape(-1)
    if a1.size == 1 and m > 1:
        a1 = np.repeat(a1, m)
    if a1.size != m:
        raise ValueError("'a1' must be of length m")
    if P1 is None:
        P1 = np.zeros((m, m))
    P1 = np.asarray(P1, dtype=float)
    if P1.ndim == 0:
        P1 = P1.reshape(1, 1)
    if P1.shape != (m, m):
        raise ValueError("'P1' must be m x m")
    return _f(a1), _f(P1)


def _check_D(D, n):
    if D is None:
        return _f(np.zeros(1))
    D = np.asarray(D, dtype=float).reshape(-1)
    if D.size not in (1, n):
        raise ValueError("'D' must be a scalar or of length n")
    return _f(D)


def _check_C(C, m, n):
    if C is None:
        return _f(np.zeros((m, 1)))
    C = np.asarray(C, dtype=float)
    if C.ndim <= 1:
        C = C.reshape(m, -1)
    if C.shape[0] != m or C.shape[1] not in (1, n):
        raise ValueError("'C' must be m x 1 or m x n")
    return _f(C)


def _check_u(u, y):
    n = y.shape[0]
    if u is None:
        return _f(np.ones(n))
    u = np.asarray(u, dtype=float).reshape(-1)
    if u.size == 1:
        u = np.repeat(u, n)
    if u.size != n:
        raise ValueError("'u' must be of length n")
    if np.any(u <= 0):
        raise ValueError("All values of 'u' must be positive")
    return _f(u)


def _regression(beta, xreg, n):
    """R/check_arguments.R:253-288 (create_regression), init values only"""
    if xreg is None or beta is None:
        return _f(np.zeros((n, 0))), _f(np.zeros(0))
    xreg = np.asarray(xreg, dtype=float)
    if xreg.ndim == 1:
        xreg = xreg.reshape(-1, 1)
    if xreg.shape[0] != n:
        raise ValueError("Number of rows in xreg is not equal to the length of the series y.")
    if isinstance(beta, (list, tuple)):
        b = np.array([_val(x) for x in beta], dtype=float)
    else:
        b = np.atleast_1d(np.asarray(_val(beta), dtype=float))
    if b.size != xreg.shape[1]:
        raise ValueError("Number of regression coefficients does not match the columns of xreg")
    return _f(xreg), _f(b)


def _common_tail(fields):
    fields.setdefault("max_iter", 100)
    fields.setdefault("conv_tol", 1e-8)
    fields.setdefault("local_approx", True)
    return fields


def ssm_ung(y, Z, T, R, a1=None, P1=None, distribution="poisson", phi=1.0, u=None, init_theta=(), D=None, C=None,
            update_fn=None, prior_fn=None):
    """R/models.R:309-358"""
    distribution = distribution.lower()
    if distribution not in ("poisson", "binomial", "negative binomial", "gamma"):
        raise ValueError("'arg' should be one of 'poisson', 'binomial', 'negative binomial', 'gamma'")
    y = _f(np.asarray(y, dtype=float).reshape(-1))
    n = y.shape[0]
    u = _check_u(u, y)
    Z = _check_Z(Z, n)
    m = Z.shape[0]
    T = _check_cube(T, m, m, n, "T")
    R = _check_cube(R, m, np.asarray(R).size // m if np.asarray(R).ndim <= 1 else np.asarray(R).shape[1], n, "R")
    a1, P1 = _check_a1P1(a1, P1, m)
    if not phi > 0:
        raise ValueError("Parameter 'phi' must be positive.")
    return BssmModel(("ssm_ung", "nongaussian", "bssm_model"), **_common_tail(dict(
        y=y, Z=Z, T=T, R=R, a1=a1, P1=P1, phi=float(phi), u=u, D=_check_D(D, n), C=_check_C(C, m, n),
        distribution=distribution, initial_mode=_f(init_mode(y, u, distribution)).reshape(-1, 1),
        update_fn=update_fn, prior_fn=prior_fn, theta=np.asarray(init_theta, dtype=float),
        xreg=_f(np.zeros((n, 0))), beta=_f(np.zeros(0)))))


def _bsm_structure(n, sd_level, sd_slope, sd_seasonal, sd_noise, period, a1, P1, y_frequency=1):
    slope = sd_slope is not None
    seasonal = sd_seasonal is not None
    noise = sd_noise is not None
    if sd_level is None:
        raise ValueError("Provide either prior or fixed value for sd_level.")
    if not seasonal:
        period = 1
    elif period is None:
        period = y_frequency
    npar_R = 1 + int(slope) + int(seasonal) + int(noise)
    m = 1 + int(slope) + int(seasonal) * (period - 1) + int(noise)
    a1v = np.zeros(m) if a1 is None else np.asarray(a1, dtype=float).reshape(-1)
    P1m = np.diag(np.full(m, 100.0)) if P1 is None else np.asarray(P1, dtype=float).reshape(m, m).copy()
    Z = np.zeros((m, 1)); Z[0, 0] = 1
    if seasonal:
        Z[1 + slope, 0] = 1
    T = np.zeros((m, m)); T[0, 0] = 1
    if slope:
        T[0:2, 1] = 1
    if seasonal:
        s0 = 1 + slope
        T[s0, s0:s0 + period - 1] = -1
        for j in range(period - 2):
            T[s0 + 1 + j, s0 + j] = 1
    R = np.zeros((m, max(1, npar_R)))
    R[0, 0] = _val(sd_level)
    if slope:
        R[1, 1] = _val(sd_slope)
    if seasonal:
        R[1 + slope, 1 + slope] = _val(sd_seasonal)
    if noise:
        P1m[m - 1, m - 1] = _val(sd_noise) ** 2
        Z[m - 1, 0] = 1
        R[m - 1, max(0, R.shape[1] - 1)] = _val(sd_noise)
    fixed = [int(not _is_prior(sd_level)), int(not (slope and _is_prior(sd_slope))),
             int(not (seasonal and _is_prior(sd_seasonal)))]
    return slope, seasonal, noise, period, m, _f(a1v), _f(P1m), _f(Z), _f(T).reshape(m, m, 1, order="F"), \
        _f(R).reshape(m, R.shape[1], 1, order="F"), fixed


def bsm_ng(y, sd_level=None, sd_slope=None, sd_seasonal=None, sd_noise=None, distribution="poisson", phi=None, u=None,
           beta=None, xreg=None, period=None, a1=None, P1=None, C=None):
    """R/models.R:926-1114"""
    distribution = distribution.lower()
    if distribution not in ("poisson", "binomial", "negative binomial", "gamma"):
        raise ValueError("'arg' should be one of 'poisson', 'binomial', 'negative binomial', 'gamma'")
    y = _f(np.asarray(y, dtype=float).reshape(-1))
    n = y.shape[0]
    u = _check_u(u, y)
    xr, bt = _regression(beta, xreg, n)
    slope, seasonal, noise, period, m, a1v, P1m, Z, T, R, fixed = _bsm_structure(
        n, sd_level, sd_slope, sd_seasonal, sd_noise, period, a1, P1)
    use_phi = distribution in ("negative binomial", "gamma")
    phi_est = False
    if use_phi:
        if phi is None:
            raise ValueError("phi is required for this distribution")
        phi_est = _is_prior(phi)
        phi_v = _val(phi)
        if not phi_v > 0:
            raise ValueError("Parameter 'phi' must be positive.")
    else:
        phi_v = 1.0
    named = [(nm, p) for nm, p in (("sd_level", sd_level), ("sd_slope", sd_slope), ("sd_seasonal", sd_seasonal),
                                   ("sd_noise", sd_noise)) if _is_prior(p)]
    theta = [p.init for _, p in named]
    priors, names = [p for _, p in named], [nm for nm, _ in named]
    if phi_est:
        theta.append(phi_v); priors.append(phi); names.append("phi")
    bpri, bnames = _beta_priors(beta) if xr.shape[1] else ([], [])
    theta += list(bt) if bpri else []
    priors += bpri; names += bnames
    return BssmModel(("bsm_ng", "ssm_ung", "nongaussian", "bssm_model"), **_common_tail(dict(
        y=y, Z=Z, T=T, R=R, a1=a1v, P1=P1m, phi=phi_v, u=u, xreg=xr, beta=bt, D=_f(np.zeros(1)),
        C=_check_C(C, m, n), slope=slope, seasonal=seasonal, noise=noise, period=period, fixed=fixed,
        distribution=distribution, initial_mode=_f(init_mode(y, u, distribution)).reshape(-1, 1),
        theta=np.asarray(theta, dtype=float), phi_est=phi_est, priors=priors, theta_names=names)))


def svm(y, mu=None, rho=None, sd_ar=None, sigma=None):
    """R/models.R:1183-1238"""
    if sigma is not None and mu is not None:
        raise ValueError("Define either sigma or mu, but not both.")
    y = _f(np.asarray(y, dtype=float).reshape(-1))
    rho_v, sd_v = _val(rho), _val(sd_ar)
    if not abs(rho_v) < 1:
        raise ValueError("Argument 'rho' must be strictly between -1 and 1.")
    if mu is not None:
        svm_type = 1
        initial_mode = np.log(np.maximum(1e-4, y ** 2))
        a1 = np.array([_val(mu)])
        C = np.array([[_val(mu) * (1 - rho_v)]])
        phi = 1.0
        third = _val(mu)
    else:
        svm_type = 0
        initial_mode = np.log(np.maximum(1e-4, y ** 2)) - 2 * np.log(_val(sigma))
        a1 = np.zeros(1)
        C = np.zeros((1, 1))
        phi = _val(sigma)
        third = phi
    return BssmModel(("svm", "ssm_ung", "nongaussian", "bssm_model"), **_common_tail(dict(
        y=y, Z=_f(np.ones((1, 1))), T=_f(np.full((1, 1, 1), rho_v)), R=_f(np.full((1, 1, 1), sd_v)),
        a1=_f(a1), P1=_f(np.array([[sd_v ** 2 / (1 - rho_v ** 2)]])), phi=phi, xreg=_f(np.zeros((y.shape[0], 0))),
        beta=_f(np.zeros(0)), D=_f(np.zeros(1)), C=_f(C), initial_mode=_f(initial_mode).reshape(-1, 1),
        svm_type=svm_type, distribution="svm", u=_f(np.ones(y.shape[0])), phi_est=not svm_type,
        theta=np.array([rho_v, sd_v, third]), priors=[rho, sd_ar, mu if mu is not None else sigma],
        theta_names=["rho", "sd_ar", "mu" if mu is not None else "sigma"])))


def ar1_ng(y, rho, sigma, mu, distribution, phi=None, u=None, beta=None, xreg=None):
    """R/models.R:1285-1400"""
    distribution = distribution.lower()
    if distribution not in ("poisson", "binomial", "negative binomial", "gamma"):
        raise ValueError("'arg' should be one of 'poisson', 'binomial', 'negative binomial', 'gamma'")
    y = _f(np.asarray(y, dtype=float).reshape(-1))
    n = y.shape[0]
    u = _check_u(u, y)
    xr, bt = _regression(beta, xreg, n)
    rho_v, sig_v, mu_v = _val(rho), _val(sigma), _val(mu)
    if not abs(rho_v) < 1:
        raise ValueError("Argument 'rho' must be strictly between -1 and 1.")
    use_phi = distribution in ("negative binomial", "gamma")
    phi_est = use_phi and _is_prior(phi)
    phi_v = _val(phi) if use_phi else 1.0
    bpri, bnames = _beta_priors(beta) if xr.shape[1] else ([], [])
    theta = [rho_v, sig_v] + ([mu_v] if _is_prior(mu) else []) + ([phi_v] if phi_est else []) + list(bt)
    priors = [rho, sigma] + ([mu] if _is_prior(mu) else []) + ([phi] if phi_est else []) + bpri
    names = ["rho", "sigma"] + (["mu"] if _is_prior(mu) else []) + (["phi"] if phi_est else []) + bnames
    return BssmModel(("ar1_ng", "ssm_ung", "nongaussian", "bssm_model"), **_common_tail(dict(
        y=y, Z=_f(np.ones((1, 1))), T=_f(np.full((1, 1, 1), rho_v)), R=_f(np.full((1, 1, 1), sig_v)),
        a1=_f(np.array([mu_v])), P1=_f(np.array([[sig_v ** 2 / (1 - rho_v ** 2)]])), phi=phi_v, u=u, xreg=xr, beta=bt,
        D=_f(np.zeros(1)), C=_f(np.array([[mu_v * (1 - rho_v)]])),
        initial_mode=_f(init_mode(y, u, distribution)).reshape(-1, 1), distribution=distribution,
        mu_est=_is_prior(mu), phi_est=phi_est, theta=np.asarray(theta, dtype=float), priors=priors, theta_names=names)))


def ssm_mng(y, Z, T, R, a1=None, P1=None, distribution="poisson", phi=1.0, u=None, init_theta=(), D=None, C=None,
            update_fn=None, prior_fn=None):
    """R/models.R:538-597: multivariate non-Gaussian model; y n x p, Z p x m x (1|n), D p x (1|n), u n x p."""
    y = np.asarray(y, dtype=float)
    if y.ndim != 2:
        raise ValueError("Argument 'y' must be a matrix or multivariate ts object for a multivariate model")
    n, p = y.shape
    u = np.ones((n, p)) if u is None else np.asarray(u, dtype=float).reshape(n, p)
    dists = [distribution] * p if isinstance(distribution, str) else list(distribution)
    if len(dists) != p:
        raise ValueError("'distribution' must have one entry per series")
    dists = [d.lower() for d in dists]
    for d in dists:
        if d not in ("poisson", "binomial", "negative binomial", "gamma", "gaussian"):
            raise ValueError("'arg' should be one of 'poisson', 'binomial', 'negative binomial', 'gamma', 'gaussian'")
    phi = np.repeat(float(phi), p) if np.ndim(phi) == 0 else np.asarray(phi, dtype=float).reshape(p)
    if np.any(phi <= 0):
        raise ValueError("Parameter 'phi' must be positive.")
    Z = np.asarray(Z, dtype=float)
    if Z.ndim == 2:
        Z = Z.reshape(Z.shape[0], Z.shape[1], 1)
    if Z.shape[0] != p or Z.shape[2] not in (1, n):
        raise ValueError("'Z' must be p x m or p x m x n")
    m = Z.shape[1]
    T = _check_cube(T, m, m, n, "T")
    R = _check_cube(R, m, np.asarray(R).shape[1] if np.asarray(R).ndim >= 2 else 1, n, "R")
    a1, P1 = _check_a1P1(a1, P1, m)
    if D is None:
        D = np.zeros((p, 1))
    D = np.asarray(D, dtype=float)
    D = D.reshape(p, -1) if D.ndim <= 1 else D
    if D.shape[0] != p or D.shape[1] not in (1, n):
        raise ValueError("'D' must be p x 1 or p x n")
    initial_mode = np.column_stack([init_mode(y[:, i], u[:, i], dists[i]) for i in range(p)])
    return BssmModel(("ssm_mng", "nongaussian", "bssm_model"), **_common_tail(dict(
        y=_f(y), Z=_f(Z), T=T, R=R, a1=a1, P1=P1, phi=_f(phi), u=_f(u), D=_f(D), C=_check_C(C, m, n),
        distribution=dists, initial_mode=_f(initial_mode), update_fn=update_fn, prior_fn=prior_fn,
        theta=np.asarray(init_theta, dtype=float))))


# ---- linear-Gaussian models: bootstrap_filter(gaussian=TRUE) path and the analytic pins ----
def ssm_ulg(y, Z, H, T, R, a1=None, P1=None, D=None, C=None):
    """R/models.R ssm_ulg: y_t = D_t + Z_t' alpha_t + H_t eps_t"""
    y = _f(np.asarray(y, dtype=float).reshape(-1))
    n = y.shape[0]
    Z = _check_Z(Z, n)
    m = Z.shape[0]
    T = _check_cube(T, m, m, n, "T")
    R = _check_cube(R, m, np.asarray(R).size // m if np.asarray(R).ndim <= 1 else np.asarray(R).shape[1], n, "R")
    a1, P1 = _check_a1P1(a1, P1, m)
    H = _f(np.asarray(H, dtype=float).reshape(-1))
    if H.size not in (1, n):
        raise ValueError("'H' must be a scalar or of length n")
    return BssmModel(("ssm_ulg", "lineargaussian", "bssm_model"), y=y, Z=Z, H=H, T=T, R=R, a1=a1, P1=P1,
                     D=_check_D(D, n), C=_check_C(C, m, n), distribution="gaussian", phi=1.0, u=_f(np.ones(n)),
                     xreg=_f(np.zeros((n, 0))), beta=_f(np.zeros(0)), theta=np.zeros(0))


def ssm_mlg(y, Z, H, T, R, a1=None, P1=None, D=None, C=None):
    """R/models.R ssm_mlg: y_t = D_t + Z_t alpha_t + H_t eps_t with y n x p, Z p x m x (1|n), H p x p x (1|n) (lower factor of
    the observation noise), D p x (1|n).  On the path only through the gaussian psi smoother (src/R_psi.cpp:20-25)."""
    y = np.asarray(y, dtype=float)
    if y.ndim != 2:
        raise ValueError("Argument 'y' must be a matrix or multivariate ts object for a multivariate model")
    n, p = y.shape
    Z = np.asarray(Z, dtype=float)
    if Z.ndim == 2:
        Z = Z.reshape(Z.shape[0], Z.shape[1], 1)
    if Z.shape[0] != p or Z.shape[2] not in (1, n):
        raise ValueError("'Z' must be p x m or p x m x n")
    m = Z.shape[1]
    H = np.asarray(H, dtype=float)
    if H.ndim == 2:
        H = H.reshape(p, p, 1)
    if H.shape[:2] != (p, p) or H.shape[2] not in (1, n):
        raise ValueError("'H' must be p x p or p x p x n")
    T = _check_cube(T, m, m, n, "T")
    R = _check_cube(R, m, np.asarray(R).shape[1] if np.asarray(R).ndim >= 2 else 1, n, "R")
    a1, P1 = _check_a1P1(a1, P1, m)
    D = np.zeros((p, 1)) if D is None else np.asarray(D, dtype=float)
    D = D.reshape(p, -1) if D.ndim <= 1 else D
    if D.shape[0] != p or D.shape[1] not in (1, n):
        raise ValueError("'D' must be p x 1 or p x n")
    return BssmModel(("ssm_mlg", "lineargaussian", "bssm_model"), y=_f(y), Z=_f(Z), H=_f(H), T=T, R=R, a1=a1, P1=P1,
                     D=_f(D), C=_check_C(C, m, n), distribution="gaussian", theta=np.zeros(0))


def bsm_lg(y, sd_y, sd_level=None, sd_slope=None, sd_seasonal=None, period=None, a1=None, P1=None, D=None, C=None):
    """R/models.R bsm_lg (structure shared with bsm_ng), gaussian observation sd_y"""
    y = _f(np.asarray(y, dtype=float).reshape(-1))
    n = y.shape[0]
    slope, seasonal, noise, period, m, a1v, P1m, Z, T, R, fixed = _bsm_structure(
        n, sd_level, sd_slope, sd_seasonal, None, period, a1, P1)
    if P1 is None:
        P1m = _f(np.diag(np.full(m, 100.0)))
    return BssmModel(("bsm_lg", "ssm_ulg", "lineargaussian", "bssm_model"), y=y, Z=Z, H=_f(np.array([_val(sd_y)])),
                     T=T, R=R, a1=a1v, P1=P1m, D=_check_D(D, n), C=_check_C(C, m, n), distribution="gaussian",
                     phi=1.0, u=_f(np.ones(n)), xreg=_f(np.zeros((n, 0))), beta=_f(np.zeros(0)), theta=np.zeros(0),
                     slope=slope, seasonal=seasonal, period=period, fixed=fixed)


def ar1_lg(y, rho, sigma, mu, sd_y):
    y = _f(np.asarray(y, dtype=float).reshape(-1))
    rho_v, sig_v, mu_v = _val(rho), _val(sigma), _val(mu)
    return BssmModel(("ar1_lg", "ssm_ulg", "lineargaussian", "bssm_model"), y=y, Z=_f(np.ones((1, 1))),
                     H=_f(np.array([_val(sd_y)])), T=_f(np.full((1, 1, 1), rho_v)), R=_f(np.full((1, 1, 1), sig_v)),
                     a1=_f(np.array([mu_v])), P1=_f(np.array([[sig_v ** 2 / (1 - rho_v ** 2)]])), D=_f(np.zeros(1)),
                     C=_f(np.array([[mu_v * (1 - rho_v)]])), distribution="gaussian", phi=1.0,
                     u=_f(np.ones(y.shape[0])), xreg=_f(np.zeros((y.shape[0], 0))), beta=_f(np.zeros(0)),
                     theta=np.zeros(0))


FN_IDS = {"nlg_growth": 1, "nlg_ar_exp": 2, "nlg_sin_exp": 3, "nlg_linear_gaussian": 4, "sde_gbm": 5, "sde_poisson_ou": 6}


def _dnorm(x, mu, sd):
    return -0.9189385332046727 - np.log(sd) - 0.5 * ((x - mu) / sd) ** 2


def example_log_prior(example: str):
    """log_prior_pdf of the shipped example models (R/cpp_example_models.R:59-70, 145-158, 290-299, 454-473, 595-597,
    718-721), the default ``prior_fn`` of ssm_nlg / ssm_sde models built from them."""
    def growth(th):
        return float(np.sum(_dnorm(np.exp(th[:3]), 0.0, 2.0)) + np.sum(th))

    def ar_exp(th):
        if not 0.0 < th[1] < 1.0:
            return -np.inf
        return float(_dnorm(th[0], 0, 10) + np.log(6.0) + np.log(th[1]) + np.log1p(-th[1]) +
                     _dnorm(np.exp(th[2]), 0, 1) + th[2] + _dnorm(np.exp(th[3]), 0, 1) + th[3])

    def sin_exp(th):
        return float(_dnorm(np.exp(th[0]), 0, 1) + th[0] + _dnorm(np.exp(th[1]), 0, 1) + th[1])

    def linear_gaussian(th):
        return float(_dnorm(np.exp(th[0]), 0, 1) + th[0])

    def gbm(th):
        if th[0] < 0 or th[1] < 0 or th[2] < 0.5:
            return -np.inf
        return float(_dnorm(th[0], 0, 0.1) + _dnorm(th[1], 0, 0.5) + _dnorm(th[2], 1.5, 0.5))

    def poisson_ou(th):
        x = np.exp(th[0])
        return float(np.log(x) - 2.0 * x + 2.0 * np.log(2.0) + _dnorm(th[1], 0, 4) + _dnorm(np.exp(th[2]), 0, 1) + th[0] + th[2])

    return {"nlg_growth": growth, "nlg_ar_exp": ar_exp, "nlg_sin_exp": sin_exp, "nlg_linear_gaussian": linear_gaussian,
            "sde_gbm": gbm, "sde_poisson_ou": poisson_ou}[example]


def ssm_nlg(y, example: str, theta, known_params=(), n_states=None, n_etas=None):
    """R/models.R:1472-1570 with the function pointers of `cpp_example_model(example)` (R/cpp_example_models.R):
    only the shipped example models are available on the device (nlg_growth, nlg_ar_exp, nlg_sin_exp,
    nlg_linear_gaussian); theta / known_params keep the snippets' meaning."""
    if example not in ("nlg_growth", "nlg_ar_exp", "nlg_sin_exp", "nlg_linear_gaussian"):
        raise ValueError("only the example models of cpp_example_model() are built in: nlg_growth, nlg_ar_exp, "
                         "nlg_sin_exp, nlg_linear_gaussian")
    y = _f(np.asarray(y, dtype=float).reshape(-1))
    m = 2 if example == "nlg_growth" else 1
    return BssmModel(("ssm_nlg", "bssm_model"), y=y, model_id=FN_IDS[example], example=example,
                     theta=np.asarray(theta, dtype=float), known_params=np.asarray(known_params, dtype=float),
                     n_states=m, n_etas=m, a1=np.zeros(m), x0=0.0, positive=False, L=0)


def ssm_sde(y, example: str, theta, x0, positive: bool):
    """R/models.R:1600-1647 with the pointers of cpp_example_model('sde_gbm' | 'sde_poisson_ou')."""
    if example not in ("sde_gbm", "sde_poisson_ou"):
        raise ValueError("only the example models of cpp_example_model() are built in: sde_gbm, sde_poisson_ou")
    y = _f(np.asarray(y, dtype=float).reshape(-1))
    return BssmModel(("ssm_sde", "bssm_model"), y=y, model_id=FN_IDS[example], example=example,
                     theta=np.asarray(theta, dtype=float), known_params=np.zeros(0), x0=float(x0),
                     positive=bool(positive), a1=np.zeros(1), L=0)


def ssm_nlg_user(y, device_code: str, theta, a1, P1, H=1.0, R_diag=None, known_params=(), n_states=None, R=None,
                 known_tv_params=None, jit=False):
    """ssm_nlg (R/models.R:1472-1570) with USER functions.  The reference takes pointers to user C++ compiled at run time
    (Z, H, T, R, Z_gn, T_gn, a1, P1); here ``device_code`` is a snippet of ``__device__`` functions compiled by hipcc into
    the filter kernels (contract: bssm_amd/csrc/pf_functors.h -- ``#define BSSM_USER_NLG 1``, ``bssm_user_T``,
    ``bssm_user_Z``), and the state-independent pieces are numbers: H (observation sd), the diagonal of R, a1, P1.
    With the Jacobians in the snippet (``BSSM_USER_JACOBIANS``, ``bssm_user_Tg``, ``bssm_user_Zg``) the psi-APF and the
    EKF particle filter work as for the built-in models (up to 8 states; any constant matrix ``R`` with R R'
    positive definite on its non-zero diagonal, any positive semi-definite P1; the EKF particle filter likewise, with a constant diagonal R
    for 1- and 2-state models).
    ``known_tv_params``: the reference's matrix of known time-varying values (R/models.R:1507); a snippet that defines
    ``BSSM_USER_TV`` receives it, with the time index, as the leading ``const bssm_user_tv& tv`` argument of every function
    (``tv.t``, ``tv.known_tv`` column-major ``tv.rows x tv.cols``).
    The bootstrap filter / smoother / bootstrap IS correction and (round 6, snippets with Jacobians and a constant R) the psi-APF take up to 16 states; the EKF particle filter: 8, with any ``R`` (n_states x n_etas matrix, or a function of theta) and the
    reference's state-dependent ``H_fn`` / ``R_fn`` as snippet functions (``BSSM_USER_HFN`` + ``bssm_user_H``,
    ``BSSM_USER_RFN`` + ``bssm_user_R``; src/model_ssm_nlg.h:12-13), which then replace ``H`` / ``R`` (``R`` still fixes
    n_etas; for the psi-APF R(a) R(a)' must be positive definite on its non-zero diagonal).
    ``jit=True``: the snippet is compiled IN PROCESS by hiprtc (no hipcc, ~1-2 s per kernel variant on first use) instead of being
    built into a shared library by hipcc (10-18 s); such a model runs the bootstrap filter / smoother / bootstrap IS correction only."""
    y = _f(np.asarray(y, dtype=float).reshape(-1))
    theta = np.asarray(theta, dtype=float)
    m = int(n_states or np.asarray(a1(theta) if callable(a1) else a1, dtype=float).size)
    if not 1 <= m <= 16:
        raise ValueError("user-defined ssm_nlg models: n_states must be 1 ... 16 (EKF particle filter: 8)")
    if R is None and R_diag is None:
        if "BSSM_USER_RFN" not in device_code:
            raise ValueError("give R_diag, R, or a snippet with BSSM_USER_RFN / bssm_user_R")
        R_diag = np.zeros(m)
    k = m if R is None else int(np.asarray(R(theta) if callable(R) else R, dtype=float).reshape(m, -1).shape[1])
    # H, R_diag, a1, P1 may be functions of theta (as the reference's H_fn / R_fn / a1_fn / P1_fn are): post_correct and
    # run_mcmc then evaluate them for every theta
    num = lambda v, shape: v if callable(v) else np.asarray(v, dtype=float).reshape(shape)
    return BssmModel(("ssm_nlg", "bssm_model"), y=y, model_id=7, example="user", user_source=str(device_code),
                     theta=theta, known_params=np.asarray(known_params, dtype=float),
                     n_states=m, n_etas=k, a1=num(a1, -1), P1=num(P1, (m, m)), H=H if callable(H) else float(H),
                     R_diag=None if R_diag is None else num(R_diag, -1), R=None if R is None else num(R, (m, k)),
                     known_tv_params=None if known_tv_params is None else np.asfortranarray(np.atleast_2d(np.asarray(known_tv_params, dtype=float))),
                     x0=0.0, positive=False, L=0, user_jit=bool(jit))


def ssm_sde_user(y, device_code: str, theta, x0, positive: bool, jit=False):
    """ssm_sde (R/models.R:1600-1647) with USER drift / diffusion / ddiffusion / observation log-density
    (``#define BSSM_USER_SDE 1``, ``bssm_user_drift``, ``bssm_user_diffusion``, ``bssm_user_ddiffusion``,
    ``bssm_user_log_obs``), compiled at run time."""
    y = _f(np.asarray(y, dtype=float).reshape(-1))
    return BssmModel(("ssm_sde", "bssm_model"), y=y, model_id=8, example="user", user_source=str(device_code),
                     theta=np.asarray(theta, dtype=float), known_params=np.zeros(0), x0=float(x0), positive=bool(positive),
                     n_states=1, a1=np.zeros(1), L=0, user_jit=bool(jit))


def model_arrays(model: BssmModel) -> dict:
    """Flatten a model into the plain column-major arrays + flags the C ABI struct carries
    (the field list of src/model_ssm_ung.cpp:9-32)."""
    if model.inherits("ssm_mng"):
        return _mng_arrays(model)
    if model.inherits("ssm_mlg"):
        return _mlg_arrays(model)
    y = np.ascontiguousarray(np.asarray(model["y"], dtype=np.float64).reshape(-1))
    n = y.shape[0]
    Z = np.asfortranarray(model["Z"], dtype=np.float64)
    T = np.asfortranarray(model["T"], dtype=np.float64)
    R = np.asfortranarray(model["R"], dtype=np.float64)
    m, k = Z.shape[0], R.shape[1]
    D = np.ascontiguousarray(np.asarray(model["D"], dtype=np.float64).reshape(-1))
    C = np.asfortranarray(model["C"], dtype=np.float64)
    xreg = np.asfortranarray(model.get("xreg", np.zeros((n, 0))), dtype=np.float64)
    beta = np.ascontiguousarray(np.asarray(model.get("beta", np.zeros(0)), dtype=np.float64).reshape(-1))
    dist = DISTRIBUTIONS.index(model["distribution"])
    out = dict(
        n=n, m=m, k=k, Ztv=int(Z.shape[1] > 1), Ttv=int(T.shape[2] > 1), Rtv=int(R.shape[2] > 1),
        Dtv=int(D.size > 1), Ctv=int(C.shape[1] > 1), Htv=0, distribution=dist, n_xreg=int(xreg.shape[1]),
        max_iter=int(model.get("max_iter", 100)), local_approx=int(bool(model.get("local_approx", True))),
        conv_tol=float(model.get("conv_tol", 1e-8)), phi=float(model.get("phi", 1.0)),
        y=y, Z=Z, T=T, R=R, a1=np.ascontiguousarray(model["a1"], dtype=np.float64),
        P1=np.asfortranarray(model["P1"], dtype=np.float64), D=D, C=C, xreg=xreg, beta=beta,
        u=np.ascontiguousarray(np.asarray(model.get("u", np.ones(n)), dtype=np.float64).reshape(-1)),
        H=None, initial_mode=None)
    if out["u"].size == 1:
        out["u"] = np.repeat(out["u"], n)
    if dist == 5:
        H = np.ascontiguousarray(np.asarray(model["H"], dtype=np.float64).reshape(-1))
        out["H"] = H
        out["Htv"] = int(H.size > 1)
    if model.get("initial_mode") is not None:
        out["initial_mode"] = np.ascontiguousarray(np.asarray(model["initial_mode"], dtype=np.float64).reshape(-1))
    out["xbeta"] = (xreg @ beta) if xreg.shape[1] else np.zeros(n)
    return out


def _mlg_arrays(model: BssmModel) -> dict:
    """ssm_mlg fields (src/model_ssm_mlg.cpp:7-31): y n x p as R holds it, H p x p x (1|n)"""
    y = np.asfortranarray(model["y"], dtype=np.float64)
    n, p = y.shape
    Z, H = np.asfortranarray(model["Z"], dtype=np.float64), np.asfortranarray(model["H"], dtype=np.float64)
    T, R = np.asfortranarray(model["T"], dtype=np.float64), np.asfortranarray(model["R"], dtype=np.float64)
    D, C = np.asfortranarray(model["D"], dtype=np.float64), np.asfortranarray(model["C"], dtype=np.float64)
    return dict(n=n, m=Z.shape[1], k=R.shape[1], p=p, Ztv=int(Z.shape[2] > 1), Ttv=int(T.shape[2] > 1), Rtv=int(R.shape[2] > 1),
                Dtv=int(D.shape[1] > 1), Ctv=int(C.shape[1] > 1), Htv=int(H.shape[2] > 1), distribution=5, n_xreg=0, max_iter=100,
                local_approx=1, conv_tol=1e-8, phi=1.0, y=y, Z=Z, T=T, R=R, a1=np.ascontiguousarray(model["a1"], dtype=np.float64),
                P1=np.asfortranarray(model["P1"], dtype=np.float64), D=D, C=C, xreg=np.zeros((n, 0)), beta=np.zeros(0), u=None, H=H,
                initial_mode=None, xbeta=np.zeros(n), distribution_p=np.full(p, 5, dtype=np.int32), phi_p=np.ones(p))


def _mng_arrays(model: BssmModel) -> dict:
    """Multivariate counterpart of model_arrays (fields of src/model_ssm_mng.cpp:7-31); y, u, initial_mode stay
    n x p as R holds them -- the library transposes on entry like the reference."""
    y = np.asfortranarray(model["y"], dtype=np.float64)
    n, p = y.shape
    Z = np.asfortranarray(model["Z"], dtype=np.float64)
    T = np.asfortranarray(model["T"], dtype=np.float64)
    R = np.asfortranarray(model["R"], dtype=np.float64)
    D = np.asfortranarray(model["D"], dtype=np.float64)
    C = np.asfortranarray(model["C"], dtype=np.float64)
    m, k = Z.shape[1], R.shape[1]
    dist = np.array([DISTRIBUTIONS.index(d) for d in model["distribution"]], dtype=np.int32)
    return dict(
        n=n, m=m, k=k, p=p, Ztv=int(Z.shape[2] > 1), Ttv=int(T.shape[2] > 1), Rtv=int(R.shape[2] > 1),
        Dtv=int(D.shape[1] > 1), Ctv=int(C.shape[1] > 1), Htv=0, distribution=int(dist[0]), n_xreg=0,
        max_iter=int(model.get("max_iter", 100)), local_approx=int(bool(model.get("local_approx", True))),
        conv_tol=float(model.get("conv_tol", 1e-8)), phi=float(np.asarray(model["phi"]).reshape(-1)[0]),
        y=y, Z=Z, T=T, R=R, a1=np.ascontiguousarray(model["a1"], dtype=np.float64),
        P1=np.asfortranarray(model["P1"], dtype=np.float64), D=D, C=C, xreg=np.zeros((n, 0)), beta=np.zeros(0),
        u=np.asfortranarray(model["u"], dtype=np.float64), H=None,
        initial_mode=np.asfortranarray(model["initial_mode"], dtype=np.float64), xbeta=np.zeros(n),
        distribution_p=dist, phi_p=np.ascontiguousarray(model["phi"], dtype=np.float64))
