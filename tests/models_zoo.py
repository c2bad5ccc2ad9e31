"""Seeded synthetic models shared by the CPU and GPU tests (numpy Generator(PCG64(seed)))."""
import numpy as np

from bssm_amd import models as M


def poisson_local_level(n=200, seed=101, sd_level=0.1):
    """BASELINE configs[0] / SURVEY 8d C1: bsm_ng Poisson local level."""
    rng = np.random.default_rng(seed)
    level = np.log(5.0) + np.cumsum(rng.normal(0, sd_level, n))
    y = rng.poisson(np.exp(level)).astype(float)
    return M.bsm_ng(y, sd_level=sd_level, distribution="poisson")


def negbin_ar1(n=1000, seed=102, rho=0.95, sigma=0.3, mu=1.0, phi=5.0):
    """ssm_ung negbin AR(1), m = k = 1."""
    rng = np.random.default_rng(seed)
    a = np.zeros(n)
    a[0] = mu + rng.normal(0, sigma / np.sqrt(1 - rho ** 2))
    for t in range(1, n):
        a[t] = mu * (1 - rho) + rho * a[t - 1] + rng.normal(0, sigma)
    lam = np.exp(a)
    y = rng.negative_binomial(phi, phi / (phi + lam)).astype(float)
    return M.ssm_ung(y, Z=1.0, T=rho, R=sigma, a1=mu, P1=sigma ** 2 / (1 - rho ** 2), C=np.array([[mu * (1 - rho)]]),
                     distribution="negative binomial", phi=phi)


def negbin_m4(n=1000, seed=103, phi=5.0):
    """SURVEY 8d C2-m4 (the metric config): m = k = 4."""
    rng = np.random.default_rng(seed)
    T = np.array([[1, 1, 0, 0], [0, 1, 0, 0], [0, 0, 0.9, 0], [0, 0, 0, 0.5]], dtype=float)
    R = np.diag([0.05, 0.005, 0.2, 0.3])
    Z = np.array([1.0, 0, 1, 1])
    a1 = np.array([1.0, 0, 0, 0])
    P1 = np.diag([1, 0.01, 0.21, 0.12])
    a = a1 + np.sqrt(np.diag(P1)) * rng.normal(size=4)
    y = np.zeros(n)
    for t in range(n):
        lam = np.exp(np.clip(Z @ a, -20, 8))
        y[t] = rng.negative_binomial(phi, phi / (phi + lam))
        a = T @ a + R @ rng.normal(size=4)
    return M.ssm_ung(y, Z=Z, T=T, R=R, a1=a1, P1=P1, distribution="negative binomial", phi=phi)


def negbin_m4_informative(n=1000, seed=103, phi=5.0):
    """The metric model (same T, R, Z, a1, P1, phi as negbin_m4) on a series whose counts stay informative.

    negbin_m4 simulates y from the model itself, and the model's level is an integrated random walk: on seed 103 the signal
    leaves the range of counts after ~100 steps (-15 ... -38, clipped at -20), 98 % of the y are 0, every weight is
    1 - O(1e-9) and stratified resampling is the identity map.  Here y follows a bounded signal the same model can track
    (a slow cycle between log-means 0.5 and 2.5 plus the model's two AR components), so the weights spread over orders of
    magnitude and every step resamples for real (ancestors with 0, 1, 2, ... children)."""
    rng = np.random.default_rng(seed)
    base = negbin_m4(8, seed=seed, phi=phi)     # the model matrices
    a3, a4 = 0.0, 0.0
    y = np.zeros(n)
    for t in range(n):
        s = 1.5 + np.sin(2 * np.pi * t / 250.0) + a3 + a4
        y[t] = rng.negative_binomial(phi, phi / (phi + np.exp(s)))
        a3 = 0.9 * a3 + 0.2 * rng.normal()
        a4 = 0.5 * a4 + 0.3 * rng.normal()
    arr = M.model_arrays(base)
    return M.ssm_ung(y, Z=np.array([1.0, 0, 1, 1]), T=np.asarray(arr["T"]).reshape(4, 4, order="F"), R=np.asarray(arr["R"]).reshape(4, 4, order="F"),
                     a1=np.array([1.0, 0, 0, 0]), P1=np.diag([1, 0.01, 0.21, 0.12]), distribution="negative binomial", phi=phi)


def small_zoo():
    """Small models covering every distribution, missing values, time-varying matrices, xreg."""
    rng = np.random.default_rng(7)
    zoo = {}
    zoo["poisson_ll"] = poisson_local_level(40, seed=5)
    y = rng.poisson(4, 30).astype(float); y[[3, 4, 17]] = np.nan
    zoo["poisson_slope_missing"] = M.bsm_ng(y, sd_level=0.3, sd_slope=0.05, distribution="poisson",
                                            P1=np.diag([1.0, 0.1]), a1=[1.0, 0.0])
    yb = np.array([1, 0, 1, 1, 1, 0, 0, 0] * 3, dtype=float)
    zoo["binomial_ar1_xreg"] = M.ar1_ng(yb, rho=M.uniform(0.5, -1, 1), sigma=M.halfnormal(0.4, 2), mu=0.3,
                                        distribution="binomial", u=np.full(24, 2.0), xreg=np.arange(1, 25) / 10.0,
                                        beta=M.normal(0.1, 0, 1))
    zoo["negbin_ar1"] = negbin_ar1(35, seed=9)
    yg = rng.gamma(2.0, 1.5, 25)
    zoo["gamma_ll"] = M.bsm_ng(yg, sd_level=0.2, distribution="gamma", phi=2.0, u=np.full(25, 1.5), P1=np.array([[1.0]]))
    ysv = rng.normal(0, 1.0, 45) * np.exp(0.5 * np.cumsum(rng.normal(0, 0.2, 45)))
    zoo["svm"] = M.svm(ysv, rho=M.uniform(0.95, -1, 1), sd_ar=M.halfnormal(0.2, 5), sigma=M.halfnormal(1.0, 2))
    n = 20  # fully time-varying m = 3, k = 2
    Tt = np.zeros((3, 3, n)); Rt = np.zeros((3, 2, n)); Zt = np.zeros((3, n)); Ct = np.zeros((3, n))
    for t in range(n):
        Tt[:, :, t] = 0.8 * np.eye(3) + 0.05 * rng.normal(size=(3, 3))
        Rt[:, :, t] = 0.2 * rng.normal(size=(3, 2))
        Zt[:, t] = rng.normal(size=3) * 0.5
        Ct[:, t] = 0.05 * rng.normal(size=3)
    zoo["poisson_tv_m3"] = M.ssm_ung(rng.poisson(3, n).astype(float), Z=Zt, T=Tt, R=Rt, a1=[0.5, 0, 0],
                                     P1=np.diag([0.3, 0.2, 0.0]), D=0.1 * rng.normal(size=n), C=Ct,
                                     distribution="poisson", u=rng.uniform(0.5, 2, n))
    zoo["negbin_m4"] = negbin_m4(25, seed=11)
    zoo["gaussian_bsm"] = M.bsm_lg(np.ones(5), sd_level=0.05, sd_slope=0.01, sd_y=0.01, a1=[1, 0],
                                   P1=np.diag([0.01, 0.01]))
    yl = np.cumsum(rng.normal(0, 0.3, 30)) + rng.normal(0, 0.5, 30); yl[10] = np.nan
    zoo["gaussian_ar1"] = M.ar1_lg(yl, rho=0.9, sigma=0.3, mu=0.2, sd_y=0.5)
    return zoo


def mng_zoo():
    """ssm_mng models: mixed distributions, missing values (single series and whole rows), time-varying Z/D."""
    rng = np.random.default_rng(8)
    zoo = {}
    n, p, m = 30, 3, 4
    Z = rng.normal(0, 0.5, (p, m)); T = 0.9 * np.linalg.qr(rng.normal(size=(m, m)))[0]; R = 0.2 * np.eye(m)[:, :3]
    y = np.column_stack([rng.poisson(3, n), rng.negative_binomial(5, 0.5, n), rng.normal(0, 1, n)]).astype(float)
    y[4, 0] = np.nan; y[7, :2] = np.nan
    zoo["mixed_m4_p3"] = M.ssm_mng(y, Z, T, R, a1=np.zeros(m), P1=0.3 * np.eye(m),
                                   distribution=["poisson", "negative binomial", "gaussian"], phi=[1, 5, 0.7],
                                   u=np.abs(rng.normal(1, 0.1, (n, p))) + 0.5)
    n, p, m = 24, 2, 2
    Zt = rng.normal(0, 0.6, (p, m, n)); Dt = 0.2 * rng.normal(size=(p, n))
    y = np.column_stack([rng.binomial(5, 0.4, n), rng.gamma(2.0, 1.0, n)]).astype(float); y[10, :] = np.nan
    zoo["tv_binom_gamma"] = M.ssm_mng(y, Zt, np.array([[0.9, 0.1], [0.0, 0.7]]), np.diag([0.3, 0.2]), a1=[0.2, 0.1],
                                      P1=np.diag([0.4, 0.3]), distribution=["binomial", "gamma"], phi=[1, 2.5],
                                      u=np.column_stack([np.full(n, 5.0), np.full(n, 1.3)]), D=Dt)
    n, p, m = 20, 4, 8   # the BASELINE config shape (m = 8, p = 4) at small T
    Q = np.linalg.qr(rng.normal(size=(m, m)))[0]
    Z = rng.normal(0, 0.5, (p, m))
    y = np.column_stack([rng.poisson(2, n), rng.poisson(4, n), rng.negative_binomial(5, 0.6, n), rng.normal(0, 0.8, n)]).astype(float)
    zoo["c4_m8_p4"] = M.ssm_mng(y, Z, 0.95 * Q, 0.1 * np.eye(m), a1=np.zeros(m), P1=0.05 * np.eye(m),
                                distribution=["poisson", "poisson", "negative binomial", "gaussian"], phi=[1, 1, 5, 0.5])
    return zoo


def mng_c4(n=2000, seed=104):
    """ssm_mng m = 8, k = 8, p = 4; T = 0.95 Q (Q random orthogonal), R = 0.1 I, Z ~ N(0, 0.25),
    distributions (poisson, poisson, negbin phi = 5, gaussian sd 0.5)."""
    rng = np.random.default_rng(seed)
    m, p = 8, 4
    Q = np.linalg.qr(rng.normal(size=(m, m)))[0]
    T, R, Z = 0.95 * Q, 0.1 * np.eye(m), rng.normal(0, 0.5, (p, m))
    a = np.zeros(m)
    y = np.zeros((n, p))
    for t in range(n):
        s = Z @ a
        y[t] = [rng.poisson(np.exp(s[0])), rng.poisson(np.exp(s[1])), rng.negative_binomial(5, 5 / (5 + np.exp(s[2]))),
                s[3] + 0.5 * rng.normal()]
        a = T @ a + R @ rng.normal(size=m)
    return M.ssm_mng(y, Z, T, R, a1=np.zeros(m), P1=0.1 * np.eye(m),
                     distribution=["poisson", "poisson", "negative binomial", "gaussian"], phi=[1, 1, 5, 0.5])
