"""Python mirror of bssm's R front-ends for the particle-filter path.  Same names, argument
meaning, output fields and error behaviour as

* ``bootstrap_filter``  R/bootstrap_filter.R:24-202
* ``particle_smoother`` R/particle_smoother.R:1-291
* ``logLik``            R/loglik.R:1-161

Each call goes through the C ABI of libbssm_amd.so (HIP kernels); nothing here computes.
"""
from __future__ import annotations

import ctypes as C
import warnings

import numpy as np

from . import _lib
from .models import BssmModel, model_arrays, model_type

_INT_MAX = 2 ** 31 - 1


def check_intmax(x, name="particles", positive=True, max=1e5):
    """R/check_arguments.R:407-418"""
    if not (isinstance(x, (int, np.integer)) or (isinstance(x, float) and x == int(x))) or x < 0 or (positive and x <= 0):
        raise ValueError(f"Argument '{name}' should be a {'positive' if positive else 'non-negative'} integer. ")
    if x > max:
        shown = f"{max:.0e}" if max >= 1e5 and float(f"{max:.0e}") == max else f"{max:g}"  # R prints 1e5 as 1e+05
        raise ValueError(f"You probably do not want '{name}' > {shown}. If you really do, please file an issue at Github. ")
    return int(x)


def _default_seed():
    return int(np.random.default_rng().integers(0, _INT_MAX))


def _alloc(shape):
    return np.zeros(shape, dtype=np.float64, order="F")


def _prep(model: BssmModel, need_mode: bool):
    arr = model_arrays(model)
    if not need_mode:
        arr = dict(arr)
    smod, keep = _lib.pack_model(arr)
    return arr, smod, keep


def _warn_if_degenerate(rc):
    if rc == _lib.BSSM_WARN_DEGENERATE:
        warnings.warn("Particle filtering stopped prematurely due to nonfinite log-likelihood.")


def _functor_filter(model: BssmModel, particles: int, seed: int, smoother: bool, L, return_indices: bool, psi: bool = False,
                    ekpf: bool = False):
    """ssm_nlg / ssm_sde bootstrap filter (R/bootstrap_filter.R:126-202) and bsf smoother (R/particle_smoother.R:193-291)."""
    sde = model.inherits("ssm_sde")
    if sde:
        if L is None:
            raise ValueError("argument 'L' is missing")
        L = check_intmax(L, "L", max=20)
    smod, keep = _lib.pack_fn_model(model, L)
    n = model.n
    m = int(model.get("n_states", 1))
    out = _lib.FilterOut()
    at, att = _alloc((n + 1, m)), _alloc((n, m))
    Pt, Ptt = _alloc((m, m, n + 1)), _alloc((m, m, n))
    weights, alpha = _alloc((particles, n + 1)), _alloc((m, n + 1, particles))
    indices = np.zeros((particles, n), dtype=np.uint32, order="F") if return_indices else None
    if smoother:
        out.alphahat, out.Vt = _lib.dptr(at), _lib.dptr(Pt)
    else:
        out.at, out.att, out.Pt, out.Ptt = map(_lib.dptr, (at, att, Pt, Ptt))
    out.weights, out.alpha = _lib.dptr(weights), _lib.dptr(alpha)
    if indices is not None:
        out.indices = indices.ctypes.data_as(_lib.c_u32_p)
    L_ = _lib.lib()
    if ekpf:
        fn = L_.bssm_ekpf_smoother_nlg if smoother else L_.bssm_ekpf_nlg
        rc = fn(C.byref(smod), C.c_uint32(particles), C.c_uint32(seed), C.byref(out))
    elif psi:
        rc = L_.bssm_psi_smoother_nlg(C.byref(smod), C.c_uint32(particles), C.c_uint32(seed),
                                      C.c_int32(int(model.get("max_iter", 100))), C.c_double(float(model.get("conv_tol", 1e-8))),
                                      C.byref(out))
    else:
        fn = {(False, False): L_.bssm_bsf_nlg, (False, True): L_.bssm_bsf_smoother_nlg,
              (True, False): L_.bssm_bsf_sde, (True, True): L_.bssm_bsf_smoother_sde}[(sde, smoother)]
        rc = fn(C.byref(smod), C.c_uint32(particles), C.c_uint32(seed), C.byref(out))
    _warn_if_degenerate(_lib.check(rc))
    if smoother:
        res = dict(alphahat=at[:n, :], Vt=Pt[:, :, :n], weights=weights, logLik=out.logLik,
                   alpha=np.transpose(alpha[:, :n, :], (1, 0, 2)))
    else:
        res = dict(at=at, att=att, Pt=Pt, Ptt=Ptt, weights=weights, logLik=out.logLik, alpha=np.transpose(alpha, (1, 0, 2)))
    if indices is not None:
        res["indices"] = indices
    return res


def milstein(model: BssmModel, L: int, t: float = 1.0, seed: int = 1) -> float:
    """R_milstein (R/RcppExports.R:104-106, src/R_milstein.cpp:4-22): terminal value of one Milstein path of an ssm_sde model from
    x0 over [0, t] in 2^L steps, on the device (bssm_milstein)."""
    if not model.inherits("ssm_sde"):
        raise TypeError("milstein() is for ssm_sde models")
    L = check_intmax(L, "L", positive=False, max=20)   # (2^0 = one step is a valid discretisation, src/milstein.cpp:22)
    smod, keep = _lib.pack_fn_model(model, L)
    x = C.c_double(0.0)
    fn = _lib.lib().bssm_milstein
    fn.argtypes = [C.POINTER(_lib.FnModel), C.c_double, C.c_uint32, C.POINTER(C.c_double)]
    _lib.check(fn(C.byref(smod), C.c_double(float(t)), C.c_uint32(seed), C.byref(x)))
    return x.value


def bootstrap_filter(model: BssmModel, particles: int, seed: int | None = None, return_indices: bool = False, L=None):
    """R/bootstrap_filter.R:48-120 (lineargaussian :48-79, nongaussian :85-120).

    Returns dict(at (n+1) x m, att n x m, Pt, Ptt, weights N x (n+1), logLik, alpha (n+1) x m x N)."""
    particles = check_intmax(particles, "particles")
    seed = check_intmax(_default_seed() if seed is None else seed, "seed", False, max=_INT_MAX)
    if model.inherits("ssm_nlg") or model.inherits("ssm_sde"):
        return _functor_filter(model, particles, seed, False, L, return_indices)
    gaussian = model.inherits("lineargaussian")
    if not (gaussian or model.inherits("nongaussian")):
        raise TypeError("bootstrap_filter is implemented for lineargaussian, nongaussian, ssm_nlg and ssm_sde models")
    n, m = model.n, model.m
    if n * m * particles > 1e10:  # R/bootstrap_filter.R:100-105
        warnings.warn("Trying to sample states 1e10 times; consider smaller number of particles.")
    arr, smod, keep = _prep(model, False)
    out = _lib.FilterOut()
    at, att = _alloc((n + 1, m)), _alloc((n, m))
    Pt, Ptt = _alloc((m, m, n + 1)), _alloc((m, m, n))
    weights, alpha = _alloc((particles, n + 1)), _alloc((m, n + 1, particles))
    indices = np.zeros((particles, n), dtype=np.uint32, order="F") if return_indices else None
    out.at, out.att, out.Pt, out.Ptt = map(_lib.dptr, (at, att, Pt, Ptt))
    out.weights, out.alpha = _lib.dptr(weights), _lib.dptr(alpha)
    if indices is not None:
        out.indices = indices.ctypes.data_as(_lib.c_u32_p)
    rc = _lib.check(_lib.lib().bssm_bsf(C.byref(smod), C.c_uint32(particles), C.c_uint32(seed), int(gaussian),
                                         model_type(model), C.byref(out)))
    _warn_if_degenerate(rc)
    res = dict(at=at, att=att, Pt=Pt, Ptt=Ptt, weights=weights, logLik=out.logLik,
               alpha=np.transpose(alpha, (1, 0, 2)))  # aperm(out$alpha, c(2, 1, 3)), R/bootstrap_filter.R:118
    if indices is not None:
        res["indices"] = indices
    return res


def particle_smoother(model: BssmModel, particles: int, method: str = "psi", seed: int | None = None,
                      return_indices: bool = False, L=None):
    """R/particle_smoother.R:140-186 (nongaussian) and :84-133 (lineargaussian, bsf only here).

    Returns dict(alphahat n x m, Vt m x m x n, weights, logLik, alpha n x m x N): the extra (n+1)-th
    time point is dropped exactly as R/particle_smoother.R:178-184 does."""
    particles = check_intmax(particles, "particles")
    seed = check_intmax(_default_seed() if seed is None else seed, "seed", False, max=_INT_MAX)
    if method not in ("psi", "bsf", "ekf"):
        raise ValueError("'arg' should be one of 'psi', 'bsf', 'ekf'")
    if model.inherits("ssm_nlg") or model.inherits("ssm_sde"):
        if method == "ekf":  # R/particle_smoother.R:222-238 -> ekpf_smoother (src/R_ekpf.cpp:56-100)
            if not model.inherits("ssm_nlg"):
                raise ValueError("Method 'ekf' is only applicable to ssm_nlg models.")
            return _functor_filter(model, particles, seed, True, L, return_indices, ekpf=True)
        psi = method == "psi" and model.inherits("ssm_nlg")
        return _functor_filter(model, particles, seed, True, L, return_indices, psi=psi)
    if method == "ekf":
        raise ValueError("Method 'ekf' is only applicable to ssm_nlg models.")
    gaussian = model.inherits("lineargaussian")
    n, m = model.n, model.m
    if gaussian and method == "psi":  # R/particle_smoother.R:107-122: exact simulation smoother, moments taken over the draws
        arr, smod, keep = _prep(model, False)
        alpha = _alloc((m, n + 1, particles))
        _lib.check(_lib.lib().bssm_gaussian_psi_smoother(C.byref(smod), C.c_uint32(particles), C.c_uint32(seed),
                                                         model_type(model), _lib.dptr(alpha)))
        alphahat = alpha.mean(axis=2).T
        Vt = np.stack([np.atleast_2d(np.cov(alpha[:, t, :], ddof=1)) for t in range(n + 1)], axis=2)
        return dict(alphahat=alphahat[:n, :], Vt=Vt[:, :, :n], alpha=np.transpose(alpha, (1, 0, 2)))
    arr, smod, keep = _prep(model, method == "psi")
    out = _lib.FilterOut()
    alphahat, Vt = _alloc((n + 1, m)), _alloc((m, m, n + 1))
    weights, alpha = _alloc((particles, n + 1)), _alloc((m, n + 1, particles))
    indices = np.zeros((particles, n), dtype=np.uint32, order="F") if return_indices else None
    out.alphahat, out.Vt, out.weights, out.alpha = map(_lib.dptr, (alphahat, Vt, weights, alpha))
    if indices is not None:
        out.indices = indices.ctypes.data_as(_lib.c_u32_p)
    L = _lib.lib()
    if method == "psi":
        rc = L.bssm_psi_smoother(C.byref(smod), C.c_uint32(particles), C.c_uint32(seed), model_type(model), C.byref(out))
    else:
        rc = L.bssm_bsf_smoother(C.byref(smod), C.c_uint32(particles), C.c_uint32(seed), int(gaussian),
                                 model_type(model), C.byref(out))
    _warn_if_degenerate(_lib.check(rc))
    res = dict(alphahat=alphahat[:n, :], Vt=Vt[:, :, :n], weights=weights, logLik=out.logLik,
               alpha=np.transpose(alpha[:, :n, :], (1, 0, 2)))
    if indices is not None:
        res["indices"] = indices
    return res


def logLik(model: BssmModel, particles: int, method: str = "psi", seed: int | None = None, L=None) -> float:
    """R/loglik.R:73-100 (nongaussian): method 'psi' or 'bsf'; particles = 0 gives the
    Gaussian-approximation log-likelihood."""
    particles = check_intmax(particles, "particles", positive=False)
    seed = check_intmax(_default_seed() if seed is None else seed, "seed", False, max=_INT_MAX)
    if method not in ("psi", "bsf", "spdk", "ekf"):
        raise ValueError("'arg' should be one of 'psi', 'bsf', 'spdk', 'ekf'")
    if method == "ekf" and not model.inherits("ssm_nlg"):
        raise ValueError("Method 'ekf' is only applicable to ssm_nlg models.")
    if method == "spdk":
        raise NotImplementedError("SPDK importance sampling is not a particle-filter path (SURVEY.md 2a N23).")
    if model.inherits("ssm_nlg") or model.inherits("ssm_sde"):
        return float(logLik_batch(model, particles, 1, method=method, seed=seed, L=L)[0])
    if not model.inherits("nongaussian"):
        raise TypeError("logLik(particles, method) is for nongaussian, ssm_nlg and ssm_sde models")
    arr, smod, keep = _prep(model, True)
    ll = C.c_double(0.0)
    rc = _lib.check(_lib.lib().bssm_nongaussian_loglik(C.byref(smod), C.c_uint32(particles),
                                                        C.c_uint32(1 if method == "psi" else 2), C.c_uint32(seed),
                                                        model_type(model), C.byref(ll)))
    _warn_if_degenerate(rc)
    return ll.value


def logLik_batch(model: BssmModel, particles: int, n_rep: int, method: str = "bsf", seed: int = 1, stream0: int = 0, L=None):
    """n_rep independent log-likelihood estimates in one device batch (streams stream0..stream0+n_rep-1)."""
    if model.inherits("ssm_nlg") or model.inherits("ssm_sde"):
        smod, keep = _lib.pack_fn_model(model, L)
        out = np.zeros(n_rep)
        meth = 1 if (method == "psi" and model.inherits("ssm_nlg")) else (4 if (method == "ekf" and model.inherits("ssm_nlg")) else 2)
        rc = _lib.check(_lib.lib().bssm_fn_loglik_batch(C.byref(smod), C.c_uint32(particles), C.c_uint32(meth), C.c_uint32(seed),
                                                        C.c_uint32(stream0), C.c_uint32(n_rep), _lib.dptr(out)))
        _warn_if_degenerate(rc)
        return out
    arr, smod, keep = _prep(model, True)
    out = np.zeros(n_rep)
    rc = _lib.check(_lib.lib().bssm_nongaussian_loglik_batch(
        C.byref(smod), C.c_uint32(particles), C.c_uint32(1 if method == "psi" else 2), C.c_uint32(seed),
        C.c_uint32(stream0), C.c_uint32(n_rep), _lib.dptr(out)))
    _warn_if_degenerate(rc)
    return out


def psi_setup(model: BssmModel, is_mode=None):
    """Gaussian approximation + conditional factors used by the psi-APF (host side of the library)."""
    if model.inherits("ssm_nlg"):
        smod, keep = _lib.pack_fn_model(model)
        n, m = model.n, int(model.get("n_states", 1))
        res = dict(mode=_alloc((m, n)), scales=np.zeros(n), alphahat=_alloc((m, n + 1)), Vt=_alloc((m, m, n + 1)),
                   Ct=_alloc((m, m, n + 1)), Z=_alloc((m, n)), T=_alloc((m, m, n)), D=np.zeros(n), C=_alloc((m, n)))
        al = C.c_double(0.0)
        mode_in = None if is_mode is None else np.asfortranarray(is_mode, dtype=np.float64)
        _lib.check(_lib.lib().bssm_fn_psi_setup(
            C.byref(smod), _lib.dptr(mode_in), C.c_int32(int(model.get("max_iter", 100))), C.c_double(float(model.get("conv_tol", 1e-8))),
            _lib.dptr(res["mode"]), _lib.dptr(res["scales"]), C.byref(al), _lib.dptr(res["alphahat"]), _lib.dptr(res["Vt"]),
            _lib.dptr(res["Ct"]), _lib.dptr(res["Z"]), _lib.dptr(res["T"]), _lib.dptr(res["D"]), _lib.dptr(res["C"])))
        res["approx_loglik"] = al.value
        return res
    arr, smod, keep = _prep(model, True)
    n, m, p = arr["n"], arr["m"], arr.get("p", 1)
    res = dict(approx_y=np.zeros(n * p), approx_H=np.zeros(n * p), mode=np.zeros(n * p), scales=np.zeros(n),
               alphahat=_alloc((m, n + 1)), Vt=_alloc((m, m, n + 1)), Ct=_alloc((m, m, n + 1)))
    al = C.c_double(0.0)
    mode_in = None if is_mode is None else np.ascontiguousarray(is_mode, dtype=np.float64)
    _lib.check(_lib.lib().bssm_psi_setup(C.byref(smod), _lib.dptr(mode_in), _lib.dptr(res["approx_y"]),
                                          _lib.dptr(res["approx_H"]), _lib.dptr(res["mode"]), _lib.dptr(res["scales"]),
                                          C.byref(al), _lib.dptr(res["alphahat"]), _lib.dptr(res["Vt"]),
                                          _lib.dptr(res["Ct"])))
    res["approx_loglik"] = al.value
    if p > 1:  # the library holds these p x n
        for k in ("approx_y", "approx_H", "mode"):
            res[k] = res[k].reshape((p, n), order="F")
    return res


def stratified_sample(p, r, N=None):
    """R/RcppExports.R:228-230 `stratified_sample(p, r, N)` -> bssm_stratified_sample (src/stratified_sample.cpp:9-28).
    Returns (zero-based indices [N], the cumulative sum the reference leaves in p)."""
    p = np.array(p, dtype=np.float64).reshape(-1)
    r = np.ascontiguousarray(r, dtype=np.float64).reshape(-1)
    N = r.size if N is None else int(N)
    if N > r.size:
        raise ValueError("'r' must hold at least N uniforms")
    out = np.zeros(N, dtype=np.uint32)
    _lib.check(_lib.lib().bssm_stratified_sample(_lib.dptr(p), _lib.dptr(r), C.c_uint32(p.size), C.c_uint32(N),
                                                  out.ctypes.data_as(_lib.c_u32_p)))
    return out, p


def ekpf_filter(model: BssmModel, particles: int, seed: int | None = None, return_indices: bool = False):
    """R/ekpf_filter.R -> ekpf (src/R_ekpf.cpp:11-54): the EKF-proposal particle filter of an ssm_nlg model; outputs as
    bootstrap_filter."""
    particles = check_intmax(particles, "particles")
    seed = check_intmax(_default_seed() if seed is None else seed, "seed", False, max=_INT_MAX)
    if not model.inherits("ssm_nlg"):
        raise TypeError("ekpf_filter is for ssm_nlg models")
    return _functor_filter(model, particles, seed, False, None, return_indices, ekpf=True)
