"""ctypes wrapper of the CPU oracle (oracle/bssm_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg
of bench.py.  Nothing in bssm_amd/ imports this module.  Parity status: see bssm_oracle.h
(deterministic half pinned by the values the reference's own tests state; particle systems "parity unpinned" at the
bit level, pinned statistically and analytically).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbssm_oracle.so")

c_double_p = C.POINTER(C.c_double)
c_uint_p = C.POINTER(C.c_uint)


class OrcModel(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("n", "m", "k", "Ztv", "Ttv", "Rtv", "Dtv", "Ctv", "Htv", "distribution")] + \
               [(k, c_double_p) for k in ("y", "Z", "T", "R", "a1", "P1", "D", "C", "xbeta", "u", "H")] + \
               [("phi", C.c_double), ("initial_mode", c_double_p), ("max_iter", C.c_int), ("conv_tol", C.c_double),
                ("local_approx", C.c_int)]


class OrcRng(C.Structure):
    _fields_ = [("mode", C.c_int), ("seed", C.c_uint32), ("stream", C.c_uint32), ("init_normals", c_double_p),
                ("unif", c_double_p), ("normals", c_double_p), ("N", C.c_int), ("kk", C.c_int)]


class OrcPsiSetup(C.Structure):
    _fields_ = [("n", C.c_int), ("m", C.c_int), ("approx_y", c_double_p), ("approx_H", c_double_p),
                ("mode", c_double_p), ("scales", c_double_p), ("approx_loglik", C.c_double),
                ("alphahat", c_double_p), ("Vt", c_double_p), ("Ct", c_double_p), ("laplace_iters", C.c_int)]


class OrcUlg(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("n", "m", "k", "Ztv", "Ttv", "Rtv", "Dtv", "Ctv", "Htv")] + \
               [(k, c_double_p) for k in ("y", "Z", "T", "R", "a1", "P1", "D", "C", "xbeta", "H")]


class OrcIsJob(C.Structure):
    _fields_ = [("n_stored", C.c_int), ("models", C.POINTER(OrcModel)), ("modes", c_double_p),
                ("approx_loglik", c_double_p), ("counts", c_uint_p), ("method", C.c_int), ("is_type", C.c_int),
                ("output_type", C.c_int), ("nsim", C.c_uint), ("seed", C.c_uint32), ("n_threads", C.c_int)]


_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "bssm_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(
            os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "bssm_oracle.h"))):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.orc_ung_bsf_filter.restype = C.c_double
        L.orc_ung_psi_filter.restype = C.c_double
        L.orc_ulg_loglik.restype = C.c_double
        L.orc_ung_const_term.restype = C.c_double
        L.orc_rng_uniform.restype = C.c_double
        L.orc_psi_setup_alloc.restype = C.POINTER(OrcPsiSetup)
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(c_double_p)


def pack_model(arr: dict):
    """dict from bssm_amd.models.model_arrays -> (OrcModel, keepalive)"""
    s = OrcModel()
    for k in ("n", "m", "k", "Ztv", "Ttv", "Rtv", "Dtv", "Ctv", "Htv", "distribution", "max_iter", "local_approx"):
        setattr(s, k, int(arr[k]))
    s.phi = float(arr["phi"])
    s.conv_tol = float(arr["conv_tol"])
    keep = []
    for k in ("y", "Z", "T", "R", "a1", "P1", "D", "C", "xbeta", "u", "H", "initial_mode"):
        a = arr.get(k)
        if a is not None:
            a = np.asfortranarray(a, dtype=np.float64)
            keep.append(a)
        setattr(s, k, _p(a))
    return s, keep


def make_rng(seed=1, stream=0, injected=None):
    r = OrcRng()
    keep = []
    if injected is None:
        r.mode, r.seed, r.stream = 0, seed, stream
    else:
        init_normals, unif, normals = (np.ascontiguousarray(a, dtype=np.float64) for a in injected)
        keep = [init_normals, unif, normals]
        r.mode = 1
        r.init_normals, r.unif, r.normals = _p(init_normals), _p(unif), _p(normals)
        r.N = unif.shape[1]
        r.kk = normals.shape[2]
    return r, keep


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return list(o)


def bsf_filter(arr: dict, nsim: int, seed=1, stream=0, injected=None):
    """orc_ung_bsf_filter -> dict(logLik, alpha m x (n+1) x N, weights N x (n+1), indices N x n)"""
    L = lib()
    mod, keep = pack_model(arr)
    rng, keep2 = make_rng(seed, stream, injected)
    n, m = arr["n"], arr["m"]
    alpha = np.zeros((m, n + 1, nsim), order="F")
    weights = np.zeros((nsim, n + 1), order="F")
    indices = np.zeros((nsim, n), dtype=np.uint32, order="F")
    ll = L.orc_ung_bsf_filter(C.byref(mod), C.c_uint(nsim), C.byref(rng), _p(alpha), _p(weights),
                              indices.ctypes.data_as(c_uint_p))
    return dict(logLik=ll, alpha=alpha, weights=weights, indices=indices)


def psi_setup(arr: dict, is_mode=None):
    L = lib()
    mod, keep = pack_model(arr)
    n, m = arr["n"], arr["m"]
    su = L.orc_psi_setup_alloc(n, m)
    mode = None if is_mode is None else np.ascontiguousarray(is_mode, dtype=np.float64)
    L.orc_ung_psi_setup(C.byref(mod), _p(mode), su)
    s = su.contents
    out = dict(
        approx_y=np.ctypeslib.as_array(s.approx_y, (n,)).copy(), approx_H=np.ctypeslib.as_array(s.approx_H, (n,)).copy(),
        mode=np.ctypeslib.as_array(s.mode, (n,)).copy(), scales=np.ctypeslib.as_array(s.scales, (n,)).copy(),
        approx_loglik=s.approx_loglik,
        alphahat=np.ctypeslib.as_array(s.alphahat, ((n + 1) * m,)).copy().reshape((m, n + 1), order="F"),
        Vt=np.ctypeslib.as_array(s.Vt, ((n + 1) * m * m,)).copy().reshape((m, m, n + 1), order="F"),
        Ct=np.ctypeslib.as_array(s.Ct, ((n + 1) * m * m,)).copy().reshape((m, m, n + 1), order="F"),
        iterations=s.laplace_iters, _handle=su)
    return out


def psi_filter(arr: dict, nsim: int, seed=1, stream=0, is_mode=None, setup=None):
    L = lib()
    mod, keep = pack_model(arr)
    su = setup if setup is not None else psi_setup(arr, is_mode)
    rng, keep2 = make_rng(seed, stream)
    n, m = arr["n"], arr["m"]
    alpha = np.zeros((m, n + 1, nsim), order="F")
    weights = np.zeros((nsim, n + 1), order="F")
    indices = np.zeros((nsim, n), dtype=np.uint32, order="F")
    ll = L.orc_ung_psi_filter(C.byref(mod), su["_handle"], C.c_uint(nsim), C.byref(rng), _p(alpha), _p(weights),
                              indices.ctypes.data_as(c_uint_p))
    return dict(logLik=ll, alpha=alpha, weights=weights, indices=indices, setup=su)


def filter_smoother(alpha, indices):
    L = lib()
    m, n1, N = alpha.shape
    a = np.asfortranarray(alpha.copy())
    idx = np.asfortranarray(indices, dtype=np.uint32)
    L.orc_filter_smoother(_p(a), idx.ctypes.data_as(c_uint_p), m, n1 - 1, C.c_uint(N))
    return a


def summary(alpha):
    L = lib()
    m, n1, N = alpha.shape
    a = np.asfortranarray(alpha)
    mean = np.zeros((m, n1), order="F")
    cov = np.zeros((m, m, n1), order="F")
    L.orc_summary(_p(a), m, n1, C.c_uint(N), _p(mean), _p(cov))
    return mean, cov


def filter_summary(alpha, weights):
    L = lib()
    m, n1, N = alpha.shape
    n = n1 - 1
    a = np.asfortranarray(alpha)
    w = np.asfortranarray(weights)
    at = np.zeros((m, n + 1), order="F"); att = np.zeros((m, n), order="F")
    Pt = np.zeros((m, m, n + 1), order="F"); Ptt = np.zeros((m, m, n), order="F")
    L.orc_filter_summary(_p(a), m, n, C.c_uint(N), _p(w), _p(at), _p(att), _p(Pt), _p(Ptt))
    return at, att, Pt, Ptt


def dmvnorm_cov(x, mean, S):
    """src/dmvnorm.cpp:4-58 with lwr = FALSE, logd = TRUE"""
    L = lib()
    L.orc_dmvnorm_cov.restype = C.c_double
    x = np.ascontiguousarray(x, dtype=np.float64); mean = np.ascontiguousarray(mean, dtype=np.float64)
    S = np.asfortranarray(S, dtype=np.float64)
    return float(L.orc_dmvnorm_cov(_p(x), _p(mean), _p(S), C.c_int(x.size)))


def stratified_sample(p, r, N=None):
    """src/stratified_sample.cpp:9-28: (zero-based indices [N], cumulative sum left in p)"""
    L = lib()
    p = np.array(p, dtype=np.float64)
    r = np.ascontiguousarray(r, dtype=np.float64)
    N = r.size if N is None else int(N)
    out = np.zeros(N, dtype=np.uint32)
    L.orc_stratified_sample_n(_p(p), C.c_uint(p.size), _p(r), C.c_uint(N), out.ctypes.data_as(c_uint_p))
    return out, p


def _pack_ulg(arr: dict, y=None, H=None):
    g = OrcUlg()
    for k in ("n", "m", "k", "Ztv", "Ttv", "Rtv", "Dtv", "Ctv"):
        setattr(g, k, int(arr[k]))
    yy = np.ascontiguousarray(arr["y"] if y is None else y, dtype=np.float64)
    HH = np.ascontiguousarray(arr["H"] if H is None else H, dtype=np.float64)
    g.Htv = int(HH.size > 1)
    keep = [yy, HH]
    g.y, g.H = _p(yy), _p(HH)
    for k in ("Z", "T", "R", "a1", "P1", "D", "C", "xbeta"):
        a = np.asfortranarray(arr[k], dtype=np.float64)
        keep.append(a)
        setattr(g, k, _p(a))
    return g, keep


def ulg_loglik(arr: dict):
    g, keep = _pack_ulg(arr)
    return lib().orc_ulg_loglik(C.byref(g))


def ulg_smoother(arr: dict):
    """Exact Kalman smoother of a linear-Gaussian model: (alphahat m x (n+1), Vt m x m x (n+1), ccov)."""
    g, keep = _pack_ulg(arr)
    n, m = arr["n"], arr["m"]
    at = np.zeros((m, n + 1), order="F")
    Pt = np.zeros((m, m, n + 1), order="F")
    cc = np.zeros((m, m, n + 1), order="F")
    lib().orc_ulg_smoother_ccov(C.byref(g), _p(at), _p(Pt), _p(cc))
    return at, Pt, cc


def const_term(arr: dict):
    mod, keep = pack_model(arr)
    return lib().orc_ung_const_term(C.byref(mod))


def is_correction(arrs, nsim, seed, method, counts, modes=None, approx_loglik=None, is_type=2, output_type=3,
                  n_threads=1):
    """orc_is_correction over a list of per-draw model dicts."""
    L = lib()
    ns = len(arrs)
    Models = OrcModel * ns
    packed, keep = [], []
    for a in arrs:
        s, k = pack_model(a)
        packed.append(s); keep.append(k)
    models = Models(*packed)
    n, m = arrs[0]["n"], arrs[0]["m"]
    job = OrcIsJob()
    job.n_stored = ns
    job.models = models
    mo = None if modes is None else np.asfortranarray(modes, dtype=np.float64)
    al = None if approx_loglik is None else np.ascontiguousarray(approx_loglik, dtype=np.float64)
    cn = np.ascontiguousarray(counts, dtype=np.uint32)
    job.modes, job.approx_loglik, job.counts = _p(mo), _p(al), cn.ctypes.data_as(c_uint_p)
    job.method, job.is_type, job.output_type = method, is_type, output_type
    job.nsim, job.seed, job.n_threads = nsim, seed, n_threads
    weights = np.zeros(ns)
    ll = np.zeros(ns)
    alpha = np.zeros((n + 1, m, ns), order="F") if output_type == 1 else None
    alphahat = np.zeros((m, n + 1), order="F") if output_type == 2 else None
    Vt = np.zeros((m, m, n + 1), order="F") if output_type == 2 else None
    L.orc_is_correction(C.byref(job), _p(weights), _p(ll), _p(alpha), _p(alphahat), _p(Vt))
    return dict(weights=weights, loglik=ll, alpha=alpha, alphahat=alphahat, Vt=Vt)


# ------------------------------------------------------------------ ssm_mng
class OrcMng(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("n", "m", "k", "p", "Ztv", "Ttv", "Rtv", "Dtv", "Ctv")] + \
               [("distribution", C.POINTER(C.c_int))] + \
               [(k, c_double_p) for k in ("y", "Z", "T", "R", "a1", "P1", "D", "C", "u", "phi", "initial_mode")] + \
               [("max_iter", C.c_int), ("conv_tol", C.c_double)]


class OrcMngSetup(C.Structure):
    _fields_ = [("n", C.c_int), ("m", C.c_int), ("p", C.c_int), ("approx_y", c_double_p), ("approx_H", c_double_p),
                ("mode", c_double_p), ("scales", c_double_p), ("approx_loglik", C.c_double), ("alphahat", c_double_p),
                ("Vt", c_double_p), ("Ct", c_double_p), ("laplace_iters", C.c_int)]


def pack_mng(arr: dict):
    """dict from bssm_amd.models.model_arrays (ssm_mng) -> OrcMng with the p-major arrays the reference works on"""
    s = OrcMng()
    for k in ("n", "m", "k", "p", "Ztv", "Ttv", "Rtv", "Dtv", "Ctv", "max_iter"):
        setattr(s, k, int(arr[k]))
    s.conv_tol = float(arr["conv_tol"])
    keep = []
    dist = np.ascontiguousarray(arr["distribution_p"], dtype=np.int32)
    keep.append(dist)
    s.distribution = dist.ctypes.data_as(C.POINTER(C.c_int))
    conv = dict(y=np.asfortranarray(arr["y"].T), u=np.asfortranarray(arr["u"].T),
                initial_mode=np.asfortranarray(arr["initial_mode"].T), phi=np.ascontiguousarray(arr["phi_p"]))
    for k in ("Z", "T", "R", "a1", "P1", "D", "C"):
        conv[k] = np.asfortranarray(arr[k], dtype=np.float64)
    for k, a in conv.items():
        a = np.asfortranarray(a, dtype=np.float64)
        keep.append(a)
        setattr(s, k, _p(a))
    return s, keep


def _mng_lib():
    L = lib()
    L.orc_mng_bsf_filter.restype = C.c_double
    L.orc_mng_psi_filter.restype = C.c_double
    L.orc_mng_const_term.restype = C.c_double
    L.orc_mlg_loglik.restype = C.c_double
    L.orc_mng_setup_alloc.restype = C.POINTER(OrcMngSetup)
    return L


def mng_bsf_filter(arr: dict, nsim: int, seed=1, stream=0):
    L = _mng_lib()
    mod, keep = pack_mng(arr)
    rng, _ = make_rng(seed, stream)
    n, m = arr["n"], arr["m"]
    alpha = np.zeros((m, n + 1, nsim), order="F"); weights = np.zeros((nsim, n + 1), order="F")
    indices = np.zeros((nsim, n), dtype=np.uint32, order="F")
    ll = L.orc_mng_bsf_filter(C.byref(mod), C.c_uint(nsim), C.byref(rng), _p(alpha), _p(weights), indices.ctypes.data_as(c_uint_p))
    return dict(logLik=ll, alpha=alpha, weights=weights, indices=indices)


def mng_psi_setup(arr: dict, is_mode=None):
    """is_mode: p x n (mode_storage layout) or None"""
    L = _mng_lib()
    mod, keep = pack_mng(arr)
    n, m, p = arr["n"], arr["m"], arr["p"]
    su = L.orc_mng_setup_alloc(n, m, p)
    mode = None if is_mode is None else np.asfortranarray(np.asarray(is_mode, dtype=np.float64).reshape((p, n), order="F"))
    rc = L.orc_mng_psi_setup(C.byref(mod), _p(mode), su)
    s = su.contents
    g = lambda ptr, shape: np.ctypeslib.as_array(ptr, (int(np.prod(shape)),)).copy().reshape(shape, order="F")
    return dict(approx_y=g(s.approx_y, (p, n)), approx_H=g(s.approx_H, (p, n)), mode=g(s.mode, (p, n)), scales=g(s.scales, (n,)),
                approx_loglik=s.approx_loglik, alphahat=g(s.alphahat, (m, n + 1)), Vt=g(s.Vt, (m, m, n + 1)),
                Ct=g(s.Ct, (m, m, n + 1)), iterations=s.laplace_iters, rc=rc, _handle=su)


def mng_psi_filter(arr: dict, nsim: int, seed=1, stream=0, is_mode=None, setup=None):
    L = _mng_lib()
    mod, keep = pack_mng(arr)
    su = setup if setup is not None else mng_psi_setup(arr, is_mode)
    rng, _ = make_rng(seed, stream)
    n, m = arr["n"], arr["m"]
    alpha = np.zeros((m, n + 1, nsim), order="F"); weights = np.zeros((nsim, n + 1), order="F")
    indices = np.zeros((nsim, n), dtype=np.uint32, order="F")
    ll = L.orc_mng_psi_filter(C.byref(mod), su["_handle"], C.c_uint(nsim), C.byref(rng), _p(alpha), _p(weights),
                              indices.ctypes.data_as(c_uint_p))
    return dict(logLik=ll, alpha=alpha, weights=weights, indices=indices, setup=su)


# ------------------------------------------------------------------ ssm_nlg / ssm_sde built-in functors
class OrcFn(C.Structure):
    _fields_ = [("id", C.c_int), ("n", C.c_int), ("y", c_double_p), ("theta", C.c_double * 8), ("known", C.c_double * 8),
                ("x0", C.c_double), ("positive", C.c_int), ("L", C.c_int)]


def fn_ekf_loglik(model: dict, iekf_iter: int = 0) -> float:
    """ekf(model, iekf_iter)$logLik (src/model_ssm_nlg.cpp:284-381)"""
    L = lib()
    L.orc_fn_ekf_loglik.restype = C.c_double
    s = OrcFn()
    y = np.ascontiguousarray(model["y"], dtype=np.float64)
    s.id, s.n, s.y = int(model["model_id"]), y.size, _p(y)
    for i, v in enumerate(np.asarray(model["theta"], dtype=float)): s.theta[i] = v
    for i, v in enumerate(np.asarray(model.get("known_params", []), dtype=float)): s.known[i] = v
    return float(L.orc_fn_ekf_loglik(C.byref(s), C.c_int(iekf_iter)))


def fn_bsf_filter(model: dict, nsim: int, seed=1, stream=0):
    """model: dict(id, y, theta, known_params, x0, positive, L) as bssm_amd.models.ssm_nlg / ssm_sde produce"""
    L = lib()
    L.orc_fn_bsf_filter.restype = C.c_double
    s = OrcFn()
    y = np.ascontiguousarray(model["y"], dtype=np.float64)
    s.id, s.n, s.y = int(model["model_id"]), y.size, _p(y)
    for i, v in enumerate(np.asarray(model["theta"], dtype=float)): s.theta[i] = v
    for i, v in enumerate(np.asarray(model.get("known_params", []), dtype=float)): s.known[i] = v
    s.x0, s.positive, s.L = float(model.get("x0", 0.0)), int(bool(model.get("positive", False))), int(model.get("L", 0))
    rng, _ = make_rng(seed, stream)
    n, m = y.size, 2 if s.id == 1 else 1
    alpha = np.zeros((m, n + 1, nsim), order="F"); weights = np.zeros((nsim, n + 1), order="F")
    indices = np.zeros((nsim, n), dtype=np.uint32, order="F")
    ll = L.orc_fn_bsf_filter(C.byref(s), C.c_uint(nsim), C.byref(rng), _p(alpha), _p(weights), indices.ctypes.data_as(c_uint_p))
    return dict(logLik=ll, alpha=alpha, weights=weights, indices=indices)


class OrcFnSetup(C.Structure):
    _fields_ = [("n", C.c_int), ("m", C.c_int), ("Z", c_double_p), ("T", c_double_p), ("D", c_double_p), ("C", c_double_p),
                ("H", C.c_double), ("R", C.c_double * 4), ("mode", c_double_p), ("scales", c_double_p),
                ("approx_loglik", C.c_double), ("alphahat", c_double_p), ("Vt", c_double_p), ("Ct", c_double_p),
                ("iters", C.c_int), ("ok", C.c_int)]


def _pack_fn(model):
    s = OrcFn()
    y = np.ascontiguousarray(model["y"], dtype=np.float64)
    s.id, s.n, s.y = int(model["model_id"]), y.size, _p(y)
    for i, v in enumerate(np.asarray(model["theta"], dtype=float)): s.theta[i] = v
    for i, v in enumerate(np.asarray(model.get("known_params", []), dtype=float)): s.known[i] = v
    s.x0, s.positive, s.L = float(model.get("x0", 0.0)), int(bool(model.get("positive", False))), int(model.get("L", 0))
    return s, [y]


def fn_is_logweights(model: dict, theta, modes, nsim: int, seed=1, first_stream=0, n_threads=1):
    """log-weights of the psi-APF IS correction of an ssm_nlg model, one draw per row of theta, OpenMP over the draws with
    n_threads threads as approx_mcmc::is_correction_psi does (src/approx_mcmc.cpp:364-369)"""
    L = lib()
    theta = np.atleast_2d(np.asarray(theta, dtype=np.float64))
    cnt = theta.shape[0]
    y = np.ascontiguousarray(model["y"], dtype=np.float64)
    mods = (OrcFn * cnt)()
    for i in range(cnt):
        s = mods[i]
        s.id, s.n, s.y = int(model["model_id"]), y.size, _p(y)
        for q, v in enumerate(theta[i]): s.theta[q] = v
        for q, v in enumerate(np.asarray(model.get("known_params", []), dtype=float)): s.known[q] = v
    modes = np.asfortranarray(modes, dtype=np.float64)
    out = np.zeros(cnt)
    L.orc_fn_is_logweights(mods, C.c_int(cnt), _p(modes), C.c_uint(nsim), C.c_uint32(seed), C.c_uint32(first_stream),
                           C.c_int(n_threads), _p(out))
    return out


def fn_ekpf_filter(model: dict, nsim: int, seed=1, stream=0):
    """ssm_nlg::ekf_filter (src/model_ssm_nlg.cpp:1148-1293)"""
    L = lib()
    L.orc_fn_ekpf_filter.restype = C.c_double
    s = OrcFn()
    y = np.ascontiguousarray(model["y"], dtype=np.float64)
    s.id, s.n, s.y = int(model["model_id"]), y.size, _p(y)
    for i, v in enumerate(np.asarray(model["theta"], dtype=float)): s.theta[i] = v
    for i, v in enumerate(np.asarray(model.get("known_params", []), dtype=float)): s.known[i] = v
    rng, _ = make_rng(seed, stream)
    n, m = y.size, 2 if s.id == 1 else 1
    alpha = np.zeros((m, n + 1, nsim), order="F"); weights = np.zeros((nsim, n + 1), order="F")
    indices = np.zeros((nsim, n), dtype=np.uint32, order="F")
    ll = L.orc_fn_ekpf_filter(C.byref(s), C.c_uint(nsim), C.byref(rng), _p(alpha), _p(weights), indices.ctypes.data_as(c_uint_p))
    return dict(logLik=ll, alpha=alpha, weights=weights, indices=indices)


def fn_psi_setup(model: dict, is_mode=None, max_iter=100, conv_tol=1e-8):
    """is_mode: m x n state mode (mode_storage of the approximate chain) or None"""
    L = lib()
    L.orc_fn_setup_alloc.restype = C.POINTER(OrcFnSetup)
    s, keep = _pack_fn(model)
    n, m = s.n, 2 if s.id == 1 else 1
    su = L.orc_fn_setup_alloc(n, m)
    mode = None if is_mode is None else np.asfortranarray(np.asarray(is_mode, dtype=np.float64).reshape((m, n), order="F"))
    rc = L.orc_fn_psi_setup(C.byref(s), _p(mode), C.c_int(max_iter), C.c_double(conv_tol), su)
    c = su.contents
    g = lambda ptr, shape: np.ctypeslib.as_array(ptr, (int(np.prod(shape)),)).copy().reshape(shape, order="F")
    return dict(rc=rc, ok=c.ok, iters=c.iters, approx_loglik=c.approx_loglik, mode=g(c.mode, (m, n)), scales=g(c.scales, (n,)),
                Z=g(c.Z, (m, n)), T=g(c.T, (m, m, n)), D=g(c.D, (n,)), C=g(c.C, (m, n)), H=c.H,
                alphahat=g(c.alphahat, (m, n + 1)), Vt=g(c.Vt, (m, m, n + 1)), Ct=g(c.Ct, (m, m, n + 1)), _handle=su)


def fn_psi_filter(model: dict, nsim: int, seed=1, stream=0, is_mode=None, setup=None):
    L = lib()
    L.orc_fn_psi_filter.restype = C.c_double
    s, keep = _pack_fn(model)
    su = setup if setup is not None else fn_psi_setup(model, is_mode)
    rng, _ = make_rng(seed, stream)
    n, m = s.n, 2 if s.id == 1 else 1
    alpha = np.zeros((m, n + 1, nsim), order="F"); weights = np.zeros((nsim, n + 1), order="F")
    indices = np.zeros((nsim, n), dtype=np.uint32, order="F")
    ll = L.orc_fn_psi_filter(C.byref(s), su["_handle"], C.c_uint(nsim), C.byref(rng), _p(alpha), _p(weights),
                             indices.ctypes.data_as(c_uint_p))
    return dict(logLik=ll, alpha=alpha, weights=weights, indices=indices, setup=su)


def ulg_psi_filter(arr: dict, nsim: int, seed=1):
    """ssm_ulg::psi_filter (gaussian_psi_smoother): alpha m x (n+1) x N"""
    g, keep = _pack_ulg(arr)
    rng, _ = make_rng(seed, 0)
    n, m = arr["n"], arr["m"]
    alpha = np.zeros((m, n + 1, nsim), order="F")
    lib().orc_ulg_psi_filter(C.byref(g), C.c_uint(nsim), C.byref(rng), _p(alpha))
    return alpha
