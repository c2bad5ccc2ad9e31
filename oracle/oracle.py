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


class OrcFollow(C.Structure):
    _fields_ = [("table", c_uint_p), ("tol", C.c_double), ("ties", C.c_uint), ("unexplained", C.c_uint),
                ("max_margin", C.c_double), ("first_tie_t", C.c_int), ("first_tie_slot", C.c_int)]


def tie_tolerance(N: int) -> float:
    """In strata (1 / N): twice the worst-case rounding error of a sequential sum of N terms in (0, 1], (N - 1) 2^-53,
    seen from the top stratum (x N) -- what two correctly rounded evaluations of the reference's cumulative sum may differ
    by.  N = 65 536: 9.5e-7 of a stratum; measured margins at that size: 3e-10 ... 2.3e-8 (profiles/r04_flip_probe.json)."""
    return max(2.0 * N * N * 2.0 ** -53, 1e-10)


class following:
    """with following(table, N) as f: ...  -- the filters called inside (on this thread) adopt `table`'s ancestor at
    explained ties (bssm_oracle.c, "tie-following"); afterwards f.ties, f.unexplained, f.max_margin, f.first_tie.
    table: (n, N) C-contiguous uint32 (the device's layout) or (N, n) Fortran-ordered (the oracle's) -- the same bytes."""

    def __init__(self, table, N, tol=None):
        t = np.asarray(table, dtype=np.uint32)
        assert t.ndim == 2 and N in t.shape, t.shape
        if t.shape[0] == N and t.shape[1] != N:      # the oracle's N x n: wanted column-major
            t = np.asfortranarray(t)
        else:                                         # the device's n x N: wanted row-major (the same bytes)
            t = np.ascontiguousarray(t)
        self._table = t
        self._f = OrcFollow()
        self._f.table = self._table.ctypes.data_as(c_uint_p)
        self._f.tol = tie_tolerance(N) if tol is None else tol

    def __enter__(self):
        lib().orc_follow_begin(C.byref(self._f))
        return self

    def __exit__(self, *exc):
        lib().orc_follow_end()
        return False

    ties = property(lambda self: int(self._f.ties))
    unexplained = property(lambda self: int(self._f.unexplained))
    max_margin = property(lambda self: float(self._f.max_margin))
    first_tie = property(lambda self: (int(self._f.first_tie_t), int(self._f.first_tie_slot)))
    tol = property(lambda self: float(self._f.tol))


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


def milstein(model: dict, t_end: float, seed=1) -> float:
    """R_milstein (src/R_milstein.cpp:4-22): the terminal value of one Milstein path from x0 over [0, t_end]"""
    L = lib()
    L.orc_milstein.restype = C.c_double
    s = OrcFn()
    s.id = int(model["model_id"])
    for i, v in enumerate(np.asarray(model["theta"], dtype=float)): s.theta[i] = v
    s.x0, s.positive, s.L = float(model.get("x0", 0.0)), int(bool(model.get("positive", False))), int(model.get("L", 0))
    rng, _ = make_rng(seed, 0)
    return float(L.orc_milstein(C.byref(s), C.c_double(t_end), C.byref(rng)))


_NLG_FN = C.CFUNCTYPE(None, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p)


class OrcNlgGeneral(C.Structure):
    _fields_ = [("n", C.c_int), ("m", C.c_int), ("k", C.c_int), ("y", c_double_p), ("theta", c_double_p), ("known", c_double_p),
                ("a1", c_double_p), ("P1", c_double_p), ("Z_fn", _NLG_FN), ("H_fn", _NLG_FN), ("T_fn", _NLG_FN), ("R_fn", _NLG_FN)]


def nlg_general_bsf_filter(y, m, k, theta, known, a1, P1, Z_fn, H_fn, T_fn, R_fn, nsim: int, seed=1, stream=0):
    """ssm_nlg::bsf_filter (src/model_ssm_nlg.cpp:1066-1146) for a model given as Python functions of
    (t, alpha, theta, known) -- the reference's function-pointer interface (src/model_ssm_nlg.h:10-21): Z_fn and H_fn
    (observation sd) return scalars, T_fn an m-vector, R_fn an m x k matrix; H and R may depend on the state."""
    L = lib()
    L.orc_nlg_general_bsf_filter.restype = C.c_double
    y = np.ascontiguousarray(y, dtype=np.float64)
    th = np.ascontiguousarray(theta, dtype=np.float64); kn = np.ascontiguousarray(known if len(known) else [0.0], dtype=np.float64)
    a1 = np.ascontiguousarray(a1, dtype=np.float64); P1 = np.asfortranarray(np.asarray(P1, dtype=np.float64).reshape(m, m))
    nth, nkn = th.size, kn.size

    def wrap(fn, cnt, mat=False):
        def cb(t, al, thp, knp, out):
            v = fn(t, np.ctypeslib.as_array(al, (m,)).copy(), np.ctypeslib.as_array(thp, (nth,)), np.ctypeslib.as_array(knp, (nkn,)))
            v = np.asarray(v, dtype=float)
            v = v.reshape(m, k).reshape(-1, order="F") if mat else v.reshape(-1)
            for q in range(cnt): out[q] = v[q]
        return _NLG_FN(cb)
    cbs = [wrap(Z_fn, 1), wrap(H_fn, 1), wrap(T_fn, m), wrap(R_fn, m * k, True)]
    s = OrcNlgGeneral(y.size, m, k, _p(y), _p(th), _p(kn), _p(a1), _p(P1), *cbs)
    rng, _ = make_rng(seed, stream)
    n = y.size
    alpha = np.zeros((m, n + 1, nsim), order="F"); weights = np.zeros((nsim, n + 1), order="F")
    indices = np.zeros((nsim, n), dtype=np.uint32, order="F")
    ll = L.orc_nlg_general_bsf_filter(C.byref(s), C.c_uint(nsim), C.byref(rng), _p(alpha), _p(weights), indices.ctypes.data_as(c_uint_p))
    return dict(logLik=float(ll), alpha=alpha, weights=weights, indices=indices)


class OrcNlgGeneralPsi(C.Structure):
    _fields_ = [("base", OrcNlgGeneral), ("Z_gn", _NLG_FN), ("T_gn", _NLG_FN)]


class OrcNlgSetup(C.Structure):
    _fields_ = [("n", C.c_int), ("m", C.c_int), ("k", C.c_int), ("Z", c_double_p), ("T", c_double_p), ("D", c_double_p),
                ("C", c_double_p), ("H", c_double_p), ("R", c_double_p), ("mode", c_double_p), ("scales", c_double_p),
                ("approx_loglik", C.c_double), ("alphahat", c_double_p), ("Vt", c_double_p), ("Ct", c_double_p),
                ("iters", C.c_int), ("ok", C.c_int)]


def _nlg_general_struct(y, m, k, theta, known, a1, P1, Z_fn, H_fn, T_fn, R_fn, Z_gn, T_gn):
    """OrcNlgGeneralPsi for Python callbacks + the objects that must stay alive while it is used"""
    y = np.ascontiguousarray(y, dtype=np.float64)
    th = np.ascontiguousarray(theta, dtype=np.float64); kn = np.ascontiguousarray(known if len(known) else [0.0], dtype=np.float64)
    a1 = np.ascontiguousarray(a1, dtype=np.float64); P1 = np.asfortranarray(np.asarray(P1, dtype=np.float64).reshape(m, m))
    nth, nkn = th.size, kn.size

    def wrap(fn, shape):
        cnt = int(np.prod(shape))
        def cb(t, al, thp, knp, out):
            v = fn(t, np.ctypeslib.as_array(al, (m,)).copy(), np.ctypeslib.as_array(thp, (nth,)), np.ctypeslib.as_array(knp, (nkn,)))
            v = np.asarray(v, dtype=float).reshape(shape).reshape(-1, order="F")
            for q in range(cnt): out[q] = v[q]
        return _NLG_FN(cb)
    cbs = [wrap(Z_fn, (1,)), wrap(H_fn, (1,)), wrap(T_fn, (m,)), wrap(R_fn, (m, k)), wrap(Z_gn, (m,)), wrap(T_gn, (m, m))]
    base = OrcNlgGeneral(y.size, m, k, _p(y), _p(th), _p(kn), _p(a1), _p(P1), *cbs[:4])
    return OrcNlgGeneralPsi(base, cbs[4], cbs[5]), [y, th, kn, a1, P1, cbs]


def nlg_general_ekpf_filter(y, m, k, theta, known, a1, P1, Z_fn, H_fn, T_fn, R_fn, Z_gn, T_gn, nsim: int, seed=1, stream=0):
    """ssm_nlg::ekf_filter (EKF-proposal particle filter, src/model_ssm_nlg.cpp:1148-1293) for a model given as Python
    functions of (t, alpha, theta, known) incl. the observation Jacobian -- same interface as nlg_general_psi"""
    L = lib()
    L.orc_nlg_general_ekpf_filter.restype = C.c_double
    s, keep = _nlg_general_struct(y, m, k, theta, known, a1, P1, Z_fn, H_fn, T_fn, R_fn, Z_gn, T_gn)
    n = keep[0].size
    rng, _ = make_rng(seed, stream)
    alpha = np.zeros((m, n + 1, nsim), order="F"); weights = np.zeros((nsim, n + 1), order="F")
    indices = np.zeros((nsim, n), dtype=np.uint32, order="F")
    ll = L.orc_nlg_general_ekpf_filter(C.byref(s), C.c_uint(nsim), C.byref(rng), _p(alpha), _p(weights), indices.ctypes.data_as(c_uint_p))
    return dict(logLik=float(ll), alpha=alpha, weights=weights, indices=indices)


def nlg_general_psi(y, m, k, theta, known, a1, P1, Z_fn, H_fn, T_fn, R_fn, Z_gn, T_gn, nsim: int, seed=1, stream=0, is_mode=None,
                    max_iter=100, conv_tol=1e-8, filter=True):
    """ssm_nlg psi-APF (src/model_ssm_nlg.cpp:62-222 set-up, 854-935 scales / weights, 971-1063 filter) for a model given as
    Python functions of (t, alpha, theta, known) -- the reference's function-pointer interface incl. the Jacobians
    (src/model_ssm_nlg.h:10-21): Z_fn, H_fn scalars, T_fn m-vector, R_fn m x k, Z_gn m-vector (1 x m), T_gn m x m; H and R
    may depend on time and state.  Returns the set-up (mode, scales, approx_loglik, alphahat, Vt, Ct, D, Z, T, C) and,
    with filter=True, logLik / alpha / weights / indices; ekf_loglik is ekf(model)$logLik."""
    L = lib()
    L.orc_nlg_general_psi_filter.restype = C.c_double
    L.orc_nlg_general_ekf_loglik.restype = C.c_double
    L.orc_nlg_setup_alloc.restype = C.POINTER(OrcNlgSetup)
    y = np.ascontiguousarray(y, dtype=np.float64)
    n = y.size
    th = np.ascontiguousarray(theta, dtype=np.float64); kn = np.ascontiguousarray(known if len(known) else [0.0], dtype=np.float64)
    a1 = np.ascontiguousarray(a1, dtype=np.float64); P1 = np.asfortranarray(np.asarray(P1, dtype=np.float64).reshape(m, m))
    nth, nkn = th.size, kn.size

    def wrap(fn, shape):
        cnt = int(np.prod(shape))
        def cb(t, al, thp, knp, out):
            v = fn(t, np.ctypeslib.as_array(al, (m,)).copy(), np.ctypeslib.as_array(thp, (nth,)), np.ctypeslib.as_array(knp, (nkn,)))
            v = np.asarray(v, dtype=float).reshape(shape).reshape(-1, order="F")
            for q in range(cnt): out[q] = v[q]
        return _NLG_FN(cb)
    cbs = [wrap(Z_fn, (1,)), wrap(H_fn, (1,)), wrap(T_fn, (m,)), wrap(R_fn, (m, k))]
    base = OrcNlgGeneral(n, m, k, _p(y), _p(th), _p(kn), _p(a1), _p(P1), *cbs)
    g1, g2 = wrap(Z_gn, (m,)), wrap(T_gn, (m, m))
    s = OrcNlgGeneralPsi(base, g1, g2)
    su = L.orc_nlg_setup_alloc(n, m, k)
    mode = None if is_mode is None else np.ascontiguousarray(np.asarray(is_mode, dtype=np.float64).reshape(m, n, order="F").reshape(-1, order="F"))
    rc = L.orc_nlg_general_psi_setup(C.byref(s), _p(mode), C.c_int(max_iter), C.c_double(conv_tol), su)
    c = su.contents
    get = lambda ptr, shape: np.ctypeslib.as_array(ptr, (int(np.prod(shape)),)).copy().reshape(shape, order="F")
    out = dict(ok=rc == 0, iterations=c.iters, approx_loglik=c.approx_loglik, mode=get(c.mode, (m, n)), scales=get(c.scales, (n,)),
               alphahat=get(c.alphahat, (m, n + 1)), Vt=get(c.Vt, (m, m, n + 1)), Ct=get(c.Ct, (m, m, n + 1)),
               D=get(c.D, (n,)), Z=get(c.Z, (m, n)), T=get(c.T, (m, m, n)), C=get(c.C, (m, n)), H=get(c.H, (n,)),
               ekf_loglik=float(L.orc_nlg_general_ekf_loglik(C.byref(s))))
    if filter:
        rng, _ = make_rng(seed, stream)
        alpha = np.zeros((m, n + 1, nsim), order="F"); weights = np.zeros((nsim, n + 1), order="F")
        indices = np.zeros((nsim, n), dtype=np.uint32, order="F")
        ll = L.orc_nlg_general_psi_filter(C.byref(s), su, C.c_uint(nsim), C.byref(rng), _p(alpha), _p(weights), indices.ctypes.data_as(c_uint_p))
        out.update(logLik=float(ll), alpha=alpha, weights=weights, indices=indices)
    L.orc_nlg_setup_free(su)
    return out


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


def mlg_psi_filter(arr: dict, nsim: int, seed=1):
    """ssm_mlg::psi_filter (gaussian_psi_smoother case 0, src/R_psi.cpp:20-25): alpha m x (n+1) x N.  ``arr`` from
    bssm_amd.models.model_arrays of an ssm_mlg model (y n x p as R holds it; the reference transposes on entry)."""
    L = lib()
    n, m, p = arr["n"], arr["m"], arr["p"]
    mod = OrcMng()
    yt = np.asfortranarray(np.asarray(arr["y"], dtype=np.float64).T)  # p x n
    keep = [yt] + [np.asfortranarray(arr[k], dtype=np.float64) for k in ("Z", "T", "R", "a1", "P1", "D", "C", "H")]
    mod.n, mod.m, mod.k, mod.p = n, m, arr["k"], p
    mod.Ztv, mod.Ttv, mod.Rtv, mod.Dtv, mod.Ctv = arr["Ztv"], arr["Ttv"], arr["Rtv"], arr["Dtv"], arr["Ctv"]
    mod.y = _p(yt)
    for name, a in zip(("Z", "T", "R", "a1", "P1", "D", "C"), keep[1:8]):
        setattr(mod, name, _p(a))
    rng, _ = make_rng(seed, 0)
    alpha = np.zeros((m, n + 1, nsim), order="F")
    L.orc_mlg_psi_filter.restype = C.c_int
    ok = L.orc_mlg_psi_filter(C.byref(mod), _p(keep[8]), C.c_int(arr["Htv"]), C.c_uint(nsim), C.byref(rng), _p(alpha))
    if not ok:
        raise FloatingPointError("mlg smoother failed")
    return alpha


def ulg_psi_filter(arr: dict, nsim: int, seed=1):
    """ssm_ulg::psi_filter (gaussian_psi_smoother): alpha m x (n+1) x N"""
    g, keep = _pack_ulg(arr)
    rng, _ = make_rng(seed, 0)
    n, m = arr["n"], arr["m"]
    alpha = np.zeros((m, n + 1, nsim), order="F")
    lib().orc_ulg_psi_filter(C.byref(g), C.c_uint(nsim), C.byref(rng), _p(alpha))
    return alpha


# ------------------------------------------------------------------ chains (callers of the filters)
# Restatement of approx_mcmc::amcmc (src/approx_mcmc.cpp:110-232), mcmc::pm_mcmc (src/mcmc.cpp:337-491) and
# mcmc::da_mcmc (src/mcmc.cpp:530-685) in plain numpy over the oracle's filters.  ramcmc::adapt_S (CRAN ramcmc,
# Vihola 2012) is stated through its definition S S' <- S (I + eta (a - a*) u u' / |u|^2) S' with a fresh Cholesky
# factorisation, not through the rank-one update the library uses.
def builtin_update(arr: dict, kind: str, flags, theta):
    """bsm_ng / svm / ar1_ng ::update_model (src/model_bsm_ng.cpp:17-51, src/model_svm.cpp:12-29,
    src/model_ar1_ng.cpp:12-36) on a copy of the flattened model"""
    a = dict(arr)
    for k in ("T", "R", "P1", "a1", "C", "beta"):
        a[k] = np.array(arr[k], dtype=np.float64, order="F", copy=True)
    th = np.asarray(theta, dtype=np.float64)
    nx = a["xreg"].shape[1]
    if kind == "bsm_ng":
        slope, seasonal, noise, f0, f1, f2, phi_est = flags
        est = [f0 == 0, bool(slope) and f1 == 0, bool(seasonal) and f2 == 0]
        j = 0
        if f0 + f1 + f2 < 3 or noise:
            if est[0]:
                a["R"][0, 0, 0] = np.exp(th[j]); j += 1
            if est[1]:
                a["R"][1, 1, 0] = np.exp(th[j]); j += 1
            if est[2]:
                a["R"][1 + slope, 1 + slope, 0] = np.exp(th[j]); j += 1
            if noise:
                m = a["R"].shape[0]
                a["R"][m - 1, 1 + slope + seasonal, 0] = np.exp(th[j])
                a["P1"][m - 1, m - 1] = np.exp(th[j]) ** 2
                j += 1
        if phi_est:
            a["phi"] = float(np.exp(th[j]))
        if nx:
            a["beta"] = th[th.size - nx:].copy()
    elif kind == "svm":
        if flags[0] == 0:
            a["phi"] = float(th[2])
        else:
            a["a1"][0] = th[2]
            a["C"][...] = th[2] * (1 - th[0])
        a["T"][...] = th[0]
        a["R"][...] = th[1]
        a["P1"][...] = th[1] ** 2 / (1 - th[0] ** 2)
    elif kind == "ar1_ng":
        mu_est, phi_est = flags
        rho, sigma = th[0], np.exp(th[1])
        a["T"][...] = rho
        a["R"][...] = sigma
        if mu_est:
            a["a1"][0] = th[2]
            a["C"][...] = th[2] * (1 - rho)
        a["P1"][...] = sigma ** 2 / (1 - rho ** 2)
        if phi_est:
            a["phi"] = float(np.exp(th[2 + mu_est]))
        if nx:
            a["beta"] = th[th.size - nx:].copy()
    else:
        raise ValueError(kind)
    if nx:
        a["xbeta"] = a["xreg"] @ a["beta"]
    return a


def builtin_log_prior(kind: str, flags, n_xreg, theta, pd, pp):
    """log_prior_pdf: src/model_bsm_ng.cpp:90-136, src/model_svm.cpp:50-88, src/model_ar1_ng.cpp:63-118"""
    x = np.asarray(theta, dtype=np.float64)
    pars = x.copy()
    lp = 0.0
    if kind == "bsm_ng":
        if flags[3] + flags[4] + flags[5] < 3 or flags[2] or flags[6]:
            k = x.size - n_xreg
            pars[:k] = np.exp(x[:k])
            lp += x[:k].sum()
    elif kind == "ar1_ng":
        pars[1] = np.exp(x[1]); lp += x[1]
        if flags[1]:
            pars[2 + flags[0]] = np.exp(x[2 + flags[0]]); lp += x[2 + flags[0]]
    for i, v in enumerate(pars):
        q = pp[:, i]
        c = pd[i]
        if c == 0:
            if v < q[0] or v > q[1]:
                return -np.inf
        elif c == 1:
            if v < 0:
                return -np.inf
            lp -= 0.5 * (v / q[0]) ** 2
        elif c == 2:
            lp -= 0.5 * ((v - q[0]) / q[1]) ** 2
        elif c == 3:
            if v < q[2] or v > q[3]:
                return -np.inf
            lp -= 0.5 * ((v - q[0]) / q[1]) ** 2
        elif c == 4:
            if v < 0:
                return -np.inf
            lp += (q[0] - 1) * np.log(v) - q[1] * v
    return float(lp)


def global_approx_loglik(arr: dict, approx_y, approx_H):
    """approximate() + log_likelihood(nsim = 0) with local_approx = FALSE once an approximation exists
    (src/model_ssm_ung.cpp:104-116, 216-232): the pseudo-observations stay, one smoother pass against the new parameters
    gives the mode; log-likelihood = Gaussian part + constants + the mode-based scales (:241-290, 347-370).
    Univariate, Poisson / binomial / negative binomial / gamma."""
    ga = {**arr, "y": np.asarray(approx_y, dtype=np.float64), "H": np.asarray(approx_H, dtype=np.float64)}
    at, _, _ = ulg_smoother(ga)
    n = arr["n"]
    Z = np.asarray(arr["Z"], dtype=np.float64).reshape(arr["m"], -1, order="F")
    D = np.asarray(arr["D"], dtype=np.float64).reshape(-1)
    mode = np.array([arr["xbeta"][t] + D[t * arr["Dtv"]] + Z[:, t * arr["Ztv"]] @ at[:, t] for t in range(n)])
    gll = ulg_loglik(ga)
    y, u, phi, dist = np.asarray(arr["y"], dtype=float), np.asarray(arr["u"], dtype=float).reshape(-1), arr["phi"], arr["distribution"]
    sc = 0.0
    cterm = const_term(arr)
    for t in range(n):
        if not np.isfinite(y[t]):
            continue
        md = mode[t]
        if dist == 1:
            v = y[t] * md - u[t] * np.exp(md)
        elif dist == 2:
            v = y[t] * md - u[t] * np.log1p(np.exp(md))
        elif dist == 3:
            v = y[t] * md - (y[t] + phi) * np.log(phi + u[t] * np.exp(md))
        elif dist == 4:
            v = -phi * (md + y[t] * np.exp(-md) / u[t])
        else:
            raise ValueError("distribution")
        sc += v + 0.5 * ((ga["y"][t] - md) / ga["H"][t]) ** 2
        cterm -= -0.9189385332046727 - np.log(ga["H"][t])
    return gll + cterm + sc, mode


def chain_draws(seed, it, d):
    """the chain's counter-based draws: normals from blocks (iter_lo, iter_hi, j/2, purpose 5), uniforms from blocks
    (iter_lo, iter_hi, 2^31 + which, purpose 5) of the key (seed, 0)"""
    L = lib()
    r, _ = make_rng(seed=seed, stream=0)
    lo, hi = it & 0xFFFFFFFF, it >> 32
    u = np.zeros(d + 1)
    z0, z1 = C.c_double(), C.c_double()
    for j in range(0, d, 2):
        L.orc_rng_normal_pair(C.byref(r), 5, hi, lo, j // 2, C.byref(z0), C.byref(z1))
        u[j], u[j + 1] = z0.value, z1.value
    un = [L.orc_rng_uniform(C.byref(r), 5, hi, lo, 0x80000000 + w) for w in (0, 1)]
    return u[:d], un


def adapt_S(S, u, acc, target, i, gamma):
    d = u.size
    eta = min(1.0, d * i ** (-gamma))
    M = np.eye(d) + eta * (acc - target) * np.outer(u, u) / (u @ u)
    return np.linalg.cholesky(S @ M @ S.T)


def run_chain(update, log_prior, theta0, S0, iters, burnin, thin, gamma, target, seed, end_ram=False, kind="approx",
              nsim=0, method="psi", output_type=3, fn_model=False, local_approx=True, sde_levels=None):
    """One chain.  ``update(theta) -> arr`` gives the flattened model, ``log_prior(theta) -> float``.
    kind: 'approx' | 'pm' | 'da'.  Returns theta (n_stored x d), counts, posterior, S, acceptance_rate and by kind
    approx_loglik / prior / modes, alpha (n+1) x m x n_stored, alphahat / Vt."""
    theta = np.asarray(theta0, dtype=np.float64).copy()
    d = theta.size
    S = np.tril(np.asarray(S0, dtype=np.float64)).copy()
    filtered = kind != "approx"
    L = lib()
    fn_chain_seed = (seed ^ 0x5bd1e995) & 0xFFFFFFFF  # functor-model chains filter in a seed domain of their own

    def approx(arr, it=0):
        if sde_levels:  # first stage of sde_da_mcmc (src/mcmc.cpp:859,908): bootstrap filter at the coarse level, odd stream
            r = fn_bsf_filter({**arr, "L": sde_levels[0]}, nsim, seed=fn_chain_seed, stream=(2 * it + 1) & 0xFFFFFFFF)
            return (r["logLik"] if np.isfinite(r["logLik"]) else -np.inf), None, None
        if fn_model:  # ssm_nlg: mode-based approximation (src/model_ssm_nlg.cpp:258-277)
            su = fn_psi_setup(arr)
            return (su["approx_loglik"] if su["rc"] == 0 and su["ok"] else -np.inf), su["mode"].reshape(-1, order="F"), su
        if not local_approx and kept:  # the first approximation's pseudo-observations are kept
            ll, md = global_approx_loglik(arr, kept["y"], kept["H"])
            return ll, md, None
        su = psi_setup(arr)
        if not local_approx:
            kept.update(y=su["approx_y"], H=su["approx_H"])
        return su["approx_loglik"], su["mode"], su

    kept = {}

    def filt(arr, it):
        """(ll0, ll1, particle system) with the filter's stream = iteration (functor models: the chain's own seed domain,
        stream 2 * iteration; 2 * iteration + 1 is the coarse first stage of the two-level ssm_sde chain)"""
        if fn_model:
            sd, st = fn_chain_seed, (2 * it) & 0xFFFFFFFF
            if method == "psi":
                r = fn_psi_filter(arr, nsim, seed=sd, stream=st)
                return r["logLik"], r["setup"]["approx_loglik"], r
            if sde_levels:
                arr = {**arr, "L": sde_levels[1]}
            r = fn_ekpf_filter(arr, nsim, seed=sd, stream=st) if method == "ekf" else fn_bsf_filter(arr, nsim, seed=sd, stream=st)
            return r["logLik"], r["logLik"], r
        sd, st = seed + (it >> 32), it & 0xFFFFFFFF
        if method == "psi":
            r = psi_filter(arr, nsim, seed=sd, stream=st)
            return r["logLik"], r["setup"]["approx_loglik"], r
        r = bsf_filter(arr, nsim, seed=sd, stream=st)
        return r["logLik"], r["logLik"], r

    def sample_or_summarise(res, it):
        sm = filter_smoother(res["alpha"], res["indices"])  # m x (n+1) x N
        if output_type == 1:
            r, _ = make_rng(seed=fn_chain_seed, stream=(2 * it) & 0xFFFFFFFF) if fn_model else make_rng(seed=seed + (it >> 32), stream=it & 0xFFFFFFFF)
            pick = min(int(L.orc_rng_uniform(C.byref(r), 3, 0, 0, 0) * nsim), nsim - 1)
            return sm[:, :, pick].T.copy(), None, None
        if output_type == 2:
            mean, cov = summary(sm)
            return None, mean, cov
        return None, None, None

    arr = update(theta)
    logprior = log_prior(theta)
    assert np.isfinite(logprior)
    mode = None
    if filtered:
        ll0, ll1, res = filt(arr, 0)
        if sde_levels:
            ll1 = approx(arr, 0)[0]
        sampled, ah_i, vt_i = sample_or_summarise(res, 0)
    else:
        ll0, mode, _ = approx(arr)
        ll1 = ll0
    n, m = arr["n"], arr["m"]
    alphahat = np.zeros((m, n + 1)); Vt = np.zeros((m, m, n + 1)); valpha = np.zeros((m, m, n + 1))
    st_theta, st_counts, st_post, st_al, st_prior, st_modes, st_alpha = [], [], [], [], [], [], []
    new_value, n_values, accepted = True, 0, 0
    for i in range(1, iters + 1):
        u, (un0, un1) = chain_draws(seed, i, d)
        prop = theta + S @ u
        lp_prop = log_prior(prop)
        acc_prob = 0.0
        if lp_prop > -np.inf and not np.isnan(lp_prop):
            arr_p = update(prop)
            accept = False
            if kind == "approx":
                llp0, mode_p, _ = approx(arr_p)
                llp1 = llp0
                acc_prob = min(1.0, np.exp(llp0 - ll0 + lp_prop - logprior))
                accept = un0 < acc_prob
            elif kind == "pm":
                llp0, llp1, res_p = filt(arr_p, i)
                acc_prob = min(1.0, np.exp(llp1 - ll1 + lp_prop - logprior))
                accept = np.log(un0) < llp0 - ll0 + lp_prop - logprior
            else:
                llp1 = approx(arr_p, i)[0]
                acc_prob = min(1.0, np.exp(llp1 - ll1 + lp_prop - logprior))
                if un0 < acc_prob:
                    llp0, llp1_f, res_p = filt(arr_p, i)
                    if not sde_levels:
                        llp1 = llp1_f
                    accept = np.log(un1) < llp0 + ll1 - ll0 - llp1
            if accept:
                if i > burnin:
                    accepted += 1; n_values += 1
                if filtered:
                    sampled, ah_i, vt_i = sample_or_summarise(res_p, i)
                else:
                    mode = mode_p
                ll0, ll1, logprior, theta = llp0, llp1, lp_prop, prop
                new_value = True
        if i > burnin and filtered and output_type == 2:
            cnt = i - burnin
            diff = ah_i - alphahat
            alphahat = (alphahat * (cnt - 1) + ah_i) / cnt
            Vt = (Vt * (cnt - 1) + vt_i) / cnt
            for t in range(n + 1):
                valpha[:, :, t] += np.outer(diff[:, t], ah_i[:, t] - alphahat[:, t])
        if i > burnin and n_values % thin == 0:
            if new_value:
                st_theta.append(theta.copy()); st_counts.append(1); st_post.append(logprior + ll0)
                if not filtered:
                    st_al.append(ll0); st_prior.append(logprior); st_modes.append(np.array(mode).copy())
                elif output_type == 1:
                    st_alpha.append(sampled)
                new_value = False
            else:
                st_counts[-1] += 1
        if not end_ram or i <= burnin:
            S = adapt_S(S, u, acc_prob, target, i, gamma)
    out = dict(theta=np.array(st_theta), counts=np.array(st_counts, dtype=np.uint32), posterior=np.array(st_post), S=S,
               acceptance_rate=accepted / (iters - burnin))
    if not filtered:
        out.update(approx_loglik=np.array(st_al), prior=np.array(st_prior), modes=np.array(st_modes).T)
    elif output_type == 1:
        out["alpha"] = np.stack(st_alpha, axis=2)
    elif output_type == 2:
        out["alphahat"], out["Vt"] = alphahat, Vt + valpha / (iters - burnin)
    return out
