"""Python side of the mock R runtime (shim/mockr/): builds the SEXP arguments a `.Call` receives, runs the entry points that
shim/bssm_amd_shim.c registers, converts what comes back.  TEST HARNESS ONLY (see shim/mockr/Rinternals.h)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK_DIR = os.path.join(ROOT, "shim", "mockr")
LIB_PATH = os.path.join(MOCK_DIR, "libbssm_shim_mock.so")

NILSXP, CLOSXP, CHARSXP, LGLSXP, INTSXP, REALSXP, STRSXP, VECSXP = 0, 3, 9, 10, 13, 14, 16, 19
NA_INT = -2 ** 31
CLOSURE_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_void_p)


class Sx(int):
    """a SEXP handle (address of a mock object)"""


class RError(RuntimeError):
    """what would be an R error: Rf_error from the shim, a stack imbalance, a use of an unprotected object, ..."""


def build(force: bool = False) -> str:
    srcs = [os.path.join(ROOT, "shim", "bssm_amd_shim.c")] + [os.path.join(MOCK_DIR, f) for f in ("mockr.c", "Rinternals.h")]
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", MOCK_DIR, "-B"], stdout=subprocess.DEVNULL)
    return LIB_PATH


class MockR:
    def __init__(self):
        build()
        # the product library first, by its path (the mock library names it as a dependency with an $ORIGIN rpath)
        from bssm_amd import _lib
        _lib.lib()
        L = C.CDLL(LIB_PATH)
        for name, res, args in (
                ("mockr_vec", C.c_void_p, [C.c_int, C.c_long]), ("mockr_char", C.c_void_p, [C.c_char_p]), ("mockr_nil", C.c_void_p, []),
                ("mockr_data", C.c_void_p, [C.c_void_p]), ("mockr_type", C.c_int, [C.c_void_p]), ("mockr_len", C.c_long, [C.c_void_p]),
                ("mockr_attr", C.c_void_p, [C.c_void_p, C.c_char_p]), ("mockr_set_attr", None, [C.c_void_p, C.c_char_p, C.c_void_p]),
                ("mockr_set_elt", None, [C.c_void_p, C.c_long, C.c_void_p]), ("mockr_get_elt", C.c_void_p, [C.c_void_p, C.c_long]),
                ("mockr_closure", C.c_void_p, [CLOSURE_FN, C.c_void_p]), ("mockr_extptr", C.c_void_p, []),
                ("mockr_call", C.c_void_p, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]), ("mockr_last_error", C.c_char_p, []),
                ("mockr_last_warning", C.c_char_p, []), ("mockr_protect_depth", C.c_int, []), ("mockr_poisoned_touches", C.c_int, []),
                ("mockr_live_objects", C.c_long, []), ("mockr_gctorture", None, [C.c_int]), ("mockr_reset", None, []),
                ("mockr_lookup", C.c_void_p, [C.c_char_p, C.POINTER(C.c_int)]), ("mockr_registered_count", C.c_int, []),
                ("mockr_dynamic_symbols", C.c_int, []), ("mockr_building", None, [C.c_int]), ("mockr_release", None, [C.c_void_p]),
                ("mockr_closure_failed", None, [C.c_char_p]), ("bssm_amd_register", None, [C.c_void_p])):
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        self.L = L
        self._keep = []          # ctypes callbacks must outlive the SEXPs that point at them
        L.bssm_amd_register(None)   # what R_init_bssm does: R_registerRoutines + R_useDynamicSymbols(FALSE)
        L.mockr_building(1)      # everything built from here on is the caller's (arguments), until a closure says otherwise

    # ---- building arguments
    def null(self):
        return Sx(self.L.mockr_nil())

    def _vec(self, typ, arr, ctype, dim=None):
        arr = np.asarray(arr)
        flat = np.asfortranarray(arr).reshape(-1, order="F")
        s = self.L.mockr_vec(typ, flat.size)
        if flat.size:
            C.memmove(self.L.mockr_data(s), np.ascontiguousarray(flat, dtype=ctype).ctypes.data, flat.size * np.dtype(ctype).itemsize)
        shape = arr.shape if dim is None else tuple(dim)
        if len(shape) >= 2:
            d = self.L.mockr_vec(INTSXP, len(shape))
            C.memmove(self.L.mockr_data(d), np.asarray(shape, dtype=np.int32).ctypes.data, 4 * len(shape))
            self.L.mockr_set_attr(s, b"dim", d)
        return Sx(s)

    def real(self, x, dim=None):
        return self._vec(REALSXP, np.asarray(x, dtype=np.float64), np.float64, dim)

    def integer(self, x, dim=None):
        return self._vec(INTSXP, np.asarray(x, dtype=np.int32), np.int32, dim)

    def logical(self, x):
        return self._vec(LGLSXP, np.asarray(x, dtype=bool).astype(np.int32), np.int32)

    def string(self, x):
        xs = [x] if isinstance(x, str) else list(x)
        s = self.L.mockr_vec(STRSXP, len(xs))
        for i, v in enumerate(xs):
            self.L.mockr_set_elt(s, i, self.L.mockr_char(v.encode()))
        return Sx(s)

    def rlist(self, fields: dict):
        s = self.L.mockr_vec(VECSXP, len(fields))
        for i, (k, v) in enumerate(fields.items()):
            self.L.mockr_set_elt(s, i, self.value(v))
        self.L.mockr_set_attr(s, b"names", self.string(list(fields)))
        return Sx(s)

    def closure(self, fn):
        """an R function of one numeric argument returning a number or a named list (update_fn / prior_fn)"""
        def tramp(arg, user):
            try:
                res = fn(np.array(self.to_py(arg), dtype=float).reshape(-1))
                self.L.mockr_building(1)        # assembled as the harness's (no collection while it is being built) ...
                try:
                    out = self.value(res)
                finally:
                    self.L.mockr_building(0)
                self.L.mockr_release(out)       # ... then handed back unprotected, as an R function's value is
                return out
            except Exception as e:  # noqa: BLE001 -- becomes the R error of the evaluation
                self.L.mockr_closure_failed(("%s: %s" % (type(e).__name__, e)).encode()[:1000])
                return None
        cb = CLOSURE_FN(tramp)
        self._keep.append(cb)
        return Sx(self.L.mockr_closure(cb, None))

    def extptr(self, **attrs):
        s = self.L.mockr_extptr()
        for k, v in attrs.items():
            self.L.mockr_set_attr(s, k.encode(), self.value(v))
        return Sx(s)

    def value(self, v):
        """Python value -> SEXP by R's own typing rules: float / float array numeric, int32 array integer, bool logical, str
        character, dict list, callable function, None NULL"""
        if v is None:
            return self.null()
        if isinstance(v, Sx):
            return v
        if isinstance(v, dict):
            return self.rlist(v)
        if callable(v):
            return self.closure(v)
        if isinstance(v, str) or (isinstance(v, (list, tuple)) and v and all(isinstance(q, str) for q in v)):
            return self.string(v)
        a = np.asarray(v)
        if a.dtype == bool:
            return self.logical(a)
        if np.issubdtype(a.dtype, np.integer) and a.dtype.itemsize <= 4:
            return self.integer(a)
        return self.real(a.astype(np.float64))

    # ---- reading results
    def to_py(self, s):
        L = self.L
        t, n = L.mockr_type(s), L.mockr_len(s)
        if t == NILSXP:
            return None
        if t in (REALSXP, INTSXP, LGLSXP):
            ct = np.float64 if t == REALSXP else np.int32
            a = np.frombuffer(C.string_at(L.mockr_data(s), n * np.dtype(ct).itemsize), dtype=ct).copy() if n else np.zeros(0, dtype=ct)
            d = L.mockr_attr(s, b"dim")
            if L.mockr_type(d) != NILSXP:
                a = a.reshape(tuple(int(q) for q in self.to_py(d)), order="F")
            return a.astype(bool) if t == LGLSXP else a
        if t == STRSXP:
            return [C.string_at(L.mockr_data(L.mockr_get_elt(s, i))).decode() for i in range(n)]
        if t == VECSXP:
            names = L.mockr_attr(s, b"names")
            vals = [self.to_py(L.mockr_get_elt(s, i)) for i in range(n)]
            return dict(zip(self.to_py(names), vals)) if L.mockr_type(names) != NILSXP else vals
        raise TypeError("to_py: SEXP type %d" % t)

    # ---- .Call
    def registered(self, name):
        na = C.c_int()
        fn = self.L.mockr_lookup(name.encode(), C.byref(na))
        return (fn, na.value) if fn else (None, -1)

    def call(self, name, *args, convert=True):
        """.Call(name, ...) through the table the shim registered.  Raises RError on Rf_error / stack imbalance / use of an
        unprotected object; checks the argument count against the registration like R does."""
        fn, na = self.registered(name)
        if not fn:
            raise RError('"%s" not available for .Call() for package "bssm"' % name)
        if na != len(args):
            raise RError("Incorrect number of arguments (%d), expecting %d for '%s'" % (len(args), na, name))
        sx = (C.c_void_p * max(1, len(args)))(*[self.value(a) for a in args])
        self.L.mockr_building(0)                       # inside the call, allocations are the callee's
        try:
            res = self.L.mockr_call(fn, len(args), sx)
        finally:
            self.L.mockr_building(1)
        if not res:
            raise RError(self.L.mockr_last_error().decode())
        assert self.L.mockr_protect_depth() == 0
        return self.to_py(res) if convert else res

    def last_warning(self):
        return self.L.mockr_last_warning().decode()

    def reset(self):
        self.L.mockr_reset()
        self._keep.clear()


def r_model_list(model, distribution_code=True):
    """The named list an R caller hands to the `.Call` stubs for a model of the ssm_ung / ssm_mng / bsm_ng / svm / ar1_ng (and
    gaussian) classes, field for field as the reference's constructors build it -- /root/reference/R/models.R:351-357 (ssm_ung),
    :1103-1113 (bsm_ng), :1226-1237 (svm), :1393-1399 (ssm_mng) -- with R's storage types (numeric unless the constructor says
    as.integer / logical) and, as every caller does before the call (R/bootstrap_filter.R:57-60, R/post_correction.R:113-116),
    `distribution` replaced by pmatch(...) - 1."""
    from bssm_amd import models as M
    f = {}
    dist_codes = {d: i for i, d in enumerate(M.DISTRIBUTIONS)}
    for k, v in model.items():
        if k in ("priors", "theta_names", "prior_kinds"):
            continue
        if k == "distribution":
            if isinstance(v, str):
                f[k] = float(dist_codes[v]) if distribution_code else v       # pmatch() returns integer; - 1 makes it numeric
            else:
                f[k] = np.asarray([dist_codes[d] for d in v], dtype=np.float64)
            continue
        if k == "fixed":
            f[k] = np.asarray(v, dtype=np.int32)                               # as.integer(!notfixed)
            continue
        if isinstance(v, (bool, np.bool_)) or k in ("slope", "seasonal", "noise", "phi_est", "mu_est", "local_approx"):
            f[k] = bool(v)
            continue
        if callable(v):
            f[k] = v
            continue
        if v is None:
            continue
        f[k] = np.asarray(v, dtype=np.float64) if not isinstance(v, str) else v
    if model.get("priors"):
        pd, pp = M.prior_tables(model["priors"])
        f["prior_distributions"] = np.asarray(pd, dtype=np.float64)            # numeric in R (vapply(..., FUN.VALUE = 1)-style tables)
        f["prior_parameters"] = np.asarray(pp, dtype=np.float64)
    for k, dv in (("max_iter", 100.0), ("conv_tol", 1e-8), ("local_approx", True)):
        if not model.inherits("lineargaussian"):
            f.setdefault(k, dv)
    if "xreg" not in f and not model.inherits("ssm_mng") and not model.inherits("lineargaussian"):
        f["xreg"] = np.zeros((0, 0)); f["beta"] = np.zeros(0)                  # matrix(0, 0, 0), numeric(0): R/models.R:356
    return f
