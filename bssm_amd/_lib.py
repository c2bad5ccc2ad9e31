"""ctypes binding of libbssm_amd.so (include/bssm_amd.h).  The library is the product: if it is
missing or no HIP device is usable the compute entry points fail loudly -- there is no fallback."""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbssm_amd.so")

BSSM_OK, BSSM_WARN_DEGENERATE = 0, 1
PLAN_KEEP_HISTORY, PLAN_FORCE_GRID, PLAN_FORCE_WG, PLAN_FACTORED_WEIGHTS = 1, 2, 4, 8

c_double_p = C.POINTER(C.c_double)
c_u32_p = C.POINTER(C.c_uint32)


class UngModel(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("n", "m", "k", "Ztv", "Ttv", "Rtv", "Dtv", "Ctv", "Htv", "distribution",
                                          "n_xreg", "max_iter", "local_approx")] + \
               [("conv_tol", C.c_double), ("phi", C.c_double)] + \
               [(k, c_double_p) for k in ("y", "Z", "T", "R", "a1", "P1", "D", "C", "xreg", "beta", "u", "H",
                                          "initial_mode")] + \
               [("p", C.c_int32), ("distribution_p", C.POINTER(C.c_int32)), ("phi_p", c_double_p)]


class FnModel(C.Structure):
    _fields_ = [("model_id", C.c_int32), ("n", C.c_int32), ("y", c_double_p), ("theta", c_double_p), ("n_theta", C.c_int32),
                ("known_params", c_double_p), ("n_known", C.c_int32), ("x0", C.c_double), ("positive", C.c_int32),
                ("L", C.c_int32), ("user_slot", C.c_int32), ("n_states", C.c_int32), ("H", C.c_double),
                ("R_diag", C.c_double * 2), ("a1", C.c_double * 2), ("P1", C.c_double * 4),
                ("use_ext", C.c_int32), ("n_etas", C.c_int32), ("R_ext", C.c_double * 256), ("a1_ext", C.c_double * 16),
                ("P1_ext", C.c_double * 256), ("known_tv_params", c_double_p), ("tv_rows", C.c_int32), ("tv_cols", C.c_int32)]


def pack_fn_model(model, L=None):
    s = FnModel()
    y = np.ascontiguousarray(model["y"], dtype=np.float64)
    th = np.ascontiguousarray(model["theta"], dtype=np.float64)
    kp = np.ascontiguousarray(model.get("known_params", np.zeros(0)), dtype=np.float64)
    s.model_id, s.n, s.y, s.theta, s.n_theta = int(model["model_id"]), y.size, dptr(y), dptr(th), th.size
    s.known_params, s.n_known = (dptr(kp) if kp.size else None), kp.size
    s.x0, s.positive = float(model.get("x0", 0.0)), int(bool(model.get("positive", False)))
    s.L = int(model.get("L", 0) if L is None else L)
    if s.model_id >= 7:  # user-defined: compiled at run time, numbers for the state-independent parts
        s.user_slot = user_model_slot(model["user_source"], int(model.get("n_states", 1)), jit=bool(model.get("user_jit", False)))
        s.n_states = int(model.get("n_states", 1))
        if s.model_id == 7:
            fill_user_numbers(s, model, th)
    keep = [y, th, kp]
    tv = model.get("known_tv_params")
    if tv is not None and np.size(tv):  # the reference's known_tv_params matrix (snippets with BSSM_USER_TV read it)
        tv = np.asfortranarray(np.atleast_2d(np.asarray(tv, dtype=np.float64)))
        s.known_tv_params, s.tv_rows, s.tv_cols = dptr(tv), tv.shape[0], tv.shape[1]
        keep.append(tv)
    return s, keep


def fill_user_numbers(s, model, theta):
    """H, diag(R), a1, P1 of a user-defined ssm_nlg model: numbers, or functions of theta (the reference's H_fn / R_fn /
    a1_fn / P1_fn take theta too, src/model_ssm_nlg.h:10-21) evaluated at ``theta``"""
    m = int(s.n_states)
    val = lambda k: model[k](np.array(theta, dtype=float)) if callable(model[k]) else model[k]
    s.H = float(val("H"))
    a1 = np.asarray(val("a1"), dtype=float).reshape(-1)[:m]
    P1 = np.asarray(val("P1"), dtype=float).reshape(m, m)
    if model.get("R") is not None or m > 2:  # any m x k matrix R, up to 8 states (bootstrap paths; psi-APF / EKPF up to 4)
        k = int(model.get("n_etas", m))
        R = np.asarray(val("R"), dtype=float).reshape(m, k) if model.get("R") is not None else np.diag(np.asarray(val("R_diag"), dtype=float).reshape(-1)[:m])
        s.use_ext, s.n_etas = 1, R.shape[1]
        for j, v in enumerate(R.reshape(-1, order="F")): s.R_ext[j] = v
        for j, v in enumerate(a1): s.a1_ext[j] = v
        for j, v in enumerate(P1.reshape(-1, order="F")): s.P1_ext[j] = v
        return
    for j, v in enumerate(np.asarray(val("R_diag"), dtype=float).reshape(-1)[:m]): s.R_diag[j] = v
    for j, v in enumerate(a1): s.a1[j] = v
    for j, v in enumerate(P1.reshape(-1, order="F")): s.P1[j] = v


_user_slots = {}


def user_model_slot(source: str, n_states: int, jit: bool = False) -> int:
    """Compile (cached by content hash) and load a user-defined model's device functions; returns the registry slot.
    jit: in process through hiprtc (bssm_user_model_jit: bootstrap filter only) instead of a hipcc-built shared library."""
    key = (source, n_states, bool(jit))
    if key not in _user_slots:
        L = lib()
        if jit:
            slot = L.bssm_user_model_jit(source.encode(), C.c_int32(n_states))
            if slot < 1:
                raise BssmAmdError(f"bssm_amd error {slot}: {L.bssm_amd_last_error().decode()}")
            _user_slots[key] = slot
            return slot
        buf = C.create_string_buffer(4096)
        check(L.bssm_user_model_build(source.encode(), C.c_int32(n_states), buf, C.c_int32(4096)))
        slot = L.bssm_user_model_load(buf.value)
        if slot < 1:
            raise BssmAmdError(f"bssm_amd error {slot}: {L.bssm_amd_last_error().decode()}")
        _user_slots[key] = slot
    return _user_slots[key]


class FilterOut(C.Structure):
    _fields_ = [(k, c_double_p) for k in ("at", "att", "Pt", "Ptt", "alphahat", "Vt", "weights", "alpha")] + \
               [("indices", c_u32_p), ("logLik", C.c_double)]


class IsJob(C.Structure):
    _fields_ = [("n_stored", C.c_int32), ("first", C.c_int32), ("count", C.c_int32),
                ("models", C.POINTER(UngModel)), ("modes", c_double_p), ("approx_loglik", c_double_p),
                ("counts", c_u32_p), ("sampling_method", C.c_int32), ("is_type", C.c_int32),
                ("output_type", C.c_int32), ("nsim", C.c_uint32), ("seed", C.c_uint32), ("n_threads", C.c_int32)]


class BssmAmdError(RuntimeError):
    pass


_lib = None


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64 (torch/lib), and libbssm_amd.so
    names the runtime by its SONAME only: loaded after `import torch` it binds to torch's copy -- one runtime, torch streams are
    valid handles for the `hip_stream` arguments, RCCL comes from the same stack.  Loaded BEFORE torch it binds to the system
    ROCm's runtime, and a later `import torch` brings a second HSA runtime into the process that finds no device
    ("No HIP GPUs are available"; scripts/runtime_order_probe.py shows both orders on the GPU box).  So where torch is
    installed but not imported yet, its runtime is loaded here first (a dlopen of one file -- torch itself is not imported) and
    the order stops mattering.  BSSM_AMD_SYSTEM_HIP=1 keeps the system runtime (a process that will never import torch)."""
    if os.environ.get("BSSM_AMD_SYSTEM_HIP") == "1" or "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    trace = os.environ.get("BSSM_AMD_TRACE", "0") not in ("", "0")
    if not os.path.exists(path):
        return
    # the preload only helps if torch's copy answers to the SONAME this library asks for (libamdhip64.so.<major>): with another
    # major version libbssm_amd.so would still bind to the system copy and TWO runtimes would share the process -- skip it then
    want = [nm for nm in _elf_dynamic(LIB_PATH)[1] if nm.startswith("libamdhip64.so")]
    have = _elf_dynamic(path)[0]
    if want and have and have not in want:
        if trace:
            print(f"[bssm_amd] torch ships {have}, the library needs {want[0]}: torch's HIP runtime NOT preloaded "
                  "(import torch first, or set BSSM_AMD_SYSTEM_HIP=1)", file=sys.stderr)
        return
    try:
        C.CDLL(path, mode=C.RTLD_GLOBAL)
        if trace:
            print(f"[bssm_amd] preloaded torch's HIP runtime {path} (SONAME {have or 'unknown'})", file=sys.stderr)
    except OSError as e:
        if trace:
            print(f"[bssm_amd] torch's HIP runtime could not be preloaded ({e}): the system runtime is used", file=sys.stderr)


def _elf_dynamic(path):
    """(DT_SONAME or None, [DT_NEEDED ...]) of a 64-bit little-endian ELF shared object; (None, []) when it cannot be read"""
    import struct
    try:
        with open(path, "rb") as f:
            data = f.read()
        if data[:4] != b"\x7fELF" or data[4] != 2 or data[5] != 1:
            return None, []
        shoff, = struct.unpack_from("<Q", data, 0x28)
        shentsize, shnum = struct.unpack_from("<HH", data, 0x3A)
        dyn = strtab_idx = None
        secs = []
        for i in range(shnum):
            name, typ, flags, addr, off, size, link, info, align, entsize = struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize)
            secs.append((typ, off, size, link))
            if typ == 6:  # SHT_DYNAMIC
                dyn, strtab_idx = (off, size), link
        if dyn is None:
            return None, []
        _, soff, ssize, _ = secs[strtab_idx]
        strtab = data[soff:soff + ssize]
        cstr = lambda o: strtab[o:strtab.index(b"\0", o)].decode()
        soname, needed = None, []
        for o in range(dyn[0], dyn[0] + dyn[1], 16):
            tag, val = struct.unpack_from("<qQ", data, o)
            if tag == 0:
                break
            if tag == 1:
                needed.append(cstr(val))
            elif tag == 14:
                soname = cstr(val)
        return soname, needed
    except Exception:  # noqa: BLE001
        return None, []


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BssmAmdError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(bssm_amd has no CPU fallback)")
        _share_torch_hip_runtime()
        L = C.CDLL(LIB_PATH)
        L.bssm_amd_last_error.restype = C.c_char_p
        L.bssm_amd_version.restype = C.c_char_p
        L.bssm_plan_create.restype = C.c_void_p
        L.bssm_psi_smoother_nlg.argtypes = [C.POINTER(FnModel), C.c_uint32, C.c_uint32, C.c_int32, C.c_double, C.POINTER(FilterOut)]
        L.bssm_fn_psi_setup.argtypes = [C.POINTER(FnModel), c_double_p, C.c_int32, C.c_double] + [c_double_p] * 2 + [C.POINTER(C.c_double)] + [c_double_p] * 7
        L.bssm_plan_create.argtypes = [C.POINTER(UngModel), C.c_int32, C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.bssm_plan_destroy.argtypes = [C.c_void_p]
        L.bssm_plan_run.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.bssm_plan_loglik.argtypes = [C.c_void_p, C.c_void_p, c_double_p]
        L.bssm_plan_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                     C.POINTER(C.c_int32), c_double_p]
        L.bssm_plan_device_outputs.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        L.bssm_plan_time.argtypes = [C.c_void_p, C.c_uint32, C.c_int32, C.c_void_p, c_double_p]
        L.bssm_plan_time_kernel.argtypes = [C.c_void_p, C.c_uint32, C.c_int32, C.c_void_p, c_double_p, C.POINTER(C.c_int32)]
        _lib = L
    # the library reads its BSSM_AMD_* switches once, at load (csrc/options.h); a process that changes one afterwards (the
    # parity tests and tuning scripts do, between calls) has it re-read them
    global _env_seen
    env = tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("BSSM_AMD_")))
    if env != _env_seen:
        if _env_seen is not None or env:
            _lib.bssm_amd_reload_env()
        _env_seen = env
    return _lib


_env_seen = None


def check(rc: int) -> int:
    if rc < 0:
        raise BssmAmdError(f"bssm_amd error {rc}: {lib().bssm_amd_last_error().decode()}")
    return rc


def dptr(a):
    return None if a is None else a.ctypes.data_as(c_double_p)


_PTR_FIELDS = ("y", "Z", "T", "R", "a1", "P1", "D", "C", "xreg", "beta", "u", "H", "initial_mode")


class DrawList(list):
    """per-draw model dicts (post_correct.expand_models) that may carry the batch arrays they are views of: ``batch[k]`` is a C-contiguous (count x size)
    array whose row i is draw i's component k, ``phi`` the draws' dispersion parameters.  Replacing, adding or removing an entry drops the batch (the
    list is then packed draw by draw); the dicts themselves are views of the batch and are not to be given other arrays in place -- copy and replace."""
    batch = None
    phi = None

    def _drop(self):
        self.batch = None
        self.phi = None


def _dropping(name):
    def method(self, *a, **kw):
        self._drop()
        return getattr(list, name)(self, *a, **kw)
    method.__name__ = name
    return method


for _name in ("__setitem__", "__delitem__", "__iadd__", "__imul__", "append", "extend", "insert", "pop", "remove", "sort", "reverse", "clear"):
    setattr(DrawList, _name, _dropping(_name))


def pack_models(arrs):
    """One UngModel per dict of ``arrs`` (models.model_arrays / post_correct.expand_models) as a ctypes array: the first draw is
    packed field by field, the others are byte copies of it whose pointer fields are patched only where the draw holds a DIFFERENT
    array object (expand_models shares what update_fn did not replace) -- packing 2 048 draws one by one cost 50 ms of Python
    around a 10 ms library call.  Returns (array, keepalive)."""
    cnt = len(arrs)
    s0, keep0 = pack_model(arrs[0])
    models = (UngModel * cnt)()
    sz = C.sizeof(UngModel)
    raw = np.frombuffer(models, dtype=np.uint8).reshape(cnt, sz)
    raw[:] = np.frombuffer(bytes(s0), dtype=np.uint8)
    keep = [keep0]
    base = arrs[0]
    ints = ("n", "m", "k", "Ztv", "Ttv", "Rtv", "Dtv", "Ctv", "Htv", "distribution", "n_xreg", "max_iter", "local_approx")
    scal = ints + ("conv_tol", "p")
    base_scal = tuple(map(base.get, scal))
    base_ptrs = tuple(map(base.get, _PTR_FIELDS))
    batch = getattr(arrs, "batch", None)
    if batch is not None and all(v.flags["C_CONTIGUOUS"] and v.dtype == np.float64 and v.shape[0] == cnt for v in batch.values()):
        # built-in models expanded in one library call: every draw shares the scalars and the untouched arrays of draw 0, and its own components are
        # rows of the batch arrays -- the pointer fields of all draws are patched at once (2 048 draws: 18 ms of per-draw Python -> well under 1 ms)
        for f, v in batch.items():
            off = getattr(UngModel, f).offset
            raw[:, off:off + 8].view(np.uint64)[:, 0] = v.ctypes.data + np.arange(cnt, dtype=np.uint64) * np.uint64(v.strides[0])
            keep.append(v)
        off = UngModel.phi.offset
        raw[:, off:off + 8].view(np.float64)[:, 0] = np.asarray(arrs.phi, dtype=np.float64)
        return models, keep
    for i in range(1, cnt):
        a = arrs[i]
        if tuple(map(a.get, scal)) != base_scal or a.get("distribution_p") is not base.get("distribution_p") or a.get("phi_p") is not base.get("phi_p"):
            si, ki = pack_model(a)  # (a draw that differs in more than its arrays: the slow way)
            raw[i] = np.frombuffer(bytes(si), dtype=np.uint8)
            keep.append(ki)
            continue
        for f, v, b0 in zip(_PTR_FIELDS, map(a.get, _PTR_FIELDS), base_ptrs):
            if v is b0:
                continue
            if v is not None and v.size == 0:
                v = None
            addr = 0
            if v is not None:
                v = np.asfortranarray(v, dtype=np.float64)
                keep.append(v)
                addr = v.ctypes.data
            off = getattr(UngModel, f).offset
            raw[i, off:off + 8].view(np.uint64)[0] = addr
    off = UngModel.phi.offset
    raw[:, off:off + 8].view(np.float64)[:, 0] = np.asarray([float(a["phi"]) for a in arrs])
    return models, keep


def pack_model(arr: dict):
    """dict from models.model_arrays -> (UngModel, keepalive list)"""
    s = UngModel()
    for k in ("n", "m", "k", "Ztv", "Ttv", "Rtv", "Dtv", "Ctv", "Htv", "distribution", "n_xreg", "max_iter",
              "local_approx"):
        setattr(s, k, int(arr[k]))
    s.conv_tol = float(arr["conv_tol"])
    s.phi = float(arr["phi"])
    keep = []
    for k in ("y", "Z", "T", "R", "a1", "P1", "D", "C", "xreg", "beta", "u", "H", "initial_mode"):
        a = arr.get(k)
        if a is not None and a.size == 0:
            a = None
        if a is not None:
            a = np.asfortranarray(a, dtype=np.float64)
            keep.append(a)
        setattr(s, k, dptr(a))
    s.p = int(arr.get("p", 1))
    if s.p > 1:
        dp = np.ascontiguousarray(arr["distribution_p"], dtype=np.int32)
        pp = np.ascontiguousarray(arr["phi_p"], dtype=np.float64)
        keep += [dp, pp]
        s.distribution_p = dp.ctypes.data_as(C.POINTER(C.c_int32))
        s.phi_p = dptr(pp)
    return s, keep
