"""IS post-correction of an approximate MCMC run: Python mirror of bssm's ``post_correct``
(R/post_correction.R:245-308 -> src/R_postcorrection.cpp:150-248 -> approx_mcmc::is_correction_psi,
src/approx_mcmc.cpp:353-427).

One independent psi-APF per retained theta-draw.  The draws are the unit that shards: with
``torch.distributed`` initialised (one process per GPU, backend "nccl" = RCCL over xGMI) every rank
filters a contiguous block of draws on its own GPU through the C ABI and ONE collective phase at the
end all-gathers the per-draw results (weights, and the smoothed summaries / trajectories when they are
requested).  Random streams are keyed by the global draw index, and the summary merge runs in draw
order on every rank, so the output is bit-identical for any number of GPUs.

As in the reference, ``theta`` is handed to the model's ``update_model`` exactly as stored in
``mcmc_output`` (R/post_correction.R:266-270).
"""
from __future__ import annotations

import ctypes as C
import time

import numpy as np

from . import _lib
from .filters import check_intmax, _default_seed, _INT_MAX
from .models import BssmModel, model_arrays, model_type

_OUTPUT_TYPES = {"full": 1, "summary": 2, "theta": 3, 1: 1, 2: 2, 3: 3}


# ------------------------------------------------------------------ per-draw models (parset, src/parset_ng.cpp:5-111)
def _builtin_flags(model: BssmModel):
    cls = model.classes[0]
    if cls == "bsm_ng":  # src/model_bsm_ng.cpp:4-14
        fx = list(model["fixed"])
        return [int(model["slope"]), int(model["seasonal"]), int(model["noise"]), int(fx[0]), int(fx[1]), int(fx[2]),
                int(model["phi_est"])]
    if cls == "svm":
        return [int(model["svm_type"])]
    if cls == "ar1_ng":
        return [int(model["mu_est"]), int(model["phi_est"])]
    return None


def expand_models(model: BssmModel, theta: np.ndarray):
    """One flattened model per row of ``theta`` (n_stored x n_par).

    Built-in model classes use the library's update_model maps (bssm_update_model_builtin);
    ``ssm_ung`` calls ``model['update_fn'](theta_i)`` which returns a dict of replaced components
    (Z, T, R, a1, P1, D, C, phi, beta), as the R closure does (src/model_ssm_ung.cpp:41-78)."""
    theta = np.atleast_2d(np.asarray(theta, dtype=np.float64))
    base = model_arrays(model)
    mt = model_type(model)
    flags = _builtin_flags(model)
    out = []
    L = _lib.lib()
    if flags is not None and theta.shape[0] > 1:
        # built-in maps, every draw in ONE library call (the reference's parset_ung, src/parset_ng.cpp:5-72); the per-draw dicts
        # then hold views into the batch arrays
        cnt = theta.shape[0]
        keys = ("T", "R", "P1", "a1", "C", "beta")
        bf = {k: np.asfortranarray(base[k], dtype=np.float64) for k in keys}
        big = {k: np.empty((cnt, bf[k].size)) for k in keys}
        phi = np.empty(cnt)
        s0, keep0 = _lib.pack_model({**base, **bf})
        fl = (C.c_int32 * len(flags))(*flags)
        thc = np.ascontiguousarray(theta, dtype=np.float64)  # row i = draw i = column i of the n_par x count matrix
        _lib.check(L.bssm_update_model_builtin_batch(C.c_int(mt), fl, _lib.dptr(thc), C.c_int32(thc.shape[1]), C.c_int32(cnt), C.byref(s0),
                                                     _lib.dptr(big["T"]), _lib.dptr(big["R"]), _lib.dptr(big["P1"]), _lib.dptr(big["a1"]),
                                                     _lib.dptr(big["C"]), _lib.dptr(big["beta"]) if bf["beta"].size else None, _lib.dptr(phi)))
        has_x = bool(base["xreg"].shape[1])
        for i in range(cnt):
            arr = dict(base)
            for k in keys:
                arr[k] = big[k][i].reshape(bf[k].shape, order="F")
            arr["phi"] = float(phi[i])
            if has_x:
                arr["xbeta"] = arr["xreg"] @ arr["beta"]
            out.append(arr)
        # (the batch arrays ride along: _lib.pack_models patches the pointer fields of all draws at once -- row i of big[k] IS draw i's array)
        out = _lib.DrawList(out)
        out.batch = {k: big[k] for k in keys if bf[k].size}
        out.phi = phi
        return out
    for i in range(theta.shape[0]):
        arr = dict(base)
        th = np.ascontiguousarray(theta[i])
        if flags is not None:
            for k in ("T", "R", "P1", "a1", "C", "beta"):
                arr[k] = np.array(base[k], dtype=np.float64, order="F", copy=True)
            s, keep = _lib.pack_model(arr)
            fl = (C.c_int32 * len(flags))(*flags)
            _lib.check(L.bssm_update_model_builtin(mt, fl, _lib.dptr(th), th.size, C.byref(s), _lib.dptr(arr["T"]),
                                                   _lib.dptr(arr["R"]), _lib.dptr(arr["P1"]), _lib.dptr(arr["a1"]),
                                                   _lib.dptr(arr["C"]), _lib.dptr(arr["beta"]) if arr["beta"].size else None))
            arr["phi"] = float(s.phi)
        else:
            fn = model.get("update_fn")
            if fn is not None:
                upd = fn(th) or {}
                for k, v in upd.items():
                    if k == "phi":
                        arr["phi"] = float(v)
                    elif k in ("Z", "T", "R", "a1", "P1", "D", "C", "beta"):
                        arr[k] = np.asfortranarray(np.asarray(v, dtype=np.float64).reshape(np.shape(base[k])))
                    else:
                        raise ValueError(f"update_fn returned unknown component '{k}'")
        if arr["xreg"].shape[1]:
            arr["xbeta"] = arr["xreg"] @ arr["beta"]
        out.append(arr)
    return out


# ------------------------------------------------------------------ sharding (static blocks, like `omp for schedule(static)`)
def shard_bounds(work: np.ndarray, world: int):
    """Contiguous blocks of draws with (nearly) equal total work; returns world+1 boundaries.
    is1 multiplies the particle count by count_i (src/approx_mcmc.cpp:376-378), so work_i = count_i there."""
    work = np.asarray(work, dtype=np.float64)
    n = work.size
    cum = np.concatenate([[0.0], np.cumsum(work)])
    bounds = [0]
    for r in range(1, world):
        target = cum[-1] * r / world
        k = int(np.searchsorted(cum, target, side="left"))
        bounds.append(min(max(k, bounds[-1]), n))
    bounds.append(n)
    return np.asarray(bounds, dtype=np.int64)


def _process_group(group):
    """(rank, world, torch.distributed or None).  A process that never imported torch.distributed cannot have a process
    group, and importing torch here would pull a second ROCm stack into callers that only use this library."""
    import sys
    _dist = sys.modules.get("torch.distributed")
    if _dist is not None and _dist.is_available() and _dist.is_initialized():
        return _dist.get_rank(group), _dist.get_world_size(group), _dist
    return 0, 1, None


def _last_timing(devices=None):
    t = (C.c_double * 2)()
    _lib.lib().bssm_amd_last_is_timing(t)
    out = dict(host_setup=t[0], device=t[1])
    t4 = (C.c_double * 4)()
    _lib.lib().bssm_amd_last_is_timing4(t4)
    out.update(host_pack=t4[0], setup_call=t4[1], filter=t4[2], wall=t4[3])
    if devices is not None:  # per-shard split of the in-library multi-device call
        cap = 64
        buf = (C.c_double * (6 * cap))()
        coll, kind = C.c_double(0.0), C.c_int32(0)
        ns = _lib.lib().bssm_amd_last_is_shards(buf, C.c_int32(cap), C.byref(coll), C.byref(kind))
        out["shards"] = [dict(device=int(buf[6 * r]), first=int(buf[6 * r + 1]), count=int(buf[6 * r + 2]), setup_ms=buf[6 * r + 3],
                              device_ms=buf[6 * r + 4], total_ms=buf[6 * r + 5]) for r in range(min(ns, cap))]
        out["collective_ms"] = coll.value
        out["gather"] = {0: "none", 1: "rccl_all_gather", 2: "device_copies", 3: "host_blocks"}[int(kind.value)]
    return out


def _devices_arg(devices):
    """devices: None = the single-device entry point; "all" = every visible GPU; a list of HIP ordinals otherwise"""
    if devices is None:
        return None, 0
    if isinstance(devices, str):
        if devices != "all":
            raise ValueError("devices must be None, 'all' or a list of HIP ordinals")
        return None, 0
    d = np.ascontiguousarray(list(devices), dtype=np.int32)
    if d.size == 0:
        raise ValueError("devices must not be empty")
    return d, int(d.size)


def _run_local_block(arrs, modes, counts, first, n_stored, nsim, seed, is_type, output_type, n_threads,
                     method=1, approx_loglik=None, devices=None):
    """Filter draws [first, first+len(arrs)) on this process's GPU through the C ABI."""
    cnt = len(arrs)
    n, m = arrs[0]["n"], arrs[0]["m"]
    models, keep = _lib.pack_models(arrs)
    job = _lib.IsJob()
    job.n_stored, job.first, job.count = n_stored, first, cnt
    job.models = models
    modes_c = None if modes is None else np.asfortranarray(modes, dtype=np.float64)
    al_c = None if approx_loglik is None else np.ascontiguousarray(approx_loglik, dtype=np.float64)
    counts_c = np.ascontiguousarray(counts, dtype=np.uint32)
    job.modes = _lib.dptr(modes_c)
    job.approx_loglik = _lib.dptr(al_c)
    job.counts = counts_c.ctypes.data_as(_lib.c_u32_p)
    job.sampling_method, job.is_type, job.output_type = method, is_type, output_type
    job.nsim, job.seed, job.n_threads = nsim, seed, n_threads
    weights, loglik = np.zeros(cnt), np.zeros(cnt)
    alpha = np.zeros((n + 1, m, cnt), order="F") if output_type == 1 else None
    ah = np.zeros((m, n + 1, cnt), order="F") if output_type == 2 else None
    vt = np.zeros((m, m, n + 1, cnt), order="F") if output_type == 2 else None
    if devices is None:
        rc = _lib.check(_lib.lib().bssm_is_correction(C.byref(job), _lib.dptr(weights), _lib.dptr(loglik), _lib.dptr(alpha),
                                                      _lib.dptr(ah), _lib.dptr(vt)))
    else:  # one call, several GPUs: host threads + hipSetDevice + RCCL all-gather inside the library
        dv, nd = _devices_arg(devices)
        rc = _lib.check(_lib.lib().bssm_is_correction_multi(
            C.byref(job), None if dv is None else dv.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int32(nd), _lib.dptr(weights),
            _lib.dptr(loglik), _lib.dptr(alpha), _lib.dptr(ah), _lib.dptr(vt)))
    return dict(weights=weights, loglik=loglik, alpha=alpha, alphahat_i=ah, Vt_i=vt, degenerate=rc == 1,
                timing_ms=_last_timing(devices))


def is_correction(model: BssmModel, theta, counts, particles: int, method: str = "psi", modes=None, approx_loglik=None,
                  is_type: int = 2, output_type: int = 3, seed: int = 1, threads: int = 1, devices=None):
    """Single-process IS correction with either sampler (approx_mcmc::is_correction_psi / _bsf,
    src/approx_mcmc.cpp:353-427, 447-518) -- the step run_mcmc's `is` types run after the approximate chain."""
    counts = np.asarray(counts, dtype=np.uint32)
    if model.inherits("ssm_nlg") or model.inherits("ssm_sde"):
        # is_correction_psi / _bsf<ssm_nlg> (src/approx_mcmc.cpp:685-921) and, with the bootstrap sampler, the ssm_sde
        # correction of sde_is_mcmc / sde_state_sampler_bsf_is2 (src/approx_mcmc.cpp:1038-1273, src/R_sde_state_sampler.cpp:10-50)
        theta = np.atleast_2d(np.asarray(theta, dtype=np.float64))
        res = _run_local_block_nlg(model, theta, modes, counts, 0, theta.shape[0], particles, seed, is_type, output_type, threads,
                                   method=1 if method == "psi" else 2, approx_loglik=approx_loglik, devices=devices)
        if output_type == 2:
            m = int(model.get("n_states", 1))
            res["alphahat"], res["Vt"] = is_merge(model.n, m, res["weights"], counts, res["alphahat_i"], res["Vt_i"])
        return res
    arrs = expand_models(model, theta)
    res = _run_local_block(arrs, modes, counts, 0, len(arrs), particles, seed, is_type, output_type, threads,
                           method=1 if method == "psi" else 2, approx_loglik=approx_loglik, devices=devices)
    if output_type == 2:
        ah, vt = is_merge(model.n, model.m, res["weights"], counts, res["alphahat_i"], res["Vt_i"])
        res["alphahat"], res["Vt"] = ah, vt
    return res


def _run_local_block_nlg(model, theta, modes, counts, first, n_stored, nsim, seed, is_type, output_type, n_threads,
                         method=1, approx_loglik=None, devices=None):
    """ssm_nlg draws [first, first + len(theta)) through bssm_is_correction_nlg (only theta differs between draws)."""
    cnt = theta.shape[0]
    n = model.n
    m = int(model.get("n_states", 1))
    # one struct per draw, only the theta pointer differs: replicate the packed template bytes and patch the pointer
    # field (a Python loop over 18 720 draws cost more than the device phase)
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    s0, keep = _lib.pack_fn_model({**model, "theta": theta[0]})
    models = (_lib.FnModel * cnt)()
    sz = C.sizeof(_lib.FnModel)
    raw = np.frombuffer(models, dtype=np.uint8).reshape(cnt, sz)
    raw[:] = np.frombuffer(bytes(s0), dtype=np.uint8)
    off = _lib.FnModel.theta.offset
    raw[:, off:off + 8].view(np.uint64)[:, 0] = theta.ctypes.data + np.arange(cnt, dtype=np.uint64) * np.uint64(theta.strides[0])
    if int(s0.model_id) == 7 and any(callable(model.get(k)) for k in ("H", "R_diag", "R", "a1", "P1")):
        for i in range(cnt):  # user-defined model with theta-dependent H / R / a1 / P1 (src/model_ssm_nlg.h:10-21)
            _lib.fill_user_numbers(models[i], model, theta[i])
    modes_c = None if modes is None else np.asfortranarray(modes, dtype=np.float64)
    al_c = None if approx_loglik is None else np.ascontiguousarray(approx_loglik, dtype=np.float64)
    counts_c = np.ascontiguousarray(counts, dtype=np.uint32)
    weights, loglik = np.zeros(cnt), np.zeros(cnt)
    alpha = np.zeros((n + 1, m, cnt), order="F") if output_type == 1 else None
    ah = np.zeros((m, n + 1, cnt), order="F") if output_type == 2 else None
    vt = np.zeros((m, m, n + 1, cnt), order="F") if output_type == 2 else None
    head = (models, C.c_int32(first), C.c_int32(cnt), _lib.dptr(modes_c), _lib.dptr(al_c), counts_c.ctypes.data_as(_lib.c_u32_p),
            C.c_int32(method), C.c_int32(is_type), C.c_int32(output_type), C.c_uint32(nsim), C.c_uint32(seed), C.c_int32(n_threads))
    tail = (_lib.dptr(weights), _lib.dptr(loglik), _lib.dptr(alpha), _lib.dptr(ah), _lib.dptr(vt))
    if devices is None:
        rc = _lib.check(_lib.lib().bssm_is_correction_nlg(*head, *tail))
    else:
        dv, nd = _devices_arg(devices)
        rc = _lib.check(_lib.lib().bssm_is_correction_nlg_multi(
            *head, None if dv is None else dv.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int32(nd), *tail))
    return dict(weights=weights, loglik=loglik, alpha=alpha, alphahat_i=ah, Vt_i=vt, degenerate=rc == 1,
                timing_ms=_last_timing(devices))


def _all_gather_blocks(local: np.ndarray, bounds, rank, world, group, device):
    """All-gather variable-length blocks along the LAST axis (padded to the largest block): one collective."""
    import torch
    import torch.distributed as dist
    sizes = np.diff(bounds)
    mx = int(sizes.max())
    lead = local.shape[:-1]
    pad = np.zeros(lead + (mx,), dtype=np.float64)
    pad[..., :local.shape[-1]] = local
    t = torch.from_numpy(np.ascontiguousarray(pad)).to(device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t, group=group)
    parts = [outs[r].cpu().numpy()[..., :int(sizes[r])] for r in range(world)]
    return np.concatenate(parts, axis=-1)


def is_merge(n, m, weights, counts, alphahat_i, Vt_i):
    """West/Chan merge in draw order (src/approx_mcmc.cpp:397-414, 422-424) via the library."""
    ns = weights.size
    alphahat = np.zeros((m, n + 1), order="F")
    Vt = np.zeros((m, m, n + 1), order="F")
    w = np.ascontiguousarray(weights, dtype=np.float64)
    c = np.ascontiguousarray(counts, dtype=np.uint32)
    ah = np.asfortranarray(alphahat_i, dtype=np.float64)
    vt = np.asfortranarray(Vt_i, dtype=np.float64)
    _lib.check(_lib.lib().bssm_is_merge(n, m, ns, _lib.dptr(w), c.ctypes.data_as(_lib.c_u32_p), _lib.dptr(ah), _lib.dptr(vt),
                                        _lib.dptr(alphahat), _lib.dptr(Vt)))
    return alphahat, Vt


def post_correct(model: BssmModel, mcmc_output: dict, particles: int, threads: int = 1, is_type: str = "is2",
                 seed: int | None = None, group=None, devices=None):
    """R/post_correction.R:245-308.  ``mcmc_output`` needs: theta (n_stored x n_par), counts, modes
    (n x n_stored, or 1 x n x n_stored as R stores it), posterior, output_type ('full'|'summary'|'theta' or 1..3).
    Returns a copy with weights / posterior (and alpha or alphahat, Vt) updated.

    Two ways onto several GPUs, both cutting the draws at the same places with the same streams (identical output):
    ``devices="all"`` or a list of HIP ordinals -- ONE process, the library shards the call itself (host threads +
    hipSetDevice + one RCCL all-gather; what an R caller gets); or ``torch.distributed`` initialised with one process
    per GPU -- every rank filters its block and the results are all-gathered here."""
    particles = check_intmax(particles, "particles")
    threads = check_intmax(threads, "threads")
    seed = check_intmax(_default_seed() if seed is None else seed, "seed", False, max=_INT_MAX)
    if not isinstance(mcmc_output, dict) or "theta" not in mcmc_output or "counts" not in mcmc_output:
        raise ValueError("Object 'mcmc_output' is not valid output from 'run_mcmc'.")
    if is_type.lower() not in ("is1", "is2", "is3"):
        raise ValueError("'arg' should be one of 'is1', 'is2', 'is3'")
    is_t = int(is_type[-1])
    nlg = model.inherits("ssm_nlg")
    if not (model.inherits("nongaussian") or nlg):
        raise TypeError("Function 'post_correct' is only available for models of class 'nongaussian' and 'ssm_nlg'.")
    if not nlg and np.any(np.isnan(np.asarray(model["u"], dtype=float))):
        raise ValueError("Missing values not allowed in 'u'.")
    output_type = _OUTPUT_TYPES[mcmc_output.get("output_type", 3)]
    t0 = time.perf_counter()

    theta = np.atleast_2d(np.array(mcmc_output["theta"], dtype=np.float64))
    # run_mcmc() returns theta on the natural scale and says which columns it exponentiated: back to the sampling scale
    # the model update expects (inside run_mcmc the stored values arrive untransformed, without the marker)
    back = [int(i) for i in mcmc_output.get("transformed", [])] if mcmc_output.get("theta_scale") == "natural" else []
    if back:
        theta[:, back] = np.log(theta[:, back])
    counts = np.asarray(mcmc_output["counts"], dtype=np.uint32).reshape(-1)
    modes = np.asarray(mcmc_output["modes"], dtype=np.float64)
    n = model.n
    p_dim = (int(model.get("n_states", 1))) if nlg else model.p   # rows of one draw's mode: states (nlg) / series
    modes = modes.reshape(p_dim * n, -1, order="F")
    if theta.shape[0] != counts.size or modes.shape != (p_dim * n, counts.size):
        raise ValueError("theta, counts and modes of 'mcmc_output' do not agree")
    if is_t == 3:  # expand(): src/approx_mcmc.cpp:49-88, called from src/R_postcorrection.cpp:197-199
        rep = counts.astype(np.int64)
        theta, modes = np.repeat(theta, rep, axis=0), np.repeat(modes, rep, axis=1)
        counts = np.ones(int(rep.sum()), dtype=np.uint32)
    n_stored = counts.size

    # ---- shard the draws over the process group
    rank, world, dist = _process_group(group)
    work = counts.astype(np.float64) if is_t == 1 else np.ones(n_stored)
    bounds = shard_bounds(work, world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])

    m = (int(model.get("n_states", 1))) if nlg else model.m
    if hi > lo and nlg:
        loc = _run_local_block_nlg(model, theta[lo:hi], modes[:, lo:hi], counts[lo:hi], lo, n_stored, particles, seed, is_t,
                                   output_type, threads, devices=devices)
    elif hi > lo:
        arrs = expand_models(model, theta[lo:hi])
        loc = _run_local_block(arrs, modes[:, lo:hi], counts[lo:hi], lo, n_stored, particles, seed, is_t, output_type,
                               threads, devices=devices)
    else:
        loc = dict(weights=np.zeros(0), loglik=np.zeros(0), alpha=np.zeros((n + 1, m, 0)),
                   alphahat_i=np.zeros((m, n + 1, 0)), Vt_i=np.zeros((m, m, n + 1, 0)), degenerate=False,
                   timing_ms=dict(host_setup=0.0, device=0.0))

    # ---- ONE collective phase: all-gather the per-draw results
    if world > 1:
        import torch
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        pieces = [loc["weights"][None, :], loc["loglik"][None, :]]
        if output_type == 1:
            pieces.append(loc["alpha"].reshape(-1, hi - lo, order="F"))
        elif output_type == 2:
            pieces.append(loc["alphahat_i"].reshape(-1, hi - lo, order="F"))
            pieces.append(loc["Vt_i"].reshape(-1, hi - lo, order="F"))
        stacked = np.concatenate(pieces, axis=0)
        full = _all_gather_blocks(stacked, bounds, rank, world, group, device)
        weights, loglik = full[0], full[1]
        if output_type == 1:
            alpha = full[2:].reshape((n + 1, m, n_stored), order="F")
        elif output_type == 2:
            k1 = 2 + m * (n + 1)
            ah_i = full[2:k1].reshape((m, n + 1, n_stored), order="F")
            vt_i = full[k1:].reshape((m, m, n + 1, n_stored), order="F")
    else:
        weights, loglik = loc["weights"], loc["loglik"]
        alpha, ah_i, vt_i = loc["alpha"], loc["alphahat_i"], loc["Vt_i"]

    out = dict(mcmc_output)
    out["weights"] = weights
    with np.errstate(divide="ignore"):
        # posterior = prior (0) + approx_loglik (0) + log(weight): src/approx_mcmc.cpp:425-426, R_postcorrection.cpp:178-179
        inc = np.log(weights)
    if is_t == 3:
        if back:
            theta[:, back] = np.exp(theta[:, back])
        out["theta"], out["counts"], out["modes"] = theta, counts, modes
        out["posterior"] = np.repeat(np.asarray(mcmc_output.get("posterior", np.zeros(rep.size)), dtype=float), rep) + inc
    else:
        out["posterior"] = np.asarray(mcmc_output.get("posterior", np.zeros(n_stored)), dtype=float) + inc
    out["loglik"] = loglik
    if output_type == 1:
        out["alpha"] = alpha
    elif output_type == 2:
        alphahat, Vt = is_merge(n, m, weights, counts, ah_i, vt_i)
        out["alphahat"] = np.ascontiguousarray(alphahat.T)  # (n+1) x m, as mcmc_run.alphahat.t()
        out["Vt"] = Vt
    out["mcmc_type"] = f"is{is_t}"
    out["seed"] = list(np.atleast_1d(mcmc_output.get("seed", []))) + [seed]
    out["time_postcorrection"] = time.perf_counter() - t0
    out["shards"] = bounds
    out["timing_ms"] = loc.get("timing_ms")  # this rank's split: per-draw host set-up vs device phase
    return out


def _loglik_replications(mdl: BssmModel, particles: int, n_rep: int, seed: int, stream0: int):
    """psi-APF log-likelihoods of streams stream0 .. stream0 + n_rep - 1 as one device batch (the unit suggest_N shards)"""
    from .filters import logLik_batch
    return logLik_batch(mdl, particles, n_rep, method="psi", seed=seed, stream0=stream0)


def _suggest_N_sharded(mdl, candidates, replications, seed, dist, rank, world, group):
    """Replications of every candidate split in contiguous blocks over the ranks (independent replications
    x candidates, gather scalars): streams are those of the single-process call, so the sds agree to rounding."""
    import torch
    bounds = shard_bounds(np.ones(replications), world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    local = np.zeros((len(candidates), hi - lo))
    for i, c in enumerate(candidates):
        if hi > lo:
            local[i] = _loglik_replications(mdl, c, hi - lo, seed, i * replications + lo)
    device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    full = _all_gather_blocks(local, bounds, rank, world, group, device)
    return full.std(axis=1, ddof=1), full


def suggest_N(model: BssmModel, theta, candidates=range(10, 101, 10), replications: int = 100, seed: int | None = None,
              group=None, devices=None, return_loglik: bool = False):
    """R/post_correction.R:84-130 -> suggest_n_nongaussian (src/R_postcorrection.cpp:13-140): standard deviation
    of the psi-APF log-likelihood at ``theta`` over ``replications`` runs for each candidate N; N = first
    candidate with sd < 1 (None if there is none, R's NA).  The replications of one candidate run as ONE device
    batch (bssm_nongaussian_loglik_batch) instead of the reference's sequential loop."""
    from .filters import _prep
    replications = check_intmax(replications, "replications", max=1000)
    seed = check_intmax(_default_seed() if seed is None else seed, "seed", False, max=_INT_MAX)
    candidates = [int(c) for c in candidates]
    if not candidates or min(candidates) < 1:
        raise ValueError("Argument 'candidates' should be vector of positive integers. ")
    if max(candidates) > 1e5:
        raise ValueError("I don't believe you want to use over 1e5 particles If you really do, please file an issue at Github.")
    theta = np.asarray(theta, dtype=np.float64).reshape(-1)
    if theta.size != np.asarray(model["theta"]).size:
        raise ValueError("Length of 'theta' does not match length of 'model$theta'.")
    cand = np.asarray(candidates, dtype=np.uint32)
    sds = np.zeros(cand.size)
    rank, world, dist = _process_group(group)
    if world > 1:  # one process per GPU: the replications shard, one all-gather of the log-likelihoods
        if model.inherits("ssm_nlg"):
            mdl = BssmModel(model.classes, **{**model, "theta": theta})
        else:
            arr = expand_models(model, theta[None, :])[0]
            mdl = BssmModel(model.classes, **{**model, **{k: arr[k] for k in ("Z", "T", "R", "P1", "a1", "C", "D", "beta")},
                                              "phi": arr["phi"]})
        sds, _ = _suggest_N_sharded(mdl, candidates, replications, seed, dist, rank, world, group)
        ok = np.nonzero(sds < 1.0)[0]
        return dict(N=candidates[int(ok[0])] if ok.size else None, results=dict(N=np.asarray(candidates), sd=sds))
    # devices: None = the single-device entry point; "all" / a list of HIP ordinals = the replications of every candidate
    # sharded over those devices INSIDE the library (bssm_suggest_n_*_multi: thread per device, one exchange at the end)
    ll = np.zeros((cand.size, replications)) if (return_loglik and devices is not None) else None
    dv, nd = _devices_arg(devices)
    dvp = None if dv is None else dv.ctypes.data_as(C.POINTER(C.c_int32))

    def result():
        ok = np.nonzero(sds < 1.0)[0]
        out = dict(N=candidates[int(ok[0])] if ok.size else None, results=dict(N=np.asarray(candidates), sd=sds))
        if ll is not None:
            out["loglik"] = ll
        if devices is not None:
            out["gather"] = _last_timing(devices)["gather"]
        return out

    if model.inherits("ssm_nlg"):  # suggest_n_nonlinear, src/R_postcorrection.cpp:107-148
        smod, keep = _lib.pack_fn_model({**model, "theta": theta})
        if devices is None:
            _lib.check(_lib.lib().bssm_suggest_n_nonlinear(C.byref(smod), cand.ctypes.data_as(_lib.c_u32_p), C.c_int32(cand.size),
                                                            C.c_uint32(replications), C.c_uint32(seed), _lib.dptr(sds)))
        else:
            _lib.check(_lib.lib().bssm_suggest_n_nonlinear_multi(C.byref(smod), cand.ctypes.data_as(_lib.c_u32_p), C.c_int32(cand.size),
                                                                  C.c_uint32(replications), C.c_uint32(seed), dvp, C.c_int32(nd),
                                                                  _lib.dptr(sds), _lib.dptr(ll)))
        return result()
    arr = expand_models(model, theta[None, :])[0]
    fields = {**model, **{k: arr[k] for k in ("Z", "T", "R", "P1", "a1", "C", "D", "beta")}, "phi": arr["phi"]}
    mdl = BssmModel(model.classes, **fields)
    # one C-ABI call: bssm_suggest_n_nongaussian runs the replications of each candidate as a device batch
    arrs, smod, keep = _prep(mdl, True)
    if devices is None:
        _lib.check(_lib.lib().bssm_suggest_n_nongaussian(C.byref(smod), cand.ctypes.data_as(_lib.c_u32_p), C.c_int32(cand.size),
                                                          C.c_uint32(replications), C.c_uint32(seed), C.c_int32(model_type(mdl)),
                                                          _lib.dptr(sds)))
    else:
        _lib.check(_lib.lib().bssm_suggest_n_nongaussian_multi(C.byref(smod), cand.ctypes.data_as(_lib.c_u32_p), C.c_int32(cand.size),
                                                                C.c_uint32(replications), C.c_uint32(seed), C.c_int32(model_type(mdl)),
                                                                dvp, C.c_int32(nd), _lib.dptr(sds), _lib.dptr(ll)))
    return result()
