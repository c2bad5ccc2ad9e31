"""``run_mcmc`` for non-Gaussian models: Python mirror of R/run_mcmc.R:392-559 over the C ABI.

The chains themselves run in the library (bssm_run_mcmc, csrc/mcmc.cpp): pseudo-marginal (``pm``) and delayed
acceptance (``da``) filter on the GPU once per iteration through a device-resident plan; the approximate chain
(``approx``, and the first stage of ``is1``/``is2``/``is3``) is host arithmetic, and the IS types finish with
``post_correct``'s device pass over the stored values (src/R_mcmc.cpp:331-470).

As in the reference the standard deviations and dispersion parameters of ``bsm_ng`` / ``ar1_ng`` are sampled on the
log scale (R/run_mcmc.R:471-489) and returned on the natural scale (:537-543).
"""
from __future__ import annotations

import ctypes as C
import time

import numpy as np

from . import _lib
from .filters import check_intmax, _default_seed, _INT_MAX
from .models import BssmModel, model_arrays, model_type, prior_tables
from . import post_correct as PC

_PRIOR_FN = C.CFUNCTYPE(C.c_double, _lib.c_double_p, C.c_int32, C.c_void_p)
_UPDATE_FN = C.CFUNCTYPE(C.c_int, _lib.c_double_p, C.c_int32, C.POINTER(_lib.UngModel), C.c_void_p)


class McmcJob(C.Structure):
    _fields_ = [("model_type", C.c_int32), ("flags", C.POINTER(C.c_int32)), ("n_par", C.c_int32),
                ("prior_distributions", C.POINTER(C.c_int32)), ("prior_parameters", _lib.c_double_p),
                ("update_fn", _UPDATE_FN), ("prior_fn", _PRIOR_FN), ("user", C.c_void_p),
                ("mcmc_type", C.c_int32), ("sampling_method", C.c_uint32), ("nsim", C.c_uint32),
                ("iter", C.c_uint64), ("burnin", C.c_uint64), ("thin", C.c_uint32),
                ("gamma", C.c_double), ("target_acceptance", C.c_double), ("end_ram", C.c_int32), ("seed", C.c_uint32),
                ("output_type", C.c_int32), ("max_iter", C.c_int32), ("conv_tol", C.c_double), ("L_c", C.c_int32), ("L_f", C.c_int32), ("fn_update_fn", C.c_void_p)]


class McmcOut(C.Structure):
    _fields_ = [("capacity", C.c_int32), ("n_stored", C.c_int32), ("acceptance_rate", C.c_double),
                ("S", _lib.c_double_p), ("theta", _lib.c_double_p), ("counts", _lib.c_u32_p),
                ("posterior", _lib.c_double_p), ("approx_loglik", _lib.c_double_p), ("prior", _lib.c_double_p),
                ("modes", _lib.c_double_p), ("alpha", _lib.c_double_p), ("alphahat", _lib.c_double_p),
                ("Vt", _lib.c_double_p)]


_MCMC_TYPES = {"approx": 0, "pm": 1, "da": 2}
_UPDATABLE = ("Z", "T", "R", "a1", "P1", "D", "C", "beta")


def approximating_gaussian_model(model: BssmModel, theta_row, mode_i):
    """The Gaussian model approximate_for_is() builds at a stored mode (src/model_ssm_ung.cpp:143-161): the model at
    ``theta_row`` (sampling scale) with the pseudo-observations / standard deviations of the Laplace step.  ssm_ulg has no
    regression term, so x'beta joins D."""
    from .filters import psi_setup
    from .models import ssm_ulg
    arr = PC.expand_models(model, np.asarray(theta_row, dtype=np.float64)[None, :])[0]
    mdl = BssmModel(model.classes, **{**model, **{k: arr[k] for k in ("Z", "T", "R", "P1", "a1", "C", "D", "beta")}, "phi": arr["phi"]})
    su = psi_setup(mdl, is_mode=np.asarray(mode_i, dtype=np.float64).reshape(-1))
    n = model.n
    D = np.broadcast_to(np.asarray(arr["D"], dtype=np.float64).reshape(-1), (n,)) if np.asarray(arr["D"]).size in (1, n) else arr["D"]
    D = np.array(D, dtype=np.float64) + (arr["xreg"] @ arr["beta"] if arr["xreg"].shape[1] else 0.0)
    Z = np.asarray(arr["Z"]) if model["distribution"] != "svm" else np.eye(model.m)[:, :1]
    return ssm_ulg(su["approx_y"], Z, su["approx_H"], arr["T"], arr["R"], a1=arr["a1"], P1=arr["P1"], D=D, C=arr["C"])


def approx_state_posterior(model: BssmModel, theta, modes, seed: int):
    """approx_mcmc::approx_state_posterior (src/approx_mcmc.cpp:1399-1448): one exact draw of the states from the Gaussian
    approximation at every stored (theta, mode) -- ssm_ulg::psi_filter with one draw (the device simulation smoother,
    bssm_gaussian_psi_smoother), stream seed + 1 + i for stored value i.  Returns (n+1) x m x n_stored."""
    from .filters import _prep, _alloc
    from .models import model_type as _mt
    theta = np.atleast_2d(theta)
    n, m, ns = model.n, model.m, theta.shape[0]
    alpha = np.zeros((n + 1, m, ns), order="F")
    L = _lib.lib()
    for i in range(ns):
        g = approximating_gaussian_model(model, theta[i], modes[..., i])
        arr, smod, keep = _prep(g, False)
        a = _alloc((m, n + 1, 1))
        _lib.check(L.bssm_gaussian_psi_smoother(C.byref(smod), C.c_uint32(1), C.c_uint32((seed + 1 + i) % 2147483647), _mt(g), _lib.dptr(a)))
        alpha[:, :, i] = a[:, :, 0].T
    return alpha


def approx_state_summary(model: BssmModel, theta, modes, counts):
    """approx_mcmc::approx_state_summary (src/approx_mcmc.cpp:1494-1520): Kalman smoother of the Gaussian approximation at
    every stored (theta, mode), merged over the stored values with their counts.  Returns alphahat (n+1) x m, Vt."""
    from .filters import _prep, _alloc
    theta = np.atleast_2d(theta)
    n, m, ns = model.n, model.m, theta.shape[0]
    ah_i, vt_i = np.zeros((m, n + 1, ns), order="F"), np.zeros((m, m, n + 1, ns), order="F")
    L = _lib.lib()
    L.bssm_gaussian_smoother.argtypes = [C.POINTER(_lib.UngModel), _lib.c_double_p, _lib.c_double_p]
    for i in range(ns):
        g = approximating_gaussian_model(model, theta[i], modes[..., i])
        arr, smod, keep = _prep(g, False)
        a, v = _alloc((m, n + 1)), _alloc((m, m, n + 1))
        _lib.check(L.bssm_gaussian_smoother(C.byref(smod), _lib.dptr(a), _lib.dptr(v)))
        ah_i[:, :, i], vt_i[:, :, :, i] = a, v
    alphahat, Vt = PC.is_merge(n, m, np.ones(ns), np.asarray(counts, dtype=np.uint32), ah_i, vt_i)
    return np.ascontiguousarray(alphahat.T), Vt


def _transformed(model: BssmModel):
    """names sampled on the log scale, R/run_mcmc.R:471-489"""
    names = list(model.get("theta_names", []))
    dispersion = model["distribution"] in ("negative binomial", "gamma")
    if model.inherits("bsm_ng"):
        return [i for i, nm in enumerate(names) if nm.startswith("sd_") or (nm == "phi" and dispersion)]
    if model.inherits("ar1_ng"):
        return [i for i, nm in enumerate(names) if nm == "sigma" or (nm == "phi" and dispersion)]
    return []


def run_mcmc(model: BssmModel, iter: int, particles: int | None = None, output_type: str = "full", mcmc_type: str = "is2",
             sampling_method: str = "psi", burnin: int | None = None, thin: int = 1, gamma: float = 2 / 3,
             target_acceptance: float = 0.234, S=None, end_adaptive_phase: bool = False, local_approx: bool = True,
             threads: int = 1, seed: int | None = None, max_iter: int = 100, conv_tol: float = 1e-8,
             L_c: int | None = None, L_f: int | None = None):
    """R/run_mcmc.R:392-559 (run_mcmc.nongaussian).  Returns the ``mcmc_output`` dictionary: theta (n_stored x n_par,
    natural scale), counts, posterior, S, acceptance_rate, and by type weights / modes (approx, is*), alpha
    ((n+1) x m x n_stored, output 'full'), alphahat ((n+1) x m) and Vt (output 'summary')."""
    if model.inherits("ssm_nlg") or model.inherits("ssm_sde"):
        return _run_mcmc_fn(model, iter, particles, output_type, mcmc_type, sampling_method, burnin, thin, gamma,
                            target_acceptance, S, end_adaptive_phase, threads, seed, max_iter, conv_tol, L_c, L_f)
    if not model.inherits("nongaussian"):
        raise TypeError("run_mcmc covers models of class 'nongaussian', 'ssm_nlg' and 'ssm_sde'")
    if not isinstance(end_adaptive_phase, (bool, np.bool_)):
        raise ValueError("Argument 'end_adaptive_phase' should be TRUE or FALSE. ")
    if not isinstance(local_approx, (bool, np.bool_)):
        raise ValueError("Argument 'local_approx' should be TRUE or FALSE. ")
    seed = check_intmax(_default_seed() if seed is None else seed, "seed", False, max=_INT_MAX)
    threads = check_intmax(threads, "threads")
    thin = check_intmax(thin, "thin", max=100)
    iter = check_intmax(iter, "iter", max=int(1e12))
    burnin = check_intmax(iter // 2 if burnin is None else burnin, "burnin", False, max=int(1e12))
    if burnin > iter:
        raise ValueError("Argument 'burnin' should be smaller than 'iter'.")
    for nm, v in (("target_acceptance", target_acceptance), ("gamma", gamma)):
        if not (0 < v < 1):
            raise ValueError(f"Argument '{nm}' must be on interval (0, 1).")
    if np.asarray(model["theta"]).size == 0:
        raise ValueError("No unknown parameters ('model$theta' has length of zero).")
    if np.any(np.isnan(np.asarray(model["u"], dtype=float))):
        raise ValueError("Missing values not allowed in 'u'.")
    otype = PC._OUTPUT_TYPES[output_type.lower() if isinstance(output_type, str) else output_type]
    mcmc_type = mcmc_type.lower()
    if mcmc_type not in ("pm", "da", "is1", "is2", "is3", "approx"):
        raise ValueError("'arg' should be one of 'pm', 'da', 'is1', 'is2', 'is3', 'approx'")
    if mcmc_type == "approx":
        particles = 0
    elif particles is None or particles < 2:
        raise ValueError("Number of state samples less than 2, use 'mcmc_type' 'approx' instead.")
    else:
        particles = check_intmax(particles, "particles")
    sampling_method = sampling_method.lower()
    if sampling_method not in ("psi", "bsf"):
        raise ValueError("sampling_method must be 'psi' or 'bsf' (SPDK is not a particle filter path)")
    t0 = time.perf_counter()

    model = BssmModel(model.classes, **{**model, "max_iter": check_intmax(max_iter, "max_iter", False),
                                        "conv_tol": float(conv_tol), "local_approx": bool(local_approx)})
    theta0 = np.array(model["theta"], dtype=np.float64).reshape(-1)
    d = theta0.size
    tr = _transformed(model)
    theta0[tr] = np.log(np.maximum(0.001, theta0[tr]))
    S_in = np.diag(0.1 * np.maximum(0.1, np.abs(theta0))) if S is None else np.array(S, dtype=np.float64).reshape(d, d)
    S_buf = np.asfortranarray(np.tril(S_in))

    # ---- the model struct over writable buffers
    arr = model_arrays(model)
    for k in _UPDATABLE + ("initial_mode",):
        if arr.get(k) is not None:
            arr[k] = np.array(arr[k], dtype=np.float64, order="F", copy=True)
    smod, keep = _lib.pack_model(arr)
    mt = model_type(model)
    flags = PC._builtin_flags(model)

    job = McmcJob()
    job.model_type, job.n_par = mt, d
    cb_keep = []
    if flags is not None:
        fl = (C.c_int32 * len(flags))(*flags)
        pd, pp = prior_tables(model["priors"])
        if pd.size != d:
            raise ValueError("the model's priors do not match its theta")
        job.flags = fl
        job.prior_distributions = pd.ctypes.data_as(C.POINTER(C.c_int32))
        job.prior_parameters = _lib.dptr(pp)
        cb_keep += [fl, pd, pp]
    else:
        update, prior = model.get("update_fn"), model.get("prior_fn")
        if update is None or prior is None:
            raise ValueError("ssm_ung / ssm_mng models need 'update_fn' and 'prior_fn' for run_mcmc")
        failure = []

        def _update(th_p, n_par, mod_p, _user):
            try:
                th = np.ctypeslib.as_array(th_p, shape=(n_par,)).copy()
                for k, v in (update(th) or {}).items():
                    if k == "phi":
                        if arr.get("p", 1) > 1:
                            arr["phi_p"][...] = np.asarray(v, dtype=np.float64).reshape(-1)
                        else:
                            mod_p.contents.phi = float(v)
                    elif k in _UPDATABLE:
                        arr[k][...] = np.asarray(v, dtype=np.float64).reshape(arr[k].shape, order="F")
                    else:
                        raise ValueError(f"update_fn returned unknown component '{k}'")
                return 0
            except Exception as e:  # reported through the return code: exceptions cannot cross the C frames
                failure.append(e)
                return 1

        def _prior(th_p, n_par, _user):
            try:
                return float(prior(np.ctypeslib.as_array(th_p, shape=(n_par,)).copy()))
            except Exception as e:
                failure.append(e)
                return float("nan")

        job.update_fn, job.prior_fn = _UPDATE_FN(_update), _PRIOR_FN(_prior)
        cb_keep += [job.update_fn, job.prior_fn]
    chain = _MCMC_TYPES.get(mcmc_type, 0)
    job.mcmc_type = chain
    job.sampling_method = 1 if sampling_method == "psi" else 2
    job.nsim = particles
    job.iter, job.burnin, job.thin = iter, burnin, thin
    job.gamma, job.target_acceptance, job.end_ram, job.seed = gamma, target_acceptance, int(end_adaptive_phase), seed
    job.output_type = otype

    n, m, p = model.n, model.m, model.p
    cap = max(1, (iter - burnin) // thin + 1)
    out = McmcOut()
    out.capacity = cap
    theta_s = np.zeros((d, cap), order="F")
    counts = np.zeros(cap, dtype=np.uint32)
    posterior = np.zeros(cap)
    out.S, out.theta, out.counts, out.posterior = _lib.dptr(S_buf), _lib.dptr(theta_s), counts.ctypes.data_as(_lib.c_u32_p), _lib.dptr(posterior)
    al = prior_s = modes = alpha = ah = vt = None
    if chain == 0:
        al, prior_s, modes = np.zeros(cap), np.zeros(cap), np.zeros((p * n, cap), order="F")
        out.approx_loglik, out.prior, out.modes = _lib.dptr(al), _lib.dptr(prior_s), _lib.dptr(modes)
    elif otype == 1:
        alpha = np.zeros((n + 1, m, cap), order="F")
        out.alpha = _lib.dptr(alpha)
    elif otype == 2:
        ah, vt = np.zeros((m, n + 1), order="F"), np.zeros((m, m, n + 1), order="F")
        out.alphahat, out.Vt = _lib.dptr(ah), _lib.dptr(vt)
    L = _lib.lib()
    L.bssm_run_mcmc.argtypes = [C.POINTER(_lib.UngModel), _lib.c_double_p, C.POINTER(McmcJob), C.POINTER(McmcOut)]
    rc = L.bssm_run_mcmc(C.byref(smod), _lib.dptr(theta0), C.byref(job), C.byref(out))
    if flags is None and failure:
        raise failure[0]
    _lib.check(rc)
    ns = out.n_stored
    res = dict(theta=np.ascontiguousarray(theta_s[:, :ns].T), counts=counts[:ns].copy(), posterior=posterior[:ns].copy(),
               S=np.array(S_buf), acceptance_rate=out.acceptance_rate, iter=iter, burnin=burnin, thin=thin,
               mcmc_type=mcmc_type, output_type=otype, seed=seed, theta_names=list(model.get("theta_names", [])),
               model_type=model.classes[0], time_chain=time.perf_counter() - t0)
    if chain == 0:
        res["modes"] = modes[:, :ns].reshape(p, n, ns, order="F") if p > 1 else modes[:, :ns].reshape(n, 1, ns, order="F")
        res["approx_loglik"], res["prior"] = al[:ns].copy(), prior_s[:ns].copy()
        res["weights"] = np.ones(ns)
        if mcmc_type == "approx" and otype == 1 and p == 1:
            # state draws of the purely approximate chain (nongaussian_is_mcmc with approx = TRUE, src/R_mcmc.cpp:397-404)
            res["alpha"] = approx_state_posterior(model, res["theta"], res["modes"], seed)
        if mcmc_type == "approx" and otype == 2 and p == 1:
            res["alphahat"], res["Vt"] = approx_state_summary(model, res["theta"], res["modes"], res["counts"])
        if mcmc_type != "approx":
            # IS correction of the stored values on the sampling scale, as nongaussian_is_mcmc does before run_mcmc.R
            # maps theta back (src/R_mcmc.cpp:376-392)
            tc = time.perf_counter()
            if sampling_method == "psi":
                corrected = PC.post_correct(model, res, particles, threads=threads, is_type=mcmc_type, seed=seed)
            else:  # approx_mcmc::is_correction_bsf (src/approx_mcmc.cpp:447-518): weight = exp(loglik - approx_loglik)
                it = int(mcmc_type[-1])
                th_c, cn_c, al_c, post_c = res["theta"], res["counts"], res["approx_loglik"], res["posterior"]
                if it == 3:  # expand(), src/approx_mcmc.cpp:49-88
                    rep = cn_c.astype(np.int64)
                    th_c, al_c, post_c = np.repeat(th_c, rep, axis=0), np.repeat(al_c, rep), np.repeat(post_c, rep)
                    cn_c = np.ones(int(rep.sum()), dtype=np.uint32)
                r = PC.is_correction(model, th_c, cn_c, particles, method="bsf", approx_loglik=al_c, is_type=it,
                                     output_type=otype, seed=seed, threads=threads)
                with np.errstate(divide="ignore"):
                    corrected = dict(weights=r["weights"], loglik=r["loglik"], theta=th_c, counts=cn_c,
                                     posterior=post_c + np.log(r["weights"]), timing_ms=r.get("timing_ms"))
                if otype == 1:
                    corrected["alpha"] = r["alpha"]
                elif otype == 2:
                    corrected["alphahat"], corrected["Vt"] = np.ascontiguousarray(r["alphahat"].T), r["Vt"]
            for k in ("weights", "posterior", "theta", "counts", "modes", "alpha", "alphahat", "Vt", "loglik", "timing_ms"):
                if k in corrected:
                    res[k] = corrected[k]
            res["time_postcorrection"] = time.perf_counter() - tc
    elif otype == 1:
        res["alpha"] = alpha[:, :, :ns].copy()
    elif otype == 2:
        res["alphahat"], res["Vt"] = np.ascontiguousarray(ah.T), vt
    res["theta"] = np.array(res["theta"])
    res["theta"][:, tr] = np.exp(res["theta"][:, tr])
    # theta is now on the natural scale (run_mcmc.R maps the log-sd / log-phi columns back); post_correct() needs the
    # sampling scale again and reads which columns to log from here (the reference passes the stored theta unchanged,
    # R/post_correction.R:266, and so filters at exp(sd))
    res["theta_scale"], res["transformed"] = "natural", [int(i) for i in np.atleast_1d(tr)]
    res["time"] = time.perf_counter() - t0
    return res


def _run_mcmc_fn(model, iter, particles, output_type, mcmc_type, sampling_method, burnin, thin, gamma, target_acceptance, S,
                 end_adaptive_phase, threads, seed, max_iter, conv_tol, L_c=None, L_f=None):
    """run_mcmc.ssm_nlg / run_mcmc.ssm_sde (R/run_mcmc.R:561-900) for the model functors: the chain runs in the library
    (bssm_run_mcmc_fn), the prior is ``model['prior_fn']`` or the shipped example's log_prior_pdf."""
    from .models import example_log_prior
    sde = model.inherits("ssm_sde")
    seed = check_intmax(_default_seed() if seed is None else seed, "seed", False, max=_INT_MAX)
    threads = check_intmax(threads, "threads")
    thin = check_intmax(thin, "thin", max=100)
    iter = check_intmax(iter, "iter", max=int(1e12))
    burnin = check_intmax(iter // 2 if burnin is None else burnin, "burnin", False, max=int(1e12))
    if burnin > iter:
        raise ValueError("Argument 'burnin' should be smaller than 'iter'.")
    otype = PC._OUTPUT_TYPES[output_type.lower() if isinstance(output_type, str) else output_type]
    mcmc_type = mcmc_type.lower()
    sde_is = None
    if sde:  # run_mcmc.ssm_sde, R/run_mcmc.R:733-860: discretisation levels 2^L_c (chain) / 2^L_f (correction, pm)
        sampling_method = "bsf"
        if mcmc_type not in ("pm", "da", "is1", "is2", "is3"):
            raise ValueError("'arg' should be one of 'pm', 'da', 'is1', 'is2', 'is3'")
        sde_levels = (0, 0)
        if mcmc_type == "pm":
            L = max(int(L_c or 0), int(L_f or 0), 0) or int(model.get("L", 0))
            if L <= 0:
                raise ValueError("L should be positive.")
            model = BssmModel(model.classes, **{**model, "L": L})
        else:
            if L_c is None or L_f is None:
                raise ValueError("ssm_sde IS types need L_c and L_f")
            if L_f <= L_c:
                raise ValueError("L_f should be larger than L_c.")
            if L_c < 1:
                raise ValueError("L_c should be at least 1")
        if mcmc_type == "da":  # sde_da_mcmc: coarse filter first, fine filter for the proposals that pass
            sde_levels = (int(L_c), int(L_f))
            model = BssmModel(model.classes, **{**model, "L": int(L_f)})
        elif mcmc_type != "pm":
            # the chain is the pseudo-marginal chain at the coarse level (approx_mcmc::amcmc(ssm_sde), src/approx_mcmc.cpp:236-330)
            sde_is, mcmc_type = mcmc_type, "pm"
            model_fine = BssmModel(model.classes, **{**model, "L": int(L_f)})
            model = BssmModel(model.classes, **{**model, "L": int(L_c)})
    elif mcmc_type not in ("pm", "da", "is1", "is2", "is3", "approx"):
        raise ValueError("'arg' should be one of 'pm', 'da', 'is1', 'is2', 'is3', 'approx'")
    sampling_method = sampling_method.lower()
    if sampling_method == "ekf" and mcmc_type.lower() != "pm":
        # the reference's approximate chain / DA first stage / IS correction of method "ekf" use ekf_loglik() and an
        # EKPF correction (src/model_ssm_nlg.cpp:259-261); only the pseudo-marginal chain is built for it here
        raise ValueError("sampling_method 'ekf' is available with mcmc_type 'pm' only (use 'psi' or 'bsf' for approx / da / is*)")
    if sampling_method not in ("psi", "bsf", "ekf"):
        raise ValueError("'arg' should be one of 'psi', 'bsf', 'ekf'")
    if mcmc_type == "approx":
        particles = 0
    elif particles is None or particles < 2:
        raise ValueError("Number of state samples less than 2, use 'mcmc_type' 'approx' instead.")
    else:
        particles = check_intmax(particles, "particles")
    t0 = time.perf_counter()
    theta0 = np.array(model["theta"], dtype=np.float64).reshape(-1)
    d = theta0.size
    if d == 0:
        raise ValueError("No unknown parameters ('model$theta' has length of zero).")
    prior = model.get("prior_fn") or (example_log_prior(model["example"]) if model.get("example") not in (None, "user") else None)
    if prior is None:
        raise ValueError("user-defined models need 'prior_fn' (the model's log_prior_pdf) for run_mcmc")
    S_in = np.diag(0.1 * np.maximum(0.1, np.abs(theta0))) if S is None else np.array(S, dtype=np.float64).reshape(d, d)
    S_buf = np.asfortranarray(np.tril(S_in))
    failure = []

    def _prior(th_p, n_par, _user):
        try:
            return float(prior(np.ctypeslib.as_array(th_p, shape=(n_par,)).copy()))
        except Exception as e:  # exceptions cannot cross the C frames
            failure.append(e)
            return float("nan")

    smod, keep = _lib.pack_fn_model(model)
    job = McmcJob()
    job.n_par = d
    job.prior_fn = _PRIOR_FN(_prior)
    if int(model.get("model_id", 0)) == 7 and any(callable(model.get(k)) for k in ("H", "R_diag", "R", "a1", "P1")):
        FN_UPDATE = C.CFUNCTYPE(C.c_int, _lib.c_double_p, C.c_int32, C.POINTER(_lib.FnModel), C.c_void_p)

        def _fn_update(th_p, n_par, mod_p, _user):
            try:
                _lib.fill_user_numbers(mod_p.contents, model, np.ctypeslib.as_array(th_p, shape=(n_par,)).copy())
                return 0
            except Exception as e:
                failure.append(e)
                return 1
        fn_update_cb = FN_UPDATE(_fn_update)
        job.fn_update_fn = C.cast(fn_update_cb, C.c_void_p)
    chain = _MCMC_TYPES.get(mcmc_type, 0)
    job.mcmc_type = chain
    job.sampling_method = {"psi": 1, "bsf": 2, "ekf": 4}[sampling_method]
    job.nsim = particles
    job.iter, job.burnin, job.thin = iter, burnin, thin
    job.gamma, job.target_acceptance, job.end_ram, job.seed = gamma, target_acceptance, int(bool(end_adaptive_phase)), seed
    job.output_type = 3 if sde_is else otype
    job.max_iter, job.conv_tol = int(max_iter), float(conv_tol)
    if sde:
        job.L_c, job.L_f = sde_levels
    n, m = model.n, int(model.get("n_states", 1))
    cap = max(1, (iter - burnin) // thin + 1)
    out = McmcOut()
    out.capacity = cap
    theta_s = np.zeros((d, cap), order="F")
    counts = np.zeros(cap, dtype=np.uint32)
    posterior = np.zeros(cap)
    out.S, out.theta, out.counts, out.posterior = _lib.dptr(S_buf), _lib.dptr(theta_s), counts.ctypes.data_as(_lib.c_u32_p), _lib.dptr(posterior)
    al = prior_s = modes = alpha = ah = vt = None
    if chain == 0:
        al, prior_s, modes = np.zeros(cap), np.zeros(cap), np.zeros((m * n, cap), order="F")
        out.approx_loglik, out.prior, out.modes = _lib.dptr(al), _lib.dptr(prior_s), _lib.dptr(modes)
    elif sde_is:
        al, prior_s = np.zeros(cap), np.zeros(cap)
        out.approx_loglik, out.prior = _lib.dptr(al), _lib.dptr(prior_s)
    elif otype == 1:
        alpha = np.zeros((n + 1, m, cap), order="F")
        out.alpha = _lib.dptr(alpha)
    elif otype == 2:
        ah, vt = np.zeros((m, n + 1), order="F"), np.zeros((m, m, n + 1), order="F")
        out.alphahat, out.Vt = _lib.dptr(ah), _lib.dptr(vt)
    L = _lib.lib()
    L.bssm_run_mcmc_fn.argtypes = [C.POINTER(_lib.FnModel), _lib.c_double_p, C.POINTER(McmcJob), C.POINTER(McmcOut)]
    rc = L.bssm_run_mcmc_fn(C.byref(smod), _lib.dptr(theta0), C.byref(job), C.byref(out))
    if failure:
        raise failure[0]
    _lib.check(rc)
    ns = out.n_stored
    res = dict(theta=np.ascontiguousarray(theta_s[:, :ns].T), counts=counts[:ns].copy(), posterior=posterior[:ns].copy(),
               S=np.array(S_buf), acceptance_rate=out.acceptance_rate, iter=iter, burnin=burnin, thin=thin,
               mcmc_type=mcmc_type, output_type=otype, seed=seed, theta_names=list(model.get("theta_names", [])),
               model_type=model.classes[0], time_chain=time.perf_counter() - t0)
    if chain == 0:
        res["modes"] = modes[:, :ns].reshape(m, n, ns, order="F")
        res["approx_loglik"], res["prior"] = al[:ns].copy(), prior_s[:ns].copy()
        res["weights"] = np.ones(ns)
        if mcmc_type != "approx":
            tc = time.perf_counter()
            if sampling_method == "psi":
                corrected = PC.post_correct(model, res, particles, threads=threads, is_type=mcmc_type, seed=seed)
            else:  # is_correction_bsf<ssm_nlg> (src/approx_mcmc.cpp:807-921)
                it = int(mcmc_type[-1])
                th_c, cn_c, al_c, post_c = res["theta"], res["counts"], res["approx_loglik"], res["posterior"]
                if it == 3:
                    rep = cn_c.astype(np.int64)
                    th_c, al_c, post_c = np.repeat(th_c, rep, axis=0), np.repeat(al_c, rep), np.repeat(post_c, rep)
                    cn_c = np.ones(int(rep.sum()), dtype=np.uint32)
                r = PC.is_correction(model, th_c, cn_c, particles, method="bsf", approx_loglik=al_c, is_type=it,
                                     output_type=otype, seed=seed, threads=threads)
                with np.errstate(divide="ignore"):
                    corrected = dict(weights=r["weights"], loglik=r["loglik"], theta=th_c, counts=cn_c,
                                     posterior=post_c + np.log(r["weights"]), timing_ms=r.get("timing_ms"))
                if otype == 1:
                    corrected["alpha"] = r["alpha"]
                elif otype == 2:
                    corrected["alphahat"], corrected["Vt"] = np.ascontiguousarray(r["alphahat"].T), r["Vt"]
            for k in ("weights", "posterior", "theta", "counts", "modes", "alpha", "alphahat", "Vt", "loglik", "timing_ms"):
                if k in corrected:
                    res[k] = corrected[k]
            res["time_postcorrection"] = time.perf_counter() - tc
    elif sde_is:  # correction at the fine level: weight = exp(loglik_fine - loglik_coarse), src/approx_mcmc.cpp:523-600
        tc = time.perf_counter()
        it = int(sde_is[-1])
        res["approx_loglik"], res["prior"] = al[:ns].copy(), prior_s[:ns].copy()
        th_c, cn_c, al_c, post_c = res["theta"], res["counts"], res["approx_loglik"], res["posterior"]
        if it == 3:
            rep = cn_c.astype(np.int64)
            th_c, al_c, post_c = np.repeat(th_c, rep, axis=0), np.repeat(al_c, rep), np.repeat(post_c, rep)
            cn_c = np.ones(int(rep.sum()), dtype=np.uint32)
        r = PC.is_correction(model_fine, th_c, cn_c, particles, method="bsf", approx_loglik=al_c, is_type=it, output_type=otype,
                             seed=seed, threads=threads)
        with np.errstate(divide="ignore"):
            res.update(weights=r["weights"], loglik=r["loglik"], theta=th_c, counts=cn_c, posterior=post_c + np.log(r["weights"]))
        if otype == 1:
            res["alpha"] = r["alpha"]
        elif otype == 2:
            res["alphahat"], res["Vt"] = np.ascontiguousarray(r["alphahat"].T), r["Vt"]
        res["mcmc_type"] = sde_is
        res["time_postcorrection"] = time.perf_counter() - tc
    elif otype == 1:
        res["alpha"] = alpha[:, :, :ns].copy()
    elif otype == 2:
        res["alphahat"], res["Vt"] = np.ascontiguousarray(ah.T), vt
    res["time"] = time.perf_counter() - t0
    return res


def summary(out: dict, variable: str = "theta", times=None):
    """Weighted posterior means and standard deviations (R/summary.R: weights = counts x IS weights): for
    ``variable='theta'`` per parameter, for ``'states'`` per state at the (1-based) ``times`` from the stored
    trajectories (output 'full') or the running summary (output 'summary')."""
    w = np.asarray(out["counts"], dtype=np.float64) * np.asarray(out.get("weights", np.ones(len(out["counts"]))), dtype=np.float64)
    w = w / w.sum()
    if variable == "theta":
        mean = w @ out["theta"]
        sd = np.sqrt(np.maximum(0.0, w @ (out["theta"] - mean) ** 2))
        return dict(Mean=mean, SD=sd, names=out.get("theta_names"))
    if variable != "states":
        raise ValueError("variable must be 'theta' or 'states'")
    if "alpha" in out:
        a = out["alpha"]  # (n+1) x m x n_stored
        mean = a @ w
        sd = np.sqrt(np.maximum(0.0, (a - mean[:, :, None]) ** 2 @ w))
    elif "alphahat" in out:
        mean = np.asarray(out["alphahat"])
        sd = np.sqrt(np.stack([np.diag(out["Vt"][:, :, t]) for t in range(mean.shape[0])]))
    else:
        raise ValueError("the run kept no state output (output_type 'theta')")
    if times is not None:
        idx = np.atleast_1d(np.asarray(times, dtype=int)) - 1
        mean, sd = mean[idx], sd[idx]
    return dict(Mean=mean, SD=sd)


def asymptotic_var(x, w=None):
    """Asymptotic variance of the IS-weighted chain mean (R/asymptotic_var.R:9-33): with z = w (x - weighted mean), the
    long-run variance of z over n and mean(w)^2.  The long-run variance uses the initial-positive-sequence estimator of the
    autocovariances (the reference's 'geyer' flavour)."""
    x = np.asarray(x, dtype=np.float64)
    w = np.ones_like(x) if w is None else np.asarray(w, dtype=np.float64)
    if np.any(w < 0) or not np.all(np.isfinite(w)):
        raise ValueError("Nonfinite or negative weights in 'w'.")
    if not np.any(w > 0):
        raise ValueError("No positive weights in 'w'.")
    mu = np.sum(w * x) / np.sum(w)
    z = w * (x - mu)
    n = z.size
    zc = z - z.mean()
    f = np.fft.rfft(zc, 2 * n)
    acov = np.fft.irfft(f * np.conj(f))[:n] / n
    lrv = -acov[0]
    for k in range(0, n - 1, 2):
        g = acov[k] + acov[k + 1]
        if g <= 0:
            break
        lrv += 2 * g
    return float(max(lrv, 0.0) / n / np.mean(w) ** 2)
