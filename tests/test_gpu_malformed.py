"""Malformed models handed straight to the C ABI (no Python-side checks in between): every entry point must come back with a
negative status and a message -- or, for inputs that are merely unusual, with a result -- and the library must still work
afterwards.  The reference validates in R (R/check_arguments.R) before its C++ sees a model; a C caller of this library has no
such layer in front of it.  The calls run in a child process: a crash is reported as a test failure, not as a dead test run."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CHILD = r'''
import ctypes as C, sys, copy
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from bssm_amd import _lib, models as M
from models_zoo import negbin_m4_informative, mng_zoo, poisson_local_level

L = _lib.lib()
N = 256

def outputs(n, m, N):
    o = _lib.FilterOut()
    keep = [np.zeros((n + 1, m)), np.zeros((n, m)), np.zeros((m, m, n + 1)), np.zeros((m, m, n)), np.zeros((N, n + 1)), np.zeros((m, n + 1, N))]
    o.at, o.att, o.Pt, o.Ptt, o.weights, o.alpha = map(_lib.dptr, keep)
    return o, keep

def call(arr, nsim=N, model_type=None, psi=False):
    s, keep = _lib.pack_model(arr)
    n, m = max(int(arr["n"]), 1), max(min(int(arr["m"]), 64), 1)
    o, k2 = outputs(n, m, max(nsim, 1) if nsim < 100000 else 1)
    mt = model_type if model_type is not None else (0 if int(arr.get("p", 1)) != 1 else 1)
    if psi:
        o.alphahat, o.Vt = o.at, o.Pt
        return L.bssm_psi_smoother(C.byref(s), C.c_uint32(nsim), C.c_uint32(3), C.c_int(mt), C.byref(o))
    return L.bssm_bsf(C.byref(s), C.c_uint32(nsim), C.c_uint32(3), C.c_int(0), C.c_int(mt), C.byref(o))

base = M.model_arrays(negbin_m4_informative(12))
mng = M.model_arrays(dict(mng_zoo())["mixed_m4_p3"])
assert call(base) >= 0 and call(mng) >= 0 and call(base, psi=True) >= 0

def mutated(arr, **kw):
    a = dict(arr)
    for k, v in kw.items():
        a[k] = v
    return a

nanT = np.array(base["T"], dtype=float, copy=True); nanT.flat[0] = np.nan
infP = np.array(base["P1"], dtype=float, copy=True); infP.flat[0] = np.inf
negP = np.array(base["P1"], dtype=float, copy=True); negP.flat[0] = -1.0
nany = np.full_like(np.asarray(base["y"], dtype=float), np.nan)
cases = {
    "n = 0": (mutated(base, n=0), {}), "n < 0": (mutated(base, n=-5), {}),
    "m = 0": (mutated(base, m=0), {}), "m = 65": (mutated(base, m=65), {}), "m < 0": (mutated(base, m=-1), {}),
    "k = 0": (mutated(base, k=0), {}), "k > m": (mutated(base, k=9), {}),
    "distribution 99": (mutated(base, distribution=99), {}), "distribution -1": (mutated(base, distribution=-1), {}),
    "y NULL": (mutated(base, y=None), {}), "Z NULL": (mutated(base, Z=None), {}), "T NULL": (mutated(base, T=None), {}),
    "R NULL": (mutated(base, R=None), {}), "a1 NULL": (mutated(base, a1=None), {}), "P1 NULL": (mutated(base, P1=None), {}),
    "negbin phi <= 0": (mutated(base, phi=-1.0), {}),
    "NaN in T": (mutated(base, T=nanT), {}), "Inf in P1": (mutated(base, P1=infP), {}),
    "nsim = 0": (base, dict(nsim=0)),
    "unknown model_type": (base, dict(model_type=77)),
    "p = 1000": (mutated(mng, p=1000), {}),
    "mng without distribution_p": (mutated(mng, distribution_p=np.zeros(0, dtype=np.int32)), {}),
    "psi: NaN in T": (mutated(base, T=nanT), dict(psi=True)),
}
# (a negative diagonal entry of P1: the reference's psd_chol skips non-positive pivots, src/psd_chol.cpp -- the state starts without variance there)
# p <= 0 (a zero-initialised struct): the univariate fields are read, as documented in include/bssm_amd.h;  max_iter = 0 / a
# negative conv_tol: the mode search loop of the reference (src/model_ssm_ung.cpp:141-203) runs zero / max_iter times
benign = {"p = 0": (mutated(base, p=0), {}), "p < 0": (mutated(base, p=-3), {}),
          "psi: max_iter = 0": (mutated(base, max_iter=0), dict(psi=True)), "psi: negative conv_tol": (mutated(base, conv_tol=-1.0), dict(psi=True)),
          "negative P1 diagonal": (mutated(base, P1=negP), {}), "all observations missing": (mutated(base, y=nany), {}), "psi: all observations missing": (mutated(base, y=nany), dict(psi=True))}
report, bad = {}, []
for name, (arr, kw) in {**cases, **benign}.items():
    try:
        rc = call(arr, **kw)
    except (TypeError, ValueError, KeyError, IndexError) as e:   # the Python packer refused it: not the library's business
        report[name] = ("packer", type(e).__name__)
        continue
    msg = L.bssm_amd_last_error().decode() if rc < 0 else ""
    report[name] = (rc, msg[:90])
    if name in cases:
        # a malformed model is refused (negative status with a message) or, where the reference itself would run into
        # non-finite weights, reported as degenerate (status 1, the reference's warning) -- never silently accepted as OK
        if not ((rc < 0 and msg) or rc == 1):
            bad.append((name, rc, msg))
    elif rc < 0:
        bad.append((name, rc, msg))
    if call(base) < 0:
        bad.append(("library unusable after: " + name, 0, ""))

# ---- IS post-correction jobs (bssm_is_correction) and functor models (bssm_bsf_nlg / _sde)
from bssm_amd import post_correct as PC
rng = np.random.default_rng(4)
y = rng.poisson(4.0, 30).astype(float)
pcm = M.bsm_ng(y, sd_level=M.halfnormal(0.2, 1), sd_slope=M.halfnormal(0.02, 1), distribution="poisson", P1=np.diag([1.0, 0.1]))
ns = 5
theta = np.log(np.column_stack([rng.uniform(0.1, 0.4, ns), rng.uniform(0.01, 0.05, ns)]))
arrs = PC.expand_models(pcm, theta)
n, m = arrs[0]["n"], arrs[0]["m"]

def is_call(**kw):
    models, keep = _lib.pack_models(kw.pop("arrs", arrs))
    job = _lib.IsJob()
    job.n_stored, job.first, job.count = ns, kw.pop("first", 0), kw.pop("count", ns)
    job.models = None if kw.pop("no_models", False) else models
    al = np.zeros(ns); counts = np.ones(ns, dtype=np.uint32)
    job.approx_loglik = None if kw.pop("no_al", False) else _lib.dptr(al)
    job.counts = None if kw.pop("no_counts", False) else counts.ctypes.data_as(_lib.c_u32_p)
    job.sampling_method, job.is_type, job.output_type = kw.pop("method", 2), kw.pop("is_type", 2), kw.pop("output_type", 3)
    job.nsim, job.seed, job.n_threads = kw.pop("nsim", 32), 5, kw.pop("threads", 2)
    w, ll = np.zeros(ns), np.zeros(ns)
    alpha = np.zeros((n + 1, m, ns), order="F")
    ah, vt = np.zeros((m, n + 1, ns), order="F"), np.zeros((m, m, n + 1, ns), order="F")
    no_out = kw.pop("no_outputs", False)
    assert not kw, kw
    return L.bssm_is_correction(C.byref(job), None if kw.get("no_w") else _lib.dptr(w), _lib.dptr(ll), None if no_out else _lib.dptr(alpha),
                                None if no_out else _lib.dptr(ah), None if no_out else _lib.dptr(vt))

assert is_call() >= 0 and is_call(output_type=1) >= 0 and is_call(output_type=2) >= 0
short = [dict(a) for a in arrs]
short[3] = M.model_arrays(M.bsm_ng(y[:20], sd_level=0.2, sd_slope=0.02, distribution="poisson", P1=np.diag([1.0, 0.1])))
is_cases = {
    "is: models NULL": dict(no_models=True), "is: counts NULL": dict(no_counts=True), "is: approx_loglik NULL (bsf)": dict(no_al=True),
    "is: psi without modes": dict(method=1), "is: nsim = 0": dict(nsim=0), "is: is_type 0": dict(is_type=0), "is: is_type 9": dict(is_type=9),
    "is: output_type 0": dict(output_type=0), "is: output_type 7": dict(output_type=7), "is: first < 0": dict(first=-1),
    "is: output 1 without buffer": dict(output_type=1, no_outputs=True), "is: output 2 without buffers": dict(output_type=2, no_outputs=True),
    "is: sampling_method 5": dict(method=5), "is: a draw with another n": dict(arrs=short),
}
is_benign = {"is: count = 0": dict(count=0), "is: threads = 0": dict(threads=0), "is: threads = -4": dict(threads=-4)}
for name, kw in {**is_cases, **is_benign}.items():
    rc = is_call(**kw)
    msg = L.bssm_amd_last_error().decode() if rc < 0 else ""
    report[name] = (rc, msg[:90])
    if name in is_cases and not (rc < 0 and msg):
        bad.append((name, rc, msg))
    if name in is_benign and rc < 0:
        bad.append((name, rc, msg))
    if is_call() < 0:
        bad.append(("library unusable after: " + name, 0, ""))

yg = 50.0 + np.cumsum(rng.normal(0, 1, 25))
growth = M.ssm_nlg(yg, "nlg_growth", theta=[0.0, np.log(0.05), 0.0], known_params=[0.1, 500.0, 0.0, 50.0, 1.0, 100.0])
sde = M.ssm_sde(rng.poisson(2.0, 20).astype(float), "sde_poisson_ou", theta=[np.log(0.5), 0.7, np.log(0.4)], x0=0.5, positive=False)

def fn_call(model, Ls=None, nsim=N, mut=None, sde_entry=False, no_out=False):
    s, keep = _lib.pack_fn_model(model, Ls)
    if mut: mut(s)
    mm = 2 if model["example"] == "nlg_growth" else 1
    o, k2 = outputs(max(model.n, 1), mm, max(nsim, 1))
    fn = L.bssm_bsf_sde if sde_entry else L.bssm_bsf_nlg
    return fn(C.byref(s), C.c_uint32(nsim), C.c_uint32(7), None if no_out else C.byref(o))

assert fn_call(growth) >= 0 and fn_call(sde, 2, sde_entry=True) >= 0
def setter(**kw):
    def f(s):
        for k, v in kw.items(): setattr(s, k, v)
    return f
fn_cases = {
    "fn: y NULL": (growth, dict(mut=setter(y=None))), "fn: theta NULL": (growth, dict(mut=setter(theta=None))),
    "fn: n = 0": (growth, dict(mut=setter(n=0))), "fn: n < 0": (growth, dict(mut=setter(n=-3))),
    "fn: model_id 999": (growth, dict(mut=setter(model_id=999))), "fn: model_id -1": (growth, dict(mut=setter(model_id=-1))),
    "fn: theta too short": (growth, dict(mut=setter(n_theta=1))), "fn: known_params too short": (growth, dict(mut=setter(n_known=2))),
    "fn: n_theta = 100": (growth, dict(mut=setter(n_theta=100))),
    "fn: nsim = 0": (growth, dict(nsim=0)), "fn: out NULL": (growth, dict(no_out=True)),
    "fn: sde L = 50": (sde, dict(Ls=50, sde_entry=True)), "fn: sde L < 0": (sde, dict(Ls=-1, sde_entry=True)),
    "fn: nlg model through the sde entry": (growth, dict(sde_entry=True)), "fn: sde model through the nlg entry": (sde, dict(Ls=2)),
    "fn: user model without a library": (growth, dict(mut=setter(model_id=7, user_slot=0))),
}
for name, (mdl, kw) in fn_cases.items():
    rc = fn_call(mdl, **kw)
    msg = L.bssm_amd_last_error().decode() if rc < 0 else ""
    report[name] = (rc, msg[:90])
    if not (rc < 0 and msg):
        bad.append((name, rc, msg))
    if fn_call(growth) < 0:
        bad.append(("library unusable after: " + name, 0, ""))
for k, v in report.items():
    print("%-40s %s" % (k, v))
assert not bad, bad
print("MALFORMED-OK", len(report))
'''


def test_malformed_models_are_refused_and_the_library_survives():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", CHILD], cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "MALFORMED-OK" in r.stdout, (r.returncode, r.stdout[-3000:], r.stderr[-3000:])
