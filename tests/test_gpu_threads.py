"""The C ABI called from several host threads at once.  R calls one entry point at a time, but the library itself runs one host
thread per device shard (`run_multi_device`), a persistent filter thread per post-correction call, and keeps per-process caches
(workspace arena, user-model tables, math tables) -- and a Python or C++ host may well call it from a thread pool.  Every call
here is repeated sequentially first; the concurrent results must be the same bits (each call has its own streams, keyed by
(seed, stream0), and its own output arrays).  ctypes releases the GIL for the duration of a foreign call, so the calls below
really overlap inside the library."""
import threading

import numpy as np
import pytest

from bssm_amd import filters as F
from bssm_amd import models as M
from bssm_amd import post_correct as PC
from models_zoo import mng_zoo, negbin_ar1, negbin_m4_informative, poisson_local_level

pytestmark = pytest.mark.gpu


def _run_threads(jobs):
    """jobs: list of zero-argument callables; returns their results, run on one thread each, started together"""
    out, err = [None] * len(jobs), [None] * len(jobs)
    gate = threading.Barrier(len(jobs))

    def work(i):
        try:
            gate.wait()
            out[i] = jobs[i]()
        except BaseException as e:  # noqa: BLE001 -- re-raised on the main thread
            err[i] = e

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for e in err:
        if e is not None:
            raise e
    return out


def _same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        if isinstance(a[k], np.ndarray):
            assert np.array_equal(a[k], b[k], equal_nan=True), k
        elif isinstance(a[k], float):
            assert a[k] == b[k] or (np.isnan(a[k]) and np.isnan(b[k])), k


def test_filters_of_different_shapes_from_eight_threads():
    """bootstrap filter / psi smoother calls of different models, particle counts and kernel families (workgroup-resident, step
    kernels with one, two and four particles per lane, rows of several series) running at the same time"""
    mng = dict(mng_zoo())
    jobs_spec = [
        (lambda: F.bootstrap_filter(poisson_local_level(60), 100, seed=11, return_indices=True)),
        (lambda: F.bootstrap_filter(negbin_ar1(40, seed=5), 5000, seed=12, return_indices=True)),
        (lambda: F.bootstrap_filter(negbin_m4_informative(30), 20000, seed=13, return_indices=True)),
        (lambda: F.bootstrap_filter(negbin_m4_informative(25), 65536, seed=14, return_indices=True)),
        (lambda: F.particle_smoother(poisson_local_level(50), 300, method="psi", seed=15)),
        (lambda: F.particle_smoother(negbin_ar1(30, seed=6), 9000, method="psi", seed=16)),
        (lambda: F.bootstrap_filter(next(iter(mng.values())), 3000, seed=17, return_indices=True)),
        (lambda: {"ll": F.logLik_batch(negbin_m4_informative(40), 1024, 24, method="bsf", seed=18)}),
    ]
    sequential = [j() for j in jobs_spec]
    for _ in range(3):
        concurrent = _run_threads(jobs_spec)
        for s, c in zip(sequential, concurrent):
            _same(s, c)


def test_the_same_call_from_six_threads_gives_six_times_the_same_bits():
    model = negbin_m4_informative(50)
    ref = F.bootstrap_filter(model, 30000, seed=21, return_indices=True)
    res = _run_threads([lambda: F.bootstrap_filter(model, 30000, seed=21, return_indices=True)] * 6)
    for r in res:
        _same(ref, r)


def test_post_correction_calls_overlap(oracle):
    """two IS post-corrections (each with its own filter thread and set-up threads inside the library) and a plain filter"""
    rng = np.random.default_rng(3)
    n, ns = 40, 37
    y = rng.poisson(4.0, n).astype(float)
    model = M.bsm_ng(y, sd_level=M.halfnormal(0.2, 1), sd_slope=M.halfnormal(0.02, 1), distribution="poisson", P1=np.diag([1.0, 0.1]))
    theta = np.log(np.column_stack([rng.uniform(0.1, 0.4, ns), rng.uniform(0.01, 0.05, ns)]))
    counts = rng.integers(1, 4, ns).astype(np.uint32)

    modes = np.stack([oracle.psi_setup(a)["mode"] for a in PC.expand_models(model, theta)], axis=1)  # (the checker supplies inputs only)

    def pc(seed, particles, output_type):
        return lambda: PC.is_correction(model, theta, counts, particles, method="psi", modes=modes, seed=seed, threads=2,
                                        output_type=output_type)

    jobs = [pc(31, 64, 3), pc(32, 256, 2), lambda: F.bootstrap_filter(negbin_ar1(20, seed=9), 4096, seed=33, return_indices=True)]
    sequential = [j() for j in jobs]
    for _ in range(2):
        concurrent = _run_threads(jobs)
        for s, c in zip(sequential, concurrent):
            _same(s, c)


def test_functor_models_plans_and_nlg_post_correction_overlap(oracle):
    """ssm_nlg / ssm_sde filters (bootstrap, psi-APF, EKF particle filter), a device-resident plan and the nlg IS correction
    with its device set-up kernel, all at once"""
    import torch

    from bssm_amd.plan import FilterPlan
    from test_gpu_functors import MODELS

    growth, _ = MODELS["nlg_growth"]
    sde, L = MODELS["sde_poisson_ou"]
    rng = np.random.default_rng(2)
    ns = 40
    theta = np.array([0.0, np.log(0.05), 0.0]) + 0.05 * rng.normal(size=(ns, 3))
    counts = rng.integers(1, 4, ns).astype(np.uint32)
    modes = np.stack([oracle.fn_psi_setup({**growth, "theta": th})["mode"].reshape(-1, order="F") for th in theta], axis=1)

    def plan_job():
        st = torch.cuda.Stream()
        p = FilterPlan(negbin_m4_informative(30), 8192, method="bsf", n_rep=6, keep_history=True)
        p.run(5, hip_stream=st.cuda_stream)
        out = {"ll": p.loglik(hip_stream=st.cuda_stream), **p.fetch(3, hip_stream=st.cuda_stream)}
        p.close()
        return out

    jobs = [
        lambda: F.bootstrap_filter(growth, 2000, seed=41, return_indices=True),
        lambda: F.particle_smoother(growth, 4096, method="psi", seed=42),
        lambda: F.ekpf_filter(growth, 500, seed=43, return_indices=True),
        lambda: F.bootstrap_filter(sde, 1000, seed=44, return_indices=True, L=L),
        lambda: PC.is_correction(growth, theta, counts, 256, method="psi", modes=modes, seed=45, threads=2, output_type=2),
        lambda: PC.is_correction(growth, theta, counts, 4096, method="psi", modes=modes, seed=46, threads=1, output_type=3),
        plan_job,
    ]
    sequential = [j() for j in jobs]
    for _ in range(3):
        concurrent = _run_threads(jobs)
        for s, c in zip(sequential, concurrent):
            _same(s, c)


@pytest.mark.parametrize("order", ["library_first", "torch_first"])
def test_library_and_torch_share_one_hip_runtime_in_either_import_order(order):
    """PyTorch-ROCm ships its own HIP / HSA runtime.  A process that initialised the system runtime through the library first
    used to lose torch ("No HIP GPUs are available"): bssm_amd/_lib.py now loads torch's runtime first wherever torch is
    installed, so one runtime serves both whichever is used first (fresh process per order)."""
    import os
    import subprocess
    import sys

    code = r'''
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
def use_lib():
    from bssm_amd import filters as F
    from models_zoo import poisson_local_level
    return F.bootstrap_filter(poisson_local_level(20), 100, seed=1)["logLik"]
def use_torch():
    import torch
    assert torch.cuda.is_available()
    return float((torch.ones(4, device="cuda") * 2).sum().item())
steps = [use_lib, use_torch, use_lib] if sys.argv[1] == "library_first" else [use_torch, use_lib, use_torch]
vals = [s() for s in steps]
runtimes = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l})
assert len(runtimes) == 1, runtimes
print("OK", vals)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code, order], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
