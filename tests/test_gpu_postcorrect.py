"""GPU parity of the IS post-correction (one psi-APF / bootstrap filter per retained theta-draw) against the
oracle's restatement of approx_mcmc::is_correction_psi / _bsf (src/approx_mcmc.cpp:353-427, 447-518)."""
import numpy as np
import pytest

from bssm_amd import models as M
from bssm_amd import post_correct as PC

pytestmark = pytest.mark.gpu


def _job(ns=9, n=40, seed=4):
    rng = np.random.default_rng(seed)
    y = rng.poisson(4.0, n).astype(float); y[5] = np.nan
    model = M.bsm_ng(y, sd_level=M.halfnormal(0.2, 1), sd_slope=M.halfnormal(0.02, 1), distribution="poisson",
                     P1=np.diag([1.0, 0.1]))
    theta = np.log(np.column_stack([rng.uniform(0.1, 0.4, ns), rng.uniform(0.01, 0.05, ns)]))
    counts = rng.integers(1, 5, ns).astype(np.uint32)
    return model, theta, counts


@pytest.mark.parametrize("output_type", [1, 2, 3])
@pytest.mark.parametrize("is_type", ["is1", "is2", "is3"])
def test_post_correct_matches_oracle(oracle, is_type, output_type):
    model, theta, counts = _job()
    arrs = PC.expand_models(model, theta)
    modes = np.stack([oracle.psi_setup(a)["mode"] for a in arrs], axis=1)
    mo = dict(theta=theta, counts=counts, modes=modes, posterior=np.full(len(counts), -3.0), output_type=output_type)
    out = PC.post_correct(model, mo, 16, threads=2, is_type=is_type, seed=9)
    it = int(is_type[-1])
    if it == 3:
        rep = counts.astype(int)
        arrs_e = [a for a, r in zip(arrs, rep) for _ in range(r)]
        modes_e, counts_e = np.repeat(modes, rep, axis=1), np.ones(rep.sum(), dtype=np.uint32)
    else:
        arrs_e, modes_e, counts_e = arrs, modes, counts
    ref = oracle.is_correction(arrs_e, 16, 9, method=1, counts=counts_e, modes=modes_e, is_type=it,
                               output_type=output_type, n_threads=2)
    n = model.n
    assert np.allclose(out["loglik"], ref["loglik"], rtol=0, atol=1e-9 * n)
    assert np.allclose(out["weights"], ref["weights"], rtol=1e-8)
    assert np.allclose(out["posterior"], (-3.0 if it != 3 else -3.0) + np.log(ref["weights"]), rtol=0, atol=1e-8)
    if output_type == 1:
        assert np.allclose(out["alpha"], ref["alpha"], rtol=1e-8, atol=1e-10)
    if output_type == 2:
        assert np.allclose(out["alphahat"], ref["alphahat"].T, rtol=1e-8, atol=1e-10)
        assert np.allclose(out["Vt"], ref["Vt"], rtol=1e-7, atol=1e-10)


def test_bsf_is_correction_matches_oracle(oracle):
    model, theta, counts = _job(ns=6)
    arrs = PC.expand_models(model, theta)
    al = np.array([oracle.psi_setup(a)["approx_loglik"] for a in arrs])
    res = PC.is_correction(model, theta, counts, 64, method="bsf", approx_loglik=al, output_type=2, seed=3)
    ref = oracle.is_correction(arrs, 64, 3, method=2, counts=counts, approx_loglik=al, output_type=2)
    assert np.allclose(res["loglik"], ref["loglik"], rtol=0, atol=1e-9 * model.n)
    assert np.allclose(res["weights"], ref["weights"], rtol=1e-8)
    assert np.allclose(res["alphahat"], ref["alphahat"], rtol=1e-8, atol=1e-10)
    assert np.allclose(res["Vt"], ref["Vt"], rtol=1e-7, atol=1e-10)


def test_suggest_N(oracle):
    """tests/testthat/test_post_correct.R: suggest_N returns the first candidate with sd < 1"""
    model, theta, counts = _job()
    res = PC.suggest_N(model, theta[0], candidates=[5, 10, 20], replications=40, seed=1)
    assert res["results"]["sd"].shape == (3,) and np.all(np.isfinite(res["results"]["sd"]))
    assert res["N"] in (5, 10, 20)
    # the batch equals individual logLik calls with the same streams
    from bssm_amd.filters import logLik_batch
    arr = PC.expand_models(model, theta[:1])[0]
    ll = oracle.psi_filter(arr, 5, seed=1, stream=3)["logLik"]
    mdl = M.BssmModel(model.classes, **{**model, **{k: arr[k] for k in ("T", "R", "P1", "a1", "C")}})
    got = logLik_batch(mdl, 5, 40, method="psi", seed=1, stream0=0)
    assert abs(got[3] - ll) <= 1e-9 * model.n


@pytest.mark.parametrize("kind", ["bsm_ng", "nlg_growth"])
def test_device_psi_setup_equals_host_setup(monkeypatch, kind):
    """The batched device smoother of the psi set-up (lg_setup_kernel, one lane per draw) and the
    host smoother are the same template (lg_smoother.h): identical ancestors, log-likelihoods to 1e-9 n."""
    if kind == "bsm_ng":
        from oracle import oracle as O
        model, theta, counts = _job(ns=40, n=60)
        arrs = PC.expand_models(model, theta)
        modes = np.stack([O.psi_setup(a)["mode"] for a in arrs], axis=1)
        n = model.n
    else:
        from bssm_amd import filters as F
        rng = np.random.default_rng(12)
        n, K, dT = 80, 500.0, 0.1
        r = np.zeros(n); p = np.zeros(n); p[0] = 50.0
        for t in range(1, n):
            r[t] = r[t - 1] + rng.normal(0, 0.05)
            rr = np.exp(r[t - 1]) / (1 + np.exp(r[t - 1]))
            p[t] = K * p[t - 1] * np.exp(rr * dT) / (K + p[t - 1] * (np.exp(rr * dT) - 1)) + rng.normal(0, 1)
        y = p + rng.normal(0, 1, n); y[[7, 8]] = np.nan
        th0 = np.array([0.0, np.log(0.05), 0.0])
        model = M.ssm_nlg(y, "nlg_growth", theta=th0, known_params=[dT, K, 0.0, 50.0, 1.0, 100.0])
        theta = th0 + rng.normal(0, 0.05, (40, 3))
        counts = rng.integers(1, 4, 40).astype(np.uint32)
        mode = F.psi_setup(model)["mode"]
        modes = np.repeat(mode.reshape(-1, 1, order="F"), 40, axis=1).reshape(2, n, 40, order="F")
    res = {}
    # "0": everything on the host; "1": the smoother as a device batch; "2": linearisation + smoother + tables on the
    # device (ssm_nlg built-ins; the default there) -- for bsm_ng "2" is the same as unset
    for dev in ("2", "1", "0"):
        monkeypatch.setenv("BSSM_AMD_DEVICE_SETUP", dev)
        mo = dict(theta=theta, counts=counts, modes=modes, output_type=2)
        res[dev] = PC.post_correct(model, mo, 64, threads=3, is_type="is2", seed=4)
    for dev in ("2", "1"):
        assert np.all(np.isfinite(res[dev]["loglik"]))
        assert np.allclose(res[dev]["loglik"], res["0"]["loglik"], rtol=0, atol=1e-9 * n)
        assert np.allclose(res[dev]["alphahat"], res["0"]["alphahat"], rtol=1e-8, atol=1e-10)
        assert np.allclose(res[dev]["Vt"], res["0"]["Vt"], rtol=1e-7, atol=1e-10)
    # ... and against the ORACLE's restatement of the same set-up + filter (not only against the product's host path)
    from oracle import oracle as O
    if kind == "bsm_ng":
        ref = O.is_correction(arrs, 64, 4, method=1, counts=counts, modes=modes, is_type=2, output_type=2, n_threads=2)
        for dev in ("2", "1", "0"):
            assert np.allclose(res[dev]["loglik"], ref["loglik"], rtol=0, atol=1e-9 * n)
            assert np.allclose(res[dev]["alphahat"], ref["alphahat"].T, rtol=1e-8, atol=1e-10)
    else:
        lw = O.fn_is_logweights(model, theta, modes.reshape(2 * n, 40, order="F"), 64, seed=4, first_stream=0, n_threads=2)
        for dev in ("2", "1", "0"):
            assert np.allclose(np.log(res[dev]["weights"]), lw, rtol=0, atol=1e-9 * n)


def test_device_set_up_marks_failed_draws(monkeypatch):
    """a draw whose smoother fails (non-finite factors) gets weight 0 / logLik -Inf on every set-up path, the others are
    unaffected (src/model_ssm_nlg.cpp:988-990)"""
    from bssm_amd import filters as F
    rng = np.random.default_rng(3)
    n = 40
    y = 50 + np.cumsum(rng.normal(0.5, 1, n))
    th0 = np.array([0.0, np.log(0.05), 0.0])
    model = M.ssm_nlg(y, "nlg_growth", theta=th0, known_params=[0.1, 500.0, 0.0, 50.0, 1.0, 100.0])
    theta = th0 + rng.normal(0, 0.05, (6, 3))
    theta[2, 0] = 800.0  # exp(800) overflows: H = Inf
    counts = np.ones(6, dtype=np.uint32)
    mode = F.psi_setup(model)["mode"]
    modes = np.repeat(mode.reshape(-1, 1, order="F"), 6, axis=1).reshape(2, n, 6, order="F")
    out = {}
    for dev in ("2", "0"):
        monkeypatch.setenv("BSSM_AMD_DEVICE_SETUP", dev)
        out[dev] = PC.post_correct(model, dict(theta=theta, counts=counts, modes=modes, output_type=3), 32, is_type="is2", seed=2)
        assert out[dev]["weights"][2] == 0.0 and np.isneginf(out[dev]["loglik"][2])
        assert np.all(np.isfinite(np.delete(out[dev]["loglik"], 2)))
    assert np.allclose(np.delete(out["2"]["loglik"], 2), np.delete(out["0"]["loglik"], 2), rtol=0, atol=1e-9 * n)


@pytest.mark.parametrize("output_type", [1, 2])
def test_post_correct_is1_large_particle_counts(oracle, output_type):
    """is1 multiplies the particle count by the jump-chain count (src/approx_mcmc.cpp:376-378): mixed particle counts in
    one job, some beyond the workgroup-resident capacity, so both kernel shapes serve one call"""
    model, theta, counts = _job(ns=6, n=25, seed=14)
    counts = np.array([1, 4, 2, 1, 4, 3], dtype=np.uint32)
    arrs = PC.expand_models(model, theta)
    modes = np.stack([oracle.psi_setup(a)["mode"] for a in arrs], axis=1)
    mo = dict(theta=theta, counts=counts, modes=modes, posterior=np.zeros(6), output_type=output_type)
    nsim = 3000  # x4 = 12000 particles > 8192 (capacity of the LDS-resident shape for m <= 2)
    out = PC.post_correct(model, mo, nsim, threads=2, is_type="is1", seed=21)
    ref = oracle.is_correction(arrs, nsim, 21, method=1, counts=counts, modes=modes, is_type=1, output_type=output_type, n_threads=2)
    assert np.allclose(out["loglik"], ref["loglik"], rtol=0, atol=1e-8 * model.n)
    assert np.allclose(out["weights"], ref["weights"], rtol=1e-7)
    if output_type == 1:
        assert np.allclose(out["alpha"], ref["alpha"], rtol=1e-7, atol=1e-9)
    else:
        assert np.allclose(out["alphahat"], ref["alphahat"].T, rtol=1e-7, atol=1e-9)
        assert np.allclose(out["Vt"], ref["Vt"], rtol=1e-6, atol=1e-9)


def test_sde_bootstrap_is_correction_matches_oracle(oracle):
    """The ssm_sde state sampler / IS correction (src/R_sde_state_sampler.cpp:10-50, approx_mcmc::is_correction_bsf for
    ssm_sde): one bootstrap filter with 2^L Milstein sub-steps per stored theta, weight exp(loglik - approx_loglik_i), one
    back-traced trajectory picked uniformly per draw"""
    import ctypes as C
    rng = np.random.default_rng(3)
    x = np.exp(np.cumsum(rng.normal(0.01, 0.1, 20)))
    y = np.log(x) + rng.normal(0, 0.5, 20)
    model = M.ssm_sde(y, "sde_gbm", theta=[0.05, 0.3, 1.0], x0=1.0, positive=True)
    model = M.BssmModel(model.classes, **{**model, "L": 3})
    ns, N, seed = 7, 200, 11
    theta = np.column_stack([rng.uniform(0.02, 0.08, ns), rng.uniform(0.2, 0.4, ns), rng.uniform(0.8, 1.2, ns)])
    counts = rng.integers(1, 4, ns).astype(np.uint32)
    al = rng.normal(-25, 0.3, ns)
    res = PC.is_correction(model, theta, counts, N, method="bsf", approx_loglik=al, output_type=1, seed=seed)
    for i in range(ns):
        ref = oracle.fn_bsf_filter({**model, "theta": theta[i], "L": 3}, N, seed=seed, stream=i)
        assert abs(res["loglik"][i] - ref["logLik"]) <= 1e-9 * model.n
        assert abs(res["weights"][i] - np.exp(ref["logLik"] - al[i])) <= 1e-8 * res["weights"][i]
        sm = oracle.filter_smoother(ref["alpha"], ref["indices"])
        r, _ = oracle.make_rng(seed=seed, stream=i)
        pick = min(int(oracle.lib().orc_rng_uniform(C.byref(r), 3, 0, 0, 0) * N), N - 1)
        assert np.allclose(res["alpha"][:, 0, i], sm[0, :, pick], rtol=1e-9, atol=1e-12)


def test_suggest_N_shard_unit_matches_single_call():
    """The unit the multi-GPU suggest_N shards (`_loglik_replications`: streams candidate * replications + j) reproduces the
    one-call standard deviations of bssm_suggest_n_nongaussian, in any split of the replications"""
    model, theta, counts = _job()
    cands, R, seed = [5, 10, 20], 24, 6
    res = PC.suggest_N(model, theta[0], candidates=cands, replications=R, seed=seed)
    arr = PC.expand_models(model, theta[:1])[0]
    mdl = M.BssmModel(model.classes, **{**model, **{k: arr[k] for k in ("Z", "T", "R", "P1", "a1", "C", "D", "beta")}, "phi": arr["phi"]})
    for i, c in enumerate(cands):
        parts = [PC._loglik_replications(mdl, c, hi - lo, seed, i * R + lo) for lo, hi in ((0, 7), (7, 16), (16, 24))]
        assert abs(np.concatenate(parts).std(ddof=1) - res["results"]["sd"][i]) <= 1e-12


# ------------------------------------------------------------------ one call, several GPUs (inside the library)
@pytest.mark.parametrize("is_type,output_type", [("is2", 3), ("is1", 2), ("is2", 1), ("is3", 2)])
def test_multi_device_call_is_bit_identical_to_single_block(is_type, output_type):
    """bssm_is_correction_multi (host thread + hipSetDevice per shard, weights exchanged on the device) with two and
    three logical shards on the one GPU of the test box: every output bit-identical to the single-device call (streams
    are keyed by the global draw index; src/approx_mcmc.cpp:364-369 spreads the same loop over OpenMP threads)"""
    from oracle import oracle as O
    model, theta, counts = _job(ns=23, n=30, seed=8)
    arrs = PC.expand_models(model, theta)
    modes = np.stack([O.psi_setup(a)["mode"] for a in arrs], axis=1)
    mo = dict(theta=theta, counts=counts, modes=modes, posterior=np.zeros(len(counts)), output_type=output_type)
    one = PC.post_correct(model, mo, 32, threads=2, is_type=is_type, seed=5)
    for devs in ([0, 0], [0, 0, 0]):
        many = PC.post_correct(model, mo, 32, threads=4, is_type=is_type, seed=5, devices=devs)
        sh = many["timing_ms"]["shards"]
        assert len(sh) == len(devs) and sum(s["count"] for s in sh) == len(one["weights"])
        assert many["timing_ms"]["gather"] == "device_copies"
        for key in ("weights", "loglik", "posterior"):
            assert np.array_equal(one[key], many[key]), key
        if output_type == 1:
            assert np.array_equal(one["alpha"], many["alpha"])
        if output_type == 2:
            assert np.array_equal(one["alphahat"], many["alphahat"]) and np.array_equal(one["Vt"], many["Vt"])


def test_multi_device_nlg_and_rccl_all_gather():
    """ssm_nlg draws through bssm_is_correction_nlg_multi; with distinct devices (here the one visible GPU, a
    communicator of size one) the exchange really goes through ncclCommInitAll + ncclAllGather of the dlopen'ed RCCL"""
    from bssm_amd import _lib, filters as F
    rng = np.random.default_rng(2)
    n, K, dT = 50, 500.0, 0.1
    p = 50.0 + np.cumsum(rng.normal(0.5, 1.0, n))
    y = p + rng.normal(0, 1, n)
    th0 = np.array([0.0, np.log(0.05), 0.0])
    model = M.ssm_nlg(y, "nlg_growth", theta=th0, known_params=[dT, K, 0.0, 50.0, 1.0, 100.0])
    ns = 17
    theta = th0 + rng.normal(0, 0.05, (ns, 3))
    counts = rng.integers(1, 4, ns).astype(np.uint32)
    mode = F.psi_setup(model)["mode"]
    modes = np.repeat(mode.reshape(-1, 1, order="F"), ns, axis=1).reshape(2, n, ns, order="F")
    mo = dict(theta=theta, counts=counts, modes=modes, output_type=2)
    one = PC.post_correct(model, mo, 64, threads=2, is_type="is2", seed=4)
    two = PC.post_correct(model, mo, 64, threads=2, is_type="is2", seed=4, devices=[0, 0])
    assert np.array_equal(one["weights"], two["weights"]) and np.array_equal(one["alphahat"], two["alphahat"])
    assert _lib.lib().bssm_amd_rccl_available() == 1, _lib.lib().bssm_amd_last_error()
    rc = PC.post_correct(model, mo, 64, threads=2, is_type="is2", seed=4, devices="all")
    assert rc["timing_ms"]["gather"] == "rccl_all_gather"
    assert np.array_equal(one["weights"], rc["weights"]) and np.array_equal(one["loglik"], rc["loglik"])
    assert np.array_equal(one["Vt"], rc["Vt"])


def test_multi_device_suggest_N_is_bit_identical(monkeypatch):
    """bssm_suggest_n_{nongaussian,nonlinear}_multi: the replications of every candidate cut into blocks over two / three
    logical shards on the one GPU (device-copy exchange), over the one visible GPU as a communicator of size one (RCCL
    all-gather), and with the exchange forced to the host blocks -- log-likelihoods and standard deviations bit-identical
    to the single-device entry point (src/R_postcorrection.cpp:24-35, 118-129)"""
    from bssm_amd import filters as F
    model, theta, counts = _job()
    rng = np.random.default_rng(2)
    n = 40
    y = 50.0 + np.cumsum(rng.normal(0.5, 1.0, n)) + rng.normal(0, 1, n)
    th0 = np.array([0.0, np.log(0.05), 0.0])
    growth = M.ssm_nlg(y, "nlg_growth", theta=th0, known_params=[0.1, 500.0, 0.0, 50.0, 1.0, 100.0])
    for mdl, th in ((model, theta[0]), (growth, th0)):
        cands, R, seed = [5, 12, 30], 23, 6
        one = PC.suggest_N(mdl, th, candidates=cands, replications=R, seed=seed)
        for devs, kind in (([0, 0], "device_copies"), ([0, 0, 0], "device_copies"), ("all", "rccl_all_gather")):
            many = PC.suggest_N(mdl, th, candidates=cands, replications=R, seed=seed, devices=devs, return_loglik=True)
            assert many["gather"] == kind
            assert np.array_equal(one["results"]["sd"], many["results"]["sd"]) and one["N"] == many["N"]
            assert many["loglik"].shape == (3, R) and np.all(np.isfinite(many["loglik"]))
            assert np.array_equal(many["loglik"].std(axis=1, ddof=1) <= 1e300, np.ones(3, dtype=bool))
        monkeypatch.setenv("BSSM_AMD_GATHER", "host")
        host = PC.suggest_N(mdl, th, candidates=cands, replications=R, seed=seed, devices=[0, 0])
        monkeypatch.delenv("BSSM_AMD_GATHER")
        assert host["gather"] == "host_blocks" and np.array_equal(one["results"]["sd"], host["results"]["sd"])


def test_multi_device_exchange_falls_back_instead_of_failing(monkeypatch):
    """RCCL is an optimisation: with the exchange forced off the device (BSSM_AMD_GATHER=host) a multi-device post-correction
    returns the same weights from the shards' host blocks"""
    from oracle import oracle as O
    model, theta, counts = _job(ns=11, n=20, seed=3)
    arrs = PC.expand_models(model, theta)
    modes = np.stack([O.psi_setup(a)["mode"] for a in arrs], axis=1)
    mo = dict(theta=theta, counts=counts, modes=modes, posterior=np.zeros(len(counts)), output_type=3)
    one = PC.post_correct(model, mo, 32, threads=2, is_type="is2", seed=5)
    monkeypatch.setenv("BSSM_AMD_GATHER", "host")
    two = PC.post_correct(model, mo, 32, threads=2, is_type="is2", seed=5, devices=[0, 0])
    assert two["timing_ms"]["gather"] == "host_blocks"
    assert np.array_equal(one["weights"], two["weights"]) and np.array_equal(one["loglik"], two["loglik"])


def _ung_zoo(kind, rng, n=50):
    """(model, theta draws) per model family / distribution for the full-device psi set-up (ung_setup_kernel)"""
    ns = 70  # more than one wave of draws
    if kind == "svm":
        y = rng.normal(0, 1, n) * np.exp(0.5 * np.cumsum(rng.normal(0, 0.2, n))); y[3] = np.nan; y[10] = 0.0
        model = M.svm(y, rho=M.uniform(0.9, -0.99, 0.99), sd_ar=M.halfnormal(0.2, 5), sigma=M.halfnormal(1.0, 2))
        theta = np.column_stack([rng.uniform(0.8, 0.97, ns), rng.uniform(0.1, 0.4, ns), rng.uniform(0.7, 1.3, ns)])
    elif kind == "ar1_negbin_xreg":
        x = rng.normal(0, 1, (n, 2))
        y = rng.negative_binomial(5, 0.5, n).astype(float); y[[0, 17]] = np.nan
        model = M.ar1_ng(y, rho=M.uniform(0.7, -0.99, 0.99), sigma=M.halfnormal(0.3, 2), mu=M.normal(1.0, 0, 5),
                         distribution="negative binomial", phi=M.gamma(4.0, 2, 0.5), u=rng.uniform(0.5, 2.0, n),
                         beta=[M.normal(0.1, 0, 1), M.normal(-0.2, 0, 1)], xreg=x)
        theta = np.column_stack([rng.uniform(0.5, 0.9, ns), rng.uniform(0.2, 0.5, ns), rng.normal(1.0, 0.1, ns),
                                 rng.uniform(3, 6, ns), rng.normal(0.1, 0.05, ns), rng.normal(-0.2, 0.05, ns)])
    elif kind == "bsm_gamma_seasonal":  # m = 2 + 3 = 5 states: padded to 8
        y = rng.gamma(3.0, 2.0, n); y[5] = np.nan
        model = M.bsm_ng(y, sd_level=M.halfnormal(0.1, 1), sd_slope=M.halfnormal(0.02, 1), sd_seasonal=M.halfnormal(0.05, 1),
                         period=4, distribution="gamma", phi=M.gamma(3.0, 2, 0.5), u=rng.uniform(0.8, 1.2, n),
                         P1=np.diag([1.0, 0.1, 0.5, 0.5, 0.5]), a1=np.array([1.5, 0, 0, 0, 0]))
        theta = np.column_stack([rng.uniform(0.05, 0.2, ns), rng.uniform(0.01, 0.03, ns), rng.uniform(0.02, 0.08, ns), rng.uniform(2, 4, ns)])
    elif kind == "bsm_poisson_weekly":  # level + slope + six seasonal states: m = 8 -> the row-distributed smoother (lg_smoother_rows.h)
        y = rng.poisson(6.0, n).astype(float); y[[4, 21]] = np.nan
        model = M.bsm_ng(y, sd_level=M.halfnormal(0.1, 1), sd_slope=M.halfnormal(0.02, 1), sd_seasonal=M.halfnormal(0.05, 1),
                         period=7, distribution="poisson", u=rng.uniform(0.8, 1.2, n),
                         P1=np.diag([1.0, 0.1] + [0.5] * 6), a1=np.array([1.7] + [0.0] * 7))
        theta = np.column_stack([rng.uniform(0.05, 0.2, ns), rng.uniform(0.01, 0.03, ns), rng.uniform(0.02, 0.08, ns)])
    elif kind == "bsm_poisson_monthly":  # level + slope + eleven seasonal states: m = 13, packed padded to 16 (round 5: ung_setup_kernel<16>)
        tt = np.arange(n)
        y = rng.poisson(np.exp(1.6 + 0.4 * np.sin(2 * np.pi * tt / 12.0))).astype(float); y[[4, 21]] = np.nan
        model = M.bsm_ng(y, sd_level=M.halfnormal(0.1, 1), sd_slope=M.halfnormal(0.02, 1), sd_seasonal=M.halfnormal(0.05, 1),
                         period=12, distribution="poisson", u=rng.uniform(0.8, 1.2, n),
                         P1=np.diag([1.0, 0.1] + [0.5] * 11), a1=np.array([1.6] + [0.0] * 12))
        theta = np.column_stack([rng.uniform(0.05, 0.2, ns), rng.uniform(0.01, 0.03, ns), rng.uniform(0.02, 0.08, ns)])
    elif kind in ("bsm_poisson_hourly", "bsm_poisson_weekofyear"):
        # level + slope + 23 / 51 seasonal states: m = 25 / 53, packed padded to 32 / 64 -- the smoother with its matrices in LDS
        # (lg_smoother_wave.h), the conditional factors row-distributed (lg_factors_rows.h); fewer draws: the oracle's set-up is O(m^3) per time point
        period = 24 if kind == "bsm_poisson_hourly" else 52
        ns = 20 if period == 24 else 5
        if period == 52:
            n = 40
        tt = np.arange(n)
        y = rng.poisson(np.exp(1.6 + 0.4 * np.sin(2 * np.pi * tt / period))).astype(float); y[[4, 21]] = np.nan
        model = M.bsm_ng(y, sd_level=M.halfnormal(0.1, 1), sd_slope=M.halfnormal(0.02, 1), sd_seasonal=M.halfnormal(0.05, 1),
                         period=period, distribution="poisson", u=rng.uniform(0.8, 1.2, n),
                         P1=np.diag([1.0, 0.1] + list(0.5 + 0.01 * np.arange(period - 1))), a1=np.array([1.6] + [0.0] * period))
        # (distinct prior variances: a series shorter than the period leaves seasonal states without an observation, and EQUAL variances there
        # are a repeated eigenvalue of the smoothed initial covariance -- any basis of its eigenspace is a valid factor, two programs need not pick the same)
        theta = np.column_stack([rng.uniform(0.05, 0.2, ns), rng.uniform(0.01, 0.03, ns), rng.uniform(0.02, 0.08, ns)])
    else:  # bsm_binomial, local level + noise: m = 2... with sd_noise the state is (level, noise)
        u = rng.integers(5, 20, n).astype(float)
        y = rng.binomial(u.astype(int), 0.4).astype(float); y[8] = np.nan
        model = M.bsm_ng(y, sd_level=M.halfnormal(0.2, 1), sd_noise=M.halfnormal(0.1, 1), distribution="binomial", u=u)
        theta = np.column_stack([rng.uniform(0.1, 0.3, ns), rng.uniform(0.05, 0.2, ns)])
    counts = rng.integers(1, 4, ns).astype(np.uint32)
    return model, theta, counts


@pytest.mark.parametrize("kind", ["svm", "ar1_negbin_xreg", "bsm_gamma_seasonal", "bsm_poisson_weekly", "bsm_poisson_monthly", "bsm_poisson_hourly",
                                  "bsm_poisson_weekofyear", "bsm_binomial"])
@pytest.mark.parametrize("output_type", [2, 3])
def test_full_device_ung_setup_against_oracle(oracle, monkeypatch, kind, output_type):
    """SURVEY.md 8f rank 1 for ssm_ung draws: the Laplace step at the stored mode, the scales, the smoother and the
    filter's tables are built by ung_setup_kernel (the default); compared with the ORACLE's restatement
    (src/model_ssm_ung.cpp:143-161, 241-342, 511-609) and with the product's host set-up."""
    rng = np.random.default_rng(31)
    model, theta, counts = _ung_zoo(kind, rng)
    n = model.n
    big = ("bsm_poisson_monthly", "bsm_poisson_hourly", "bsm_poisson_weekofyear")  # states padded to 16 / 32 / 64: see the tolerances below
    arrs = PC.expand_models(model, theta)
    modes = np.stack([oracle.psi_setup(a)["mode"] for a in arrs], axis=1)
    # perturb the stored modes: approximate_for_is takes whatever the chain stored, not the converged mode of this theta
    modes = modes + rng.normal(0, 0.05, modes.shape)
    mo = dict(theta=theta, counts=counts, modes=modes, output_type=output_type)
    res = {}
    for dev in ("2", "0"):
        monkeypatch.setenv("BSSM_AMD_DEVICE_SETUP", dev)
        res[dev] = PC.post_correct(model, mo, 32, threads=2, is_type="is2", seed=6)
    monkeypatch.delenv("BSSM_AMD_DEVICE_SETUP")
    res["default"] = PC.post_correct(model, mo, 32, threads=2, is_type="is2", seed=6)
    assert np.array_equal(res["default"]["loglik"], res["2"]["loglik"])  # the full-device set-up is the default
    ref = oracle.is_correction(arrs, 32, 6, method=1, counts=counts, modes=modes, is_type=2, output_type=output_type, n_threads=2)
    for dev in ("2", "0"):
        assert np.all(np.isfinite(res[dev]["loglik"]))
        # (thirteen states, eleven of them a seasonal pattern with singular smoothing covariances: the device's FMA contraction and table-driven
        # exp / log reach 6e-8 of a log-likelihood through the factors; the host set-up agrees with the oracle to 1e-14)
        assert np.allclose(res[dev]["loglik"], ref["loglik"], rtol=0, atol=1e-9 * n)
        # (eight states with singular smoothing covariances: the same last-ulp amplification as below reaches 1e-8 of the log-weights)
        assert np.allclose(res[dev]["weights"], ref["weights"], rtol=1e-7 if kind in big else 1e-8)
        if output_type == 2:
            # (the seasonal model's smoothing covariances are singular: the factors amplify last-ulp differences between the
            # oracle's generic matrix code and the fixed-size template to ~1e-8 of the state scale)
            # (thirteen states: 1e-7 of the state scale -- the device-vs-host differences of the log-likelihoods are of the same size at m = 8
            # and m = 13, median 2e-8 / 3e-8, profiles/r05_seasonal_setup.txt)
            # (53 states padded to 64: the factors of 50 exactly singular directions per time point -- 2e-6 of the state scale, scripts/lgwave_check.hip)
            # (round 6: the null directions of the singular conditional covariances are exact zeros in every implementation -- oracle/bssm_oracle.c,
            # orc_conditional_cov -- and the amplification described above is gone: one bound for every model)
            at = 1e-8 if (dev == "2" and kind in big) else 1e-9   # (16 and more padded states on the device: pinv(V) C from a pivoted Cholesky factor, cond(V) eps away from the decomposition's)
            assert np.allclose(res[dev]["alphahat"], ref["alphahat"].T, rtol=1e-7, atol=at)
            assert np.allclose(res[dev]["Vt"], ref["Vt"], rtol=1e-6, atol=at)
    # device set-up vs host set-up: the same template and operation order (the device contracts to FMA, its exp / log differ
    # from libm's in the last place, and near-singular factors amplify that: 1e-8 of the state scale at most)
    # (the weekly-seasonal model: eight states, singular smoothing covariances -- the amplification reaches 2e-8 of the
    # log-likelihood; the one-lane and the row-distributed device kernels agree bit for bit there, the host agrees with the oracle)
    assert np.allclose(res["2"]["loglik"], res["0"]["loglik"], rtol=0, atol=(1e-9 if kind in big else 1e-10) * n)
    if output_type == 2:
        assert np.allclose(res["2"]["alphahat"], res["0"]["alphahat"], rtol=1e-7, atol=1e-8 if kind in big else 1e-9)


@pytest.mark.parametrize("kind", ["bsm_ng", "nlg_growth"])
def test_summary_behind_the_next_chunk_equals_the_synchronous_path(monkeypatch, kind):
    """output 2 (smoothed summaries): the fused back-trace + summary kernel of chunk c runs on its own stream beside the
    filters of chunk c + 1 (PendingSummary, opt-in: BSSM_AMD_DEFER_SUMMARY; measured neutral on C5, so not the default)
    -- several chunks per call here -- and must give exactly what the synchronous path gives, chunk boundaries
    notwithstanding"""
    from bssm_amd import filters as F
    if kind == "bsm_ng":
        from oracle import oracle as O
        model, theta, counts = _job(ns=37, n=30, seed=4)
        arrs = PC.expand_models(model, theta)
        modes = np.stack([O.psi_setup(a)["mode"] for a in arrs], axis=1)
    else:
        rng = np.random.default_rng(12)
        n = 40
        y = 50.0 + np.cumsum(rng.normal(0.5, 1.0, n)) + rng.normal(0, 1, n)
        th0 = np.array([0.0, np.log(0.05), 0.0])
        model = M.ssm_nlg(y, "nlg_growth", theta=th0, known_params=[0.1, 500.0, 0.0, 50.0, 1.0, 100.0])
        theta = th0 + rng.normal(0, 0.05, (37, 3))
        counts = rng.integers(1, 4, 37).astype(np.uint32)
        mode = F.psi_setup(model)["mode"]
        modes = np.repeat(mode.reshape(-1, 1, order="F"), 37, axis=1).reshape(2, n, 37, order="F")
    mo = dict(theta=theta, counts=counts, modes=modes, posterior=np.zeros(len(counts)), output_type=2)
    monkeypatch.setenv("BSSM_AMD_IS_CHUNK", "8")  # 5 chunks
    monkeypatch.setenv("BSSM_AMD_DEFER_SUMMARY", "1")
    a = PC.post_correct(model, mo, 64, threads=2, is_type="is2", seed=9)
    monkeypatch.delenv("BSSM_AMD_DEFER_SUMMARY")
    b = PC.post_correct(model, mo, 64, threads=2, is_type="is2", seed=9)
    monkeypatch.delenv("BSSM_AMD_IS_CHUNK")
    c = PC.post_correct(model, mo, 64, threads=2, is_type="is2", seed=9)  # one chunk
    for key in ("weights", "loglik", "alphahat", "Vt"):
        assert np.array_equal(a[key], b[key]), key
        assert np.array_equal(a[key], c[key]), key
    assert np.all(np.isfinite(a["alphahat"])) and np.all(np.isfinite(a["Vt"]))
