"""Edge cases of the filter entry points against the oracle: smallest and largest sizes the reference accepts
(R/check_arguments.R:407-418: particles <= 1e5), one time point, one particle, a 16-dimensional state, a series
with no observation at all."""
import numpy as np
import pytest

from bssm_amd import filters as F
from bssm_amd import models as M
from models_zoo import negbin_ar1

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.max(np.abs(a - b) / (1e-300 + np.maximum(1.0, np.abs(b))))


@pytest.mark.parametrize("n,N", [(1, 1), (1, 7), (2, 1), (3, 64), (5, 65)])
def test_tiny_sizes(oracle, n, N):
    rng = np.random.default_rng(n * 100 + N)
    y = rng.poisson(3.0, n).astype(float)
    model = M.bsm_ng(y, sd_level=0.3, sd_slope=0.05, distribution="poisson", P1=np.diag([1.0, 0.1]), a1=[1.0, 0.0])
    arr = M.model_arrays(model)
    out = F.bootstrap_filter(model, N, seed=3, return_indices=True)
    ref = oracle.bsf_filter(arr, N, seed=3)
    assert np.array_equal(out["indices"], ref["indices"])
    assert abs(out["logLik"] - ref["logLik"]) <= 1e-9 * n
    assert _rel(np.transpose(out["alpha"], (1, 0, 2)), ref["alpha"]) <= 1e-9
    sm = F.particle_smoother(model, N, method="psi", seed=3)
    rs = oracle.psi_filter(arr, N, seed=3)
    assert abs(sm["logLik"] - rs["logLik"]) <= 1e-9 * max(n, 1) * 10


def test_maximum_particles_ragged(oracle):
    """particles = 1e5 is the reference's upper limit; not a multiple of the 256-thread block nor of the wave"""
    model = negbin_ar1(12, seed=31)
    N = 100000
    out = F.bootstrap_filter(model, N, seed=5, return_indices=True)
    with oracle.following(out["indices"], N) as f:   # adopts explained last-ulp ties only (oracle/bssm_oracle.c)
        ref = oracle.bsf_filter(M.model_arrays(model), N, seed=5)
    assert f.unexplained == 0 and f.ties <= 1, (f.ties, f.unexplained, f.max_margin)
    assert np.array_equal(out["indices"], ref["indices"])
    assert abs(out["logLik"] - ref["logLik"]) <= 1e-9 * 12
    assert _rel(out["weights"], ref["weights"]) <= 1e-9
    with pytest.raises(ValueError):
        F.bootstrap_filter(model, 100001, seed=5)


@pytest.mark.parametrize("N,path", [(300, "wg"), (800, "wg"), (300, "grid"), (1500, "grid"), (40000, "grid"), (100000, "grid")])
def test_seasonal_model_thirteen_states_workgroup_resident(oracle, monkeypatch, N, path):
    """bsm_ng with a monthly seasonal (m = 13, padded to 16): the workgroup-resident filter runs its transition on the matrix
    cores (late round 4; one particle per lane with 2 KB tiles per wave, two per lane with half-wave tiles inside the CDF
    array), and so does the step kernel of sixteen-dimensional states (path = grid); bootstrap filter and psi-APF smoother against the oracle"""
    from bssm_amd.plan import FilterPlan
    monkeypatch.setenv("BSSM_AMD_FORCE_PATH", path)
    rng = np.random.default_rng(13)
    tt = np.arange(40)
    y = rng.poisson(np.exp(1.5 + 0.5 * np.sin(2 * np.pi * tt / 12.0))).astype(float)
    y[7] = np.nan
    a1 = np.zeros(13); a1[0] = 1.5
    model = M.bsm_ng(y, sd_level=0.05, sd_slope=0.005, sd_seasonal=0.02, distribution="poisson", period=12, a1=a1, P1=0.05 * np.eye(13))
    arr = M.model_arrays(model)
    assert arr["m"] == 13
    out = F.bootstrap_filter(model, N, seed=4, return_indices=True)
    assert ("pf_wg_kernel<16" if path == "wg" else "pf_grid_step_v2<16") in FilterPlan.last_kernel()
    ref = oracle.bsf_filter(arr, N, seed=4)
    assert np.array_equal(out["indices"], ref["indices"])
    assert abs(out["logLik"] - ref["logLik"]) <= 1e-9 * model.n
    assert _rel(np.transpose(out["alpha"], (1, 0, 2)), ref["alpha"]) <= 1e-9
    assert _rel(out["weights"], ref["weights"]) <= 1e-9
    if N > 20000: return   # (ragged blocks at two and at four particles per lane: the bootstrap filter is the case)
    sm = F.particle_smoother(model, N, method="psi", seed=5)
    rs = oracle.psi_filter(arr, N, seed=5)
    assert abs(sm["logLik"] - rs["logLik"]) <= 1e-9 * model.n


@pytest.mark.parametrize("path", ["wg", "grid"])
def test_state_dimension_16(oracle, monkeypatch, path):
    monkeypatch.setenv("BSSM_AMD_FORCE_PATH", path)
    rng = np.random.default_rng(16)
    m, k, n = 16, 5, 9
    T = 0.8 * np.linalg.qr(rng.normal(size=(m, m)))[0]
    R = 0.2 * rng.normal(size=(m, k))
    Z = rng.normal(0, 0.3, m)
    model = M.ssm_ung(rng.poisson(3, n).astype(float), Z=Z, T=T, R=R, a1=np.zeros(m), P1=0.1 * np.eye(m), distribution="poisson")
    out = F.bootstrap_filter(model, 300, seed=2, return_indices=True)
    ref = oracle.bsf_filter(M.model_arrays(model), 300, seed=2)
    assert np.array_equal(out["indices"], ref["indices"])
    assert abs(out["logLik"] - ref["logLik"]) <= 1e-9 * n
    assert _rel(np.transpose(out["alpha"], (1, 0, 2)), ref["alpha"]) <= 1e-9
    assert _rel(out["Pt"], ref_summary(oracle, ref)[2]) <= 1e-8


@pytest.mark.parametrize("path", ["wg", "grid"])
@pytest.mark.parametrize("m,k", [(12, 2), (16, 1), (9, 1), (6, 2), (16, 3)])
def test_stream_format_follows_the_real_noise_dimension(oracle, monkeypatch, path, m, k):
    """The random-stream format of a step draw (WIDE up to two normals, PACKED above; pf_device.h) is a function of the REAL
    number of noise terms k, not of the padded dimension the kernel is instantiated for: a 12-state seasonal model with two
    noise terms, or 16 states with one, must draw exactly as the oracle does (round-4 advisor finding: pad_noise_dim forced
    KP = 4 for MP = 16)."""
    monkeypatch.setenv("BSSM_AMD_FORCE_PATH", path)
    rng = np.random.default_rng(100 * m + k)
    n = 8
    T = 0.8 * np.linalg.qr(rng.normal(size=(m, m)))[0]
    R = 0.2 * rng.normal(size=(m, k))
    Z = rng.normal(0, 0.3, m)
    model = M.ssm_ung(rng.poisson(3, n).astype(float), Z=Z, T=T, R=R, a1=np.zeros(m), P1=0.1 * np.eye(m), distribution="poisson")
    out = F.bootstrap_filter(model, 300, seed=5, return_indices=True)
    ref = oracle.bsf_filter(M.model_arrays(model), 300, seed=5)
    assert np.array_equal(out["indices"], ref["indices"])
    assert abs(out["logLik"] - ref["logLik"]) <= 1e-9 * n
    assert _rel(np.transpose(out["alpha"], (1, 0, 2)), ref["alpha"]) <= 1e-9


def ref_summary(oracle, ref):
    return oracle.filter_summary(ref["alpha"], ref["weights"])


@pytest.mark.parametrize("path", ["wg", "grid"])
def test_no_observation_at_all(oracle, monkeypatch, path):
    """every y missing: all weight columns are ones, resampling uses uniform weights, logLik = 0 (model_ssm_ung.cpp:644-647)"""
    monkeypatch.setenv("BSSM_AMD_FORCE_PATH", path)
    model = M.bsm_ng(np.full(6, np.nan), sd_level=0.3, distribution="poisson", P1=np.array([[1.0]]))
    out = F.bootstrap_filter(model, 500, seed=8, return_indices=True)
    ref = oracle.bsf_filter(M.model_arrays(model), 500, seed=8)
    assert np.all(out["weights"] == 1.0)
    assert np.array_equal(out["indices"], ref["indices"])
    assert out["logLik"] == ref["logLik"] == 0.0


@pytest.mark.parametrize("N,R,ppt", [(20000, 8, None), (65536, 12, None), (65536, 6, "4"), (16384, 8, "2")])
def test_skewed_weights_batches_match_oracle(oracle, monkeypatch, N, R, ppt):
    """Nearly degenerate weights (a few particles carry the mass at every step) through every shape of the step kernel:
    the wave-cooperative ancestor search takes its fall-backs (many block boundaries per wave-range, wide coarse spans),
    the batch is stepped as two launch sequences, and 1 / 2 / 4 particles per lane are used.  A wrong ancestor anywhere
    moves the log-likelihood by far more than the bound."""
    from bssm_amd.filters import logLik_batch
    rng = np.random.default_rng(31)
    n = 14
    y = np.where(np.arange(n) % 3 == 0, 60.0, 1.0) + rng.poisson(1.0, n)
    model = M.bsm_ng(y, sd_level=0.9, sd_slope=0.2, distribution="poisson", P1=np.diag([4.0, 0.5]), a1=[1.0, 0.0])
    monkeypatch.setenv("BSSM_AMD_FORCE_PATH", "grid")
    if ppt:
        monkeypatch.setenv("BSSM_AMD_GRID_PPT", ppt)
    got = logLik_batch(model, N, R, method="bsf", seed=5, stream0=3)
    arr = M.model_arrays(model)
    for r in (0, R // 2, R - 1):  # first filter of each launch sequence and the last one
        ref = oracle.bsf_filter(arr, N, seed=5, stream=3 + r)
        ess = 1.0 / np.sum((ref["weights"][:, 6] / ref["weights"][:, 6].sum()) ** 2)
        assert ess < 0.05 * N  # the case is as skewed as intended
        assert abs(got[r] - ref["logLik"]) <= 1e-9 * n, (r, got[r], ref["logLik"])


@pytest.mark.parametrize("case", range(24))
def test_step_kernel_random_shapes_match_oracle(oracle, monkeypatch, case):
    """Random (particle count, particles per lane, weight skew, state dimension) through the round-2 step kernel's search:
    windows of one and two source blocks, windows that overflow (the generic search), ragged and complete last blocks --
    ancestors, weights and log-likelihood against the oracle"""
    rng = np.random.default_rng(1000 + case)
    ppt = int(rng.choice([2, 4]))
    chunk = 256 * ppt
    nblocks = int(rng.integers(2, 17))
    N = nblocks * chunk if rng.random() < 0.6 else nblocks * chunk - int(rng.integers(1, chunk))
    n = 12
    skew = float(rng.choice([0.05, 0.3, 1.0, 3.0]))          # observation scale: small = nearly degenerate weights
    kind = int(rng.integers(0, 3))
    if kind == 0:
        y = rng.poisson(np.exp(rng.normal(1.0, 0.5, n))).astype(float)
        model = M.bsm_ng(y, sd_level=0.4 / skew, sd_slope=0.1, distribution="poisson", P1=np.diag([2.0 / skew, 0.3]), a1=[1.0, 0.0])
    elif kind == 1:
        y = rng.normal(0, 1, n)
        model = M.ssm_ung(rng.poisson(3.0, n).astype(float), Z=1.0, T=0.9, R=0.5 / skew, a1=1.0, P1=1.0 / skew, distribution="poisson")
    else:
        from models_zoo import negbin_m4
        model = negbin_m4(n, seed=2000 + case)
    monkeypatch.setenv("BSSM_AMD_FORCE_PATH", "grid")
    monkeypatch.setenv("BSSM_AMD_GRID_PPT", str(ppt))
    out = F.bootstrap_filter(model, N, seed=7 + case, return_indices=True)
    ref = oracle.bsf_filter(M.model_arrays(model), N, seed=7 + case)
    assert np.array_equal(out["indices"], ref["indices"]), (N, ppt, kind, skew)
    assert abs(out["logLik"] - ref["logLik"]) <= 1e-9 * n
    assert np.max(np.abs(out["weights"] - ref["weights"])) <= 1e-9


@pytest.mark.parametrize("N,ppt", [(20000, "1"), (40000, "2"), (100000, "4"), (98304, "4"), (65536, "2"),
                                   (33000, "1"), (49153, "1"), (65535, "1"), (65536, "1"), (100000, "2")])
def test_step_kernel_more_than_64_blocks_matches_oracle(oracle, monkeypatch, N, ppt):
    """Filters of 65 ... 128 particle blocks in the round-2 step kernel (two block prefixes per lane in the wave-local
    combine, ballots over both halves) and of 129 ... 256 blocks (round 5: the upper half of the combine in LDS; 129, 193 -- the first
    block of the fourth chunk --, 256 with a ragged last block, 256 complete, 196 at two per lane): ancestors, weights and
    log-likelihood against the oracle"""
    from bssm_amd.plan import FilterPlan
    rng = np.random.default_rng(77)
    n = 8
    y = rng.poisson(np.exp(rng.normal(1.0, 0.5, n))).astype(float)
    model = M.bsm_ng(y, sd_level=0.4, sd_slope=0.1, distribution="poisson", P1=np.diag([2.0, 0.3]), a1=[1.0, 0.0])
    monkeypatch.setenv("BSSM_AMD_FORCE_PATH", "grid")
    monkeypatch.setenv("BSSM_AMD_GRID_PPT", ppt)
    out = F.bootstrap_filter(model, N, seed=21, return_indices=True)
    k = FilterPlan.last_kernel()
    assert k.startswith("pf_grid_step_v2<"), k
    ref = oracle.bsf_filter(M.model_arrays(model), N, seed=21)
    assert np.array_equal(out["indices"], ref["indices"])
    assert abs(out["logLik"] - ref["logLik"]) <= 1e-9 * n
    assert np.max(np.abs(out["weights"] - ref["weights"])) <= 1e-9


@pytest.mark.parametrize("case", ["ragged_ppt2", "psi_ppt4", "mng_ppt2", "degenerate", "wg_shape"])
def test_factored_weights_layout_multiplies_out_to_the_reference_form(oracle, monkeypatch, case):
    """BSSM_PLAN_FACTORED_WEIGHTS (opt-in layout of device-resident plans: block-scaled weights written once + one factor per
    (time point, particle block)): fetch() multiplies the factors out, and weights / ancestors / log-likelihood then meet the
    oracle exactly as the standard layout does -- ragged last block, psi-APF, rows of several series, a filter that
    degenerates half way (later columns zero), and a kernel shape that cannot produce the layout (the flag is then not
    granted and the weights are in the reference's form already)"""
    from bssm_amd.plan import FilterPlan
    from models_zoo import mng_zoo, negbin_m4
    rng = np.random.default_rng(5)
    n, method, force, ppt, N = 10, "bsf", "grid", "2", 3000
    if case == "ragged_ppt2":
        model = negbin_m4(n, seed=7)
    elif case == "psi_ppt4":
        y = rng.poisson(np.exp(rng.normal(1.0, 0.5, n))).astype(float)
        model, method, ppt, N = M.bsm_ng(y, sd_level=0.4, sd_slope=0.1, distribution="poisson", P1=np.diag([2.0, 0.3]), a1=[1.0, 0.0]), "psi", "4", 4096
    elif case == "mng_ppt2":
        model = mng_zoo()["mixed_m4_p3"]; n = model.n
    elif case == "degenerate":
        y = rng.poisson(3.0, n).astype(float)
        D = np.zeros(n); D[5] = 800.0  # exp(signal) overflows at t = 5: every log-weight is -Inf there
        model = M.ssm_ung(y, Z=1.0, T=1.0, R=0.3, a1=0.0, P1=1.0, distribution="poisson", D=D)
    else:
        model, force, N = negbin_m4(n, seed=7), "wg", 512
    monkeypatch.setenv("BSSM_AMD_GRID_PPT", ppt)
    outs = {}
    for fw in (False, True):
        plan = FilterPlan(model, N, method=method, keep_history=True, force=force, n_rep=2, factored_weights=fw)
        plan.run(11, stream0=4)
        lay = plan.weights_layout()
        assert lay["factored"] == (fw and case != "wg_shape"), (case, lay, FilterPlan.last_kernel())
        outs[fw] = (plan.fetch(1), plan.loglik()[1])
        plan.close()
    (a, lla), (b, llb) = outs[False], outs[True]
    assert np.array_equal(a["indices"], b["indices"]) and np.array_equal(a["alpha"], b["alpha"])
    assert (lla == llb) or (np.isneginf(lla) and np.isneginf(llb))
    # same numbers up to the order of two exact-to-1-ulp multiplications (exp(w - max_block) * exp(max_block - max_t))
    np.testing.assert_allclose(a["weights"], b["weights"], rtol=4e-16, atol=0.0, equal_nan=True)
    arr = M.model_arrays(model)
    if case == "mng_ppt2":
        ref = oracle.mng_bsf_filter(arr, N, seed=11, stream=5)
    elif method == "psi":
        ref = oracle.psi_filter(arr, N, seed=11, stream=5)
    else:
        ref = oracle.bsf_filter(arr, N, seed=11, stream=5)
    assert np.array_equal(b["indices"].T, ref["indices"])
    np.testing.assert_allclose(b["weights"].T, ref["weights"], rtol=1e-9, atol=1e-13, equal_nan=True)
    if case == "degenerate":
        assert np.isneginf(llb) and np.all(b["weights"][6:] == 0.0) and np.all(b["indices"][5:] == 0)


def test_series_longer_than_65535_time_points(oracle):
    """n = 70 000: beyond what a grid's y dimension carries (the set-up kernels of the psi paths index time points with it
    and hand such series to the host set-up), beyond 2^16 launches of the step kernels; bootstrap filter, psi-APF smoother
    and the IS post-correction of three draws against the oracle"""
    from bssm_amd import post_correct as PC
    n = 70000
    rng = np.random.default_rng(70)
    level = 1.0 + np.cumsum(rng.normal(0, 0.02, n))
    level -= np.linspace(0, level[-1] - 1.0, n)           # stays near 1: informative counts all along
    y = rng.poisson(np.exp(level)).astype(float)
    y[[17, 40000, 69999]] = np.nan
    model = M.bsm_ng(y, sd_level=M.halfnormal(0.05, 1), distribution="poisson", P1=np.array([[0.5]]), a1=[1.0])
    arr = M.model_arrays(model)
    for N in (64, 1500):   # workgroup-resident and step kernels
        out = F.bootstrap_filter(model, N, seed=8, return_indices=True)
        ref = oracle.bsf_filter(arr, N, seed=8)
        assert np.array_equal(out["indices"], ref["indices"])
        assert abs(out["logLik"] - ref["logLik"]) <= 1e-9 * n
    sm = F.particle_smoother(model, 64, method="psi", seed=9)
    rs = oracle.psi_filter(arr, 64, seed=9)
    assert abs(sm["logLik"] - rs["logLik"]) <= 1e-9 * n
    theta = np.log(np.array([[0.04], [0.05], [0.07]]))
    counts = np.array([2, 1, 3], dtype=np.uint32)
    arrs = PC.expand_models(model, theta)
    modes = np.stack([oracle.psi_setup(a)["mode"] for a in arrs], axis=1)
    res = PC.is_correction(model, theta, counts, 32, method="psi", modes=modes, seed=4, threads=2, output_type=3)
    ref = oracle.is_correction(arrs, 32, 4, method=1, counts=counts, modes=modes, is_type=2, output_type=3, n_threads=2)
    assert np.allclose(res["loglik"], ref["loglik"], rtol=0, atol=1e-9 * n)
    assert np.allclose(res["weights"], ref["weights"], rtol=1e-7)
