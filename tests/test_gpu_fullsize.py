"""BASELINE.json configurations at their full sizes, checked through size-independent properties (the oracle needs
minutes per case at these sizes): sortedness of the stratified ancestors, weight normalisation, determinism,
agreement of independent kernel paths (MFMA vs VALU transition, grid vs workgroup shape, sharded vs unsharded
post-correction) and psi-APF vs bootstrap agreement within Monte-Carlo error."""
import os

import numpy as np
import pytest

from bssm_amd import filters as F
from bssm_amd import models as M
from bssm_amd import post_correct as PC
from bssm_amd.plan import FilterPlan
from models_zoo import mng_c4, negbin_ar1, negbin_m4

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _check_particle_system(out, n, N, y):
    idx, w = out["indices"], out["weights"]
    assert idx.shape == (N, n) and w.shape == (N, n + 1)
    assert np.all(np.diff(idx.astype(np.int64), axis=0) >= 0), "stratified ancestors must be non-decreasing per column"
    assert idx.max() < N
    assert np.all(np.isfinite(w)) and w.min() >= 0.0
    obs = np.isfinite(np.asarray(y).reshape(n, -1)).any(axis=1)
    assert np.all(w[:, :n][:, obs].max(axis=0) == 1.0), "exp(w - max) has maximum exactly 1 on observed columns"
    assert np.all(w[:, n] == 1.0) and np.all(w[:, :n][:, ~obs] == 1.0)
    assert np.isfinite(out["logLik"])


def test_c2_full_size_properties_and_determinism():
    """configs[1]: ssm_ung negbin AR(1), T=1000, m=1, N=65536, full-output bootstrap filter"""
    model = negbin_ar1(1000, seed=102)
    out = F.bootstrap_filter(model, 65536, seed=7, return_indices=True)
    _check_particle_system(out, 1000, 65536, model["y"])
    again = F.bootstrap_filter(model, 65536, seed=7, return_indices=True)
    for k in ("indices", "weights", "alpha", "at", "att", "Pt", "Ptt"):
        assert np.array_equal(out[k], again[k]), k
    assert out["logLik"] == again["logLik"]
    # filter means follow from the particle system: at = plain mean, att = weight-normalised mean (summary.cpp:42-79)
    a = out["alpha"][:, 0, :]                       # (n+1) x N
    assert np.allclose(out["at"][:, 0], a.mean(axis=1), rtol=1e-10, atol=1e-12)
    wn = out["weights"][:, :1000] / out["weights"][:, :1000].sum(axis=0)
    assert np.allclose(out["att"][:, 0], (a[:1000] * wn.T).sum(axis=1), rtol=1e-9, atol=1e-12)
    # psi-APF and bootstrap filter estimate the same likelihood (test_particle_smoother.R:33-45 at scale)
    ll_psi = F.logLik(model, 65536, method="psi", seed=3)
    assert abs(ll_psi - out["logLik"]) < 0.5, (ll_psi, out["logLik"])


def test_c2_m4_batched_plan_equals_single_calls():
    """metric config (m=4, N=65536, T=1000): filters advanced together in one batch = the same filters run alone"""
    model = negbin_m4(1000, seed=103)
    p = FilterPlan(model, 65536, method="bsf", keep_history=True, force="grid", n_rep=3)
    p.run(5, stream0=10)
    ll_batch = p.loglik()
    p.close()
    p1 = FilterPlan(model, 65536, method="bsf", keep_history=False, force="grid", n_rep=1)
    for r in range(3):
        p1.run(5, stream0=10 + r)
        # different particles-per-thread / history settings: same ancestors, reductions differ in shape only
        assert abs(p1.loglik()[0] - ll_batch[r]) <= 1e-9 * 1000
    p1.close()


def test_c3_svm_exchange_psi_smoother_full_size():
    """configs[2]: svm on the exchange data, psi-APF N=32768 + filter-smoother back-trace"""
    y = np.loadtxt(os.path.join(GOLD, "exchange.txt"))
    model = M.svm(y, rho=M.uniform(0.98, -0.999, 0.999), sd_ar=M.halfnormal(0.2, 5), sigma=M.halfnormal(1.0, 2))
    n, N = y.size, 32768
    out = F.particle_smoother(model, N, method="psi", seed=11)
    assert out["alpha"].shape == (n, 1, N) and np.all(np.isfinite(out["alphahat"]))
    assert np.all(out["Vt"][0, 0, :] >= 0.0)
    # back-traced trajectories coalesce going backwards in time: distinct ancestors never increase
    uniq = [np.unique(out["alpha"][t, 0, :]).size for t in (0, n // 2, n - 1)]
    assert uniq[0] <= uniq[1] <= uniq[2] and uniq[0] < N
    # alphahat is the plain mean of the back-traced trajectories (weights are uniform after the extra time point, R_psi.cpp:95)
    assert np.allclose(out["alphahat"][:, 0], out["alpha"][:, 0, :].mean(axis=1), rtol=1e-10, atol=1e-12)
    bsf = F.particle_smoother(model, N, method="bsf", seed=12)
    assert abs(out["logLik"] - bsf["logLik"]) < 1.0
    assert np.max(np.abs(out["alphahat"] - bsf["alphahat"])) < 0.25  # smoothed log-volatility, sd ~ 0.5


def test_c5_large_remainder_beside_all_rounds_is_invisible_in_the_output(monkeypatch):
    """C5 shape, 3 x 256 + 150 draws: the remainder is more than half a round, and the chunk has three full rounds -- it goes through the row
    family BESIDE all three (which run the 96-register form of the workgroup kernel; api_postcorrect.cpp, late round 5).  Same bits as with
    every draw workgroup-resident (BSSM_AMD_TAIL_SPLIT=0) and as with the remainder in a round of its own (BSSM_AMD_TAIL_CONCURRENT=0)."""
    import torch
    from bssm_amd.plan import FilterPlan
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    rng = np.random.default_rng(107)
    n, K, dT = 60, 500.0, 0.1
    r = np.zeros(n); pp = np.zeros(n); pp[0] = 50.0
    for t in range(1, n):
        r[t] = r[t - 1] + rng.normal(0, 0.05)
        rr = np.exp(r[t - 1]) / (1 + np.exp(r[t - 1]))
        pp[t] = K * pp[t - 1] * np.exp(rr * dT) / (K + pp[t - 1] * (np.exp(rr * dT) - 1)) + rng.normal(0, 1)
    y = pp + rng.normal(0, 1, n)
    th0 = np.array([0.0, np.log(0.05), 0.0])
    model = M.ssm_nlg(y, "nlg_growth", theta=th0, known_params=[dT, K, 0.0, 50.0, 1.0, 100.0])
    ns = 3 * cus + (cus * 5) // 8
    theta = th0 + rng.normal(0, 0.05, (ns, 3))
    counts = np.ones(ns, dtype=np.uint32)
    mode = F.psi_setup(model)["mode"].reshape(-1, order="F")
    modes = np.tile(mode[:, None], (1, ns))
    c0 = FilterPlan.family_launches()
    beside = PC._run_local_block_nlg(model, theta, modes, counts, 0, ns, 4096, 9, 2, 3, 4)
    c1 = FilterPlan.family_launches()
    assert c1["row"] == c0["row"] + 1 and c1["wg"] == c0["wg"] + 1, (c0, c1)   # one lean launch of three rounds + the row family
    monkeypatch.setenv("BSSM_AMD_TAIL_CONCURRENT", "0")
    own_round = PC._run_local_block_nlg(model, theta, modes, counts, 0, ns, 4096, 9, 2, 3, 4)
    c2 = FilterPlan.family_launches()
    assert c2["row"] == c1["row"] and c2["wg"] == c1["wg"] + 1, (c1, c2)       # (more than half a round: no split without the concurrency)
    monkeypatch.delenv("BSSM_AMD_TAIL_CONCURRENT")
    for k in ("loglik", "weights"):
        assert np.array_equal(beside[k], own_round[k]), k
    assert np.all(np.isfinite(beside["loglik"]))


def test_c4_mng_two_matrix_core_transitions_agree(monkeypatch):
    """configs[3]: ssm_mng m=8 p=4 T=2000 N=16384: the step kernel's fp64-MFMA state transition through LDS tiles (one and two
    particles per lane) and the row family's fragment-native transition (pf_big.h, one 16-row block) are independent
    implementations of alpha' = C + T a + R eta (the round-1 step kernel with the plain FMA transition, the third one until
    round 5, is gone; the oracle comparison at full size is tests/test_gpu_parity.py)"""
    model = mng_c4(2000)
    N = 16384
    monkeypatch.setenv("BSSM_AMD_FORCE_PATH", "grid")
    mfma = F.bootstrap_filter(model, N, seed=4, return_indices=True)
    from bssm_amd.plan import FilterPlan
    assert "pf_grid_step_v2<8," in FilterPlan.last_kernel() and "true" in FilterPlan.last_kernel()
    monkeypatch.setenv("BSSM_AMD_GRID_PPT", "2")
    mfma2 = F.bootstrap_filter(model, N, seed=4, return_indices=True)
    assert np.mean(mfma["indices"] != mfma2["indices"]) < 1e-6
    monkeypatch.delenv("BSSM_AMD_GRID_PPT")
    monkeypatch.setenv("BSSM_AMD_FORCE_PATH", "row")
    valu = F.bootstrap_filter(model, N, seed=4, return_indices=True)
    assert "pf_big_row_step<1,8>" in FilterPlan.last_kernel(), FilterPlan.last_kernel()
    _check_particle_system(mfma, 2000, N, model["y"])
    frac = np.mean(mfma["indices"] != valu["indices"])
    assert frac < 1e-6, f"ancestor mismatch fraction {frac}"
    tol = 1e-9 if frac == 0 else 1e-4
    assert abs(mfma["logLik"] - valu["logLik"]) <= tol * 2000
    if frac == 0:
        assert np.max(np.abs(mfma["alpha"] - valu["alpha"])) <= 1e-9 * max(1.0, np.abs(valu["alpha"]).max())


def test_c5_post_correct_block_invariance():
    """configs[4] shape (ssm_nlg growth, psi N=4096): filtering the draws in two blocks (as two GPUs would) gives
    bit-identical weights and summaries to one block -- streams are keyed by the global draw index"""
    rng = np.random.default_rng(105)
    n, K, dT = 300, 500.0, 0.1
    r = np.zeros(n); pp = np.zeros(n); pp[0] = 50.0
    for t in range(1, n):
        r[t] = r[t - 1] + rng.normal(0, 0.05)
        rr = np.exp(r[t - 1]) / (1 + np.exp(r[t - 1]))
        pp[t] = K * pp[t - 1] * np.exp(rr * dT) / (K + pp[t - 1] * (np.exp(rr * dT) - 1)) + rng.normal(0, 1)
    y = pp + rng.normal(0, 1, n)
    th0 = np.array([0.0, np.log(0.05), 0.0])
    model = M.ssm_nlg(y, "nlg_growth", theta=th0, known_params=[dT, K, 0.0, 50.0, 1.0, 100.0])
    ns = 96
    theta = th0 + rng.normal(0, 0.05, (ns, 3))
    counts = rng.geometric(0.234, ns).astype(np.uint32)
    mode = F.psi_setup(model)["mode"].reshape(-1, order="F")
    modes = np.tile(mode[:, None], (1, ns))
    full = PC._run_local_block_nlg(model, theta, modes, counts, 0, ns, 4096, 9, 2, 2, 4)
    lo = PC._run_local_block_nlg(model, theta[:40], modes[:, :40], counts[:40], 0, ns, 4096, 9, 2, 2, 1)
    hi = PC._run_local_block_nlg(model, theta[40:], modes[:, 40:], counts[40:], 40, ns, 4096, 9, 2, 2, 7)
    assert np.all(np.isfinite(full["loglik"])) and np.all(full["weights"] > 0)
    for k in ("weights", "loglik"):
        assert np.array_equal(full[k], np.concatenate([lo[k], hi[k]])), k
    for k in ("alphahat_i", "Vt_i"):
        assert np.array_equal(full[k], np.concatenate([lo[k], hi[k]], axis=-1)), k
    # the IS weights are likelihood ratios to the Gaussian approximation at the stored mode: spread stays moderate
    lw = full["loglik"]
    assert lw.std() < 5.0


def test_c5_post_correct_tail_round_is_invisible_in_the_output(monkeypatch):
    """configs[4] shape, 300 draws: one workgroup-resident filter per CU, so 256 draws fill a round and 44 are left over.  By
    DEFAULT (round 5) the 44 go through the row family beside the full round (pf_row_step.h: one wave per row of 64 particles, one
    launch per time step; the round in the 96-register form of its kernel).  Both families evaluate pf_canon.h, so a draw's result is the SAME BITS whichever ran it: the job with the
    split, the job without it (BSSM_AMD_TAIL_SPLIT=0: every draw workgroup-resident), two blocks of 150 (what two GPUs would filter:
    no tail round there) and the in-library two-shard call agree bit for bit -- log-weights, the smoothing summaries (output 2) and
    the sampled trajectories (output 1).  (Rounds 3-4: the tail took the step kernels, opt-in, equal to reduction-order rounding.)"""
    from bssm_amd.plan import FilterPlan
    rng = np.random.default_rng(106)
    n, K, dT = 300, 500.0, 0.1
    r = np.zeros(n); pp = np.zeros(n); pp[0] = 50.0
    for t in range(1, n):
        r[t] = r[t - 1] + rng.normal(0, 0.05)
        rr = np.exp(r[t - 1]) / (1 + np.exp(r[t - 1]))
        pp[t] = K * pp[t - 1] * np.exp(rr * dT) / (K + pp[t - 1] * (np.exp(rr * dT) - 1)) + rng.normal(0, 1)
    y = pp + rng.normal(0, 1, n)
    th0 = np.array([0.0, np.log(0.05), 0.0])
    model = M.ssm_nlg(y, "nlg_growth", theta=th0, known_params=[dT, K, 0.0, 50.0, 1.0, 100.0])
    ns = 300
    theta = th0 + rng.normal(0, 0.05, (ns, 3))
    counts = np.ones(ns, dtype=np.uint32)
    mode = F.psi_setup(model)["mode"].reshape(-1, order="F")
    modes = np.tile(mode[:, None], (1, ns))
    for output_type, keys in ((3, ("loglik", "weights")), (2, ("loglik", "weights", "alphahat_i", "Vt_i")), (1, ("loglik", "weights", "alpha"))):
        c0 = FilterPlan.family_launches()
        split = PC._run_local_block_nlg(model, theta, modes, counts, 0, ns, 4096, 9, 2, output_type, 4)
        c1 = FilterPlan.family_launches()
        assert c1["row"] == c0["row"] + 1 and c1["wg"] == c0["wg"] + 1, (c0, c1)   # the tail really took the row family
        monkeypatch.setenv("BSSM_AMD_TAIL_SPLIT", "0")
        whole = PC._run_local_block_nlg(model, theta, modes, counts, 0, ns, 4096, 9, 2, output_type, 4)
        c2 = FilterPlan.family_launches()
        assert c2["row"] == c1["row"] and c2["wg"] == c1["wg"] + 1, (c1, c2)
        monkeypatch.delenv("BSSM_AMD_TAIL_SPLIT")
        assert np.all(np.isfinite(split["loglik"]))
        for k in keys:
            assert np.array_equal(split[k], whole[k]), (output_type, k)
        # (late round 5) by default the tail steps BESIDE the full round, which then runs the 96-register form of the workgroup kernel
        # (pf_wg_kernel_lean); BSSM_AMD_TAIL_CONCURRENT=0 puts it behind the round and the round back on the 128-register kernel: same bits
        monkeypatch.setenv("BSSM_AMD_TAIL_CONCURRENT", "0")
        behind = PC._run_local_block_nlg(model, theta, modes, counts, 0, ns, 4096, 9, 2, output_type, 4)
        monkeypatch.delenv("BSSM_AMD_TAIL_CONCURRENT")
        for k in keys:
            assert np.array_equal(split[k], behind[k]), (output_type, k)
        lo = PC._run_local_block_nlg(model, theta[:150], modes[:, :150], counts[:150], 0, ns, 4096, 9, 2, output_type, 2)
        hi = PC._run_local_block_nlg(model, theta[150:], modes[:, 150:], counts[150:], 150, ns, 4096, 9, 2, output_type, 2)
        for k in keys:
            ax = 0 if split[k].ndim == 1 else -1
            assert np.array_equal(split[k], np.concatenate([lo[k], hi[k]], axis=ax)), (output_type, k)
        # the same job cut over two logical devices inside the library (bssm_is_correction_nlg_multi; both shards on this GPU)
        shard = PC._run_local_block_nlg(model, theta, modes, counts, 0, ns, 4096, 9, 2, output_type, 4, devices=[0, 0])
        for k in keys:
            assert np.array_equal(shard[k], split[k]), (output_type, k)
