"""GPU parity of the exported helpers: stratified_sample (src/stratified_sample.cpp:9-28, `_bssm_stratified_sample`) and
suggest_N (src/R_postcorrection.cpp:13-148) through the C ABI."""
import numpy as np
import pytest

from bssm_amd import filters as F
from bssm_amd import models as M
from bssm_amd import post_correct as PC

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_elem,N", [(1, 1), (1, 7), (4, 4), (64, 64), (1000, 1000), (1000, 37), (37, 1000), (65536, 65536), (100000, 100000)])
def test_stratified_sample_bit_exact(oracle, n_elem, N):
    rng = np.random.default_rng(n_elem * 31 + N)
    w = rng.gamma(0.3, 1.0, n_elem)  # heavy-tailed weights: many near-zero entries, a few dominant ones
    if n_elem > 8:
        w[rng.integers(0, n_elem, n_elem // 8)] = 0.0  # exact zeros: ties in the cumulative sum
    p = w / w.sum()
    r = rng.random(N)
    idx, cdf = F.stratified_sample(p, r, N)
    ref_idx, ref_cdf = oracle.stratified_sample(p, r, N)
    assert np.array_equal(cdf, ref_cdf)  # sequential cumsum reproduced bit for bit, last entry forced to 1
    assert np.array_equal(idx, ref_idx)
    assert np.all(np.diff(idx.astype(np.int64)) >= 0)  # sortedness: stratified ancestors are non-decreasing


def test_stratified_sample_edge_cases(oracle):
    # hand case of tests/test_oracle_pins.py: all mass on one particle
    idx, cdf = F.stratified_sample(np.array([0, 0, 1.0, 0]), np.array([0.01, 0.99, 0.3, 0.7]))
    assert list(idx) == [2, 2, 2, 2] and list(cdf) == [0, 0, 1, 1]
    # uniform weights, r = 0.5: identity
    idx, _ = F.stratified_sample(np.full(8, 1 / 8), np.full(8, 0.5))
    assert list(idx) == list(range(8))
    # weights summing to < 1: the forced last entry catches the leftover samples (stratified_sample.cpp:13,24-27)
    p = np.array([0.1, 0.1, 0.1])
    r = np.full(5, 0.9)
    idx, cdf = F.stratified_sample(p, r)
    ref, _ = oracle.stratified_sample(p, r)
    assert np.array_equal(idx, ref) and cdf[-1] == 1.0
    with pytest.raises(Exception):
        F.stratified_sample(np.array([0.5, 0.5]), np.array([0.5, 1.5]))  # r outside [0, 1)


def test_suggest_N_c_abi_matches_batches(oracle):
    rng = np.random.default_rng(3)
    y = rng.poisson(4.0, 30).astype(float)
    model = M.bsm_ng(y, sd_level=M.halfnormal(0.2, 1), sd_slope=M.halfnormal(0.02, 1), distribution="poisson", P1=np.diag([1.0, 0.1]))
    theta = np.log([0.25, 0.03])
    cands, reps = [4, 8, 16], 12
    res = PC.suggest_N(model, theta, candidates=cands, replications=reps, seed=5)
    arr = PC.expand_models(model, theta[None, :])[0]
    for i, N in enumerate(cands):
        ll = [oracle.psi_filter(arr, N, seed=5, stream=i * reps + j)["logLik"] for j in range(reps)]
        assert abs(res["results"]["sd"][i] - np.std(ll, ddof=1)) <= 1e-8


def test_suggest_N_nonlinear(oracle):
    rng = np.random.default_rng(8)
    a = np.zeros(30); a[0] = 0.3
    for t in range(1, 30):
        a[t] = 0.2 * (1 - 0.9) + 0.9 * a[t - 1] + rng.normal(0, 0.3)
    model = M.ssm_nlg(np.exp(a) + rng.normal(0, 0.4, 30), "nlg_ar_exp", theta=[0.2, 0.9, np.log(0.3), np.log(0.4)])
    res = PC.suggest_N(model, model["theta"], candidates=[8, 32], replications=10, seed=2)
    ll = F.logLik_batch(model, 32, 10, method="psi", seed=2, stream0=10)
    assert abs(res["results"]["sd"][1] - np.std(ll, ddof=1)) <= 1e-12
    assert np.all(np.isfinite(res["results"]["sd"]))


def _mlg_model(n=40, p=3, m=2, tv=False, seed=5):
    """a small ssm_mlg model with a full (non-diagonal) observation-noise factor and missing values"""
    rng = np.random.default_rng(seed)
    T = np.array([[0.9, 0.1], [0.0, 0.7]])
    R = np.diag([0.3, 0.2])
    Z = rng.normal(0, 1, (p, m, n if tv else 1))
    H = np.tril(rng.normal(0, 0.3, (p, p))) + np.diag(np.full(p, 0.5))
    Hc = np.repeat(H[:, :, None], n, axis=2) * (1 + 0.1 * np.sin(np.arange(n)))[None, None, :] if tv else H
    x = np.zeros((n, m)); y = np.zeros((n, p))
    a = rng.normal(0, 1, m)
    for t in range(n):
        y[t] = 0.2 + Z[:, :, t if tv else 0] @ a + (Hc[:, :, t] if tv else H) @ rng.normal(0, 1, p)
        x[t] = a
        a = T @ a + R @ rng.normal(0, 1, m)
    y[3, 1] = np.nan; y[7, :] = np.nan; y[12, [0, 2]] = np.nan
    return M.ssm_mlg(y, Z, Hc, T, R, a1=np.zeros(m), P1=np.eye(m), D=np.full((p, 1), 0.2))


@pytest.mark.parametrize("tv", [False, True])
def test_gaussian_psi_smoother_mlg_matches_oracle(oracle, tv):
    """gaussian_psi_smoother for ssm_mlg (src/R_psi.cpp:20-25 -> ssm_mlg::psi_filter, src/model_ssm_mlg.cpp:473-501): the
    multivariate smoother_ccov with a full H H' and partially / fully missing observation vectors, then exact draws"""
    import ctypes as C
    from bssm_amd import _lib
    model = _mlg_model(tv=tv)
    arr = M.model_arrays(model)
    smod, keep = _lib.pack_model(arr)
    N = 300
    alpha = np.zeros((arr["m"], arr["n"] + 1, N), order="F")
    _lib.check(_lib.lib().bssm_gaussian_psi_smoother(C.byref(smod), C.c_uint32(N), C.c_uint32(7), C.c_int(0), _lib.dptr(alpha)))
    ref = oracle.mlg_psi_filter(arr, N, seed=7)
    assert np.all(np.isfinite(alpha))
    assert np.allclose(alpha, ref, rtol=1e-9, atol=1e-10)
    # and through the front-end: smoothed moments over the draws
    out = F.particle_smoother(model, 20000, method="psi", seed=3)
    assert out["alphahat"].shape == (arr["n"], arr["m"]) and np.all(np.isfinite(out["Vt"]))
