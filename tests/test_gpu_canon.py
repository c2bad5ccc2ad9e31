"""The two kernel families that can run a filter of at most 8192 particles -- pf_wg_kernel (one workgroup per filter, particle
system in LDS) and pf_row_step (one wave per row of 64 particles, one launch per time step) -- evaluate ONE normalisation /
resampling arithmetic (bssm_amd/csrc/pf_canon.h), so a filter's result is the SAME BITS whichever ran it.  That is what lets a
post-correction shard send a sparsely filled last round through the row family without the output depending on the shard layout
(round-4 review item 1; reference loop: src/approx_mcmc.cpp:364-417)."""
import os

import numpy as np
import pytest

from bssm_amd import filters as F
from bssm_amd import models as M
from bssm_amd.plan import FilterPlan
from models_zoo import negbin_ar1, negbin_m4, small_zoo

pytestmark = pytest.mark.gpu


def _both(monkeypatch, run):
    out = {}
    for path in ("wg", "row"):
        monkeypatch.setenv("BSSM_AMD_FORCE_PATH", path)
        c0 = FilterPlan.family_launches()
        out[path] = run()
        c1 = FilterPlan.family_launches()
        assert c1[path] > c0[path], (path, c0, c1)   # the call really took that family
    monkeypatch.delenv("BSSM_AMD_FORCE_PATH")
    return out["wg"], out["row"]


@pytest.mark.parametrize("N", [100, 1000, 2048, 3000, 4096])
@pytest.mark.parametrize("method", ["bsf", "psi"])
def test_linear_state_filters_same_bits_in_both_families(monkeypatch, N, method):
    """log-likelihoods of 5 replicas: m = 1 (WIDE stream), m = 2, m = 4 with four noise terms (PACKED stream), ragged and
    complete rows, one or two row prefixes per lane"""
    for name, model in (("negbin_ar1", negbin_ar1(40, seed=3)), ("negbin_m4", negbin_m4(40, seed=4)), ("poisson_slope_missing", small_zoo()["poisson_slope_missing"]), ("poisson_tv_m3", small_zoo()["poisson_tv_m3"])):
        if N > {1: 8192, 2: 4096}.get(M.model_arrays(model)["T"].shape[0], 2048):
            continue   # beyond the workgroup-resident capacity of this state dimension
        wg, row = _both(monkeypatch, lambda: F.logLik_batch(model, N, 5, method=method, seed=11, stream0=3))
        assert np.all(np.isfinite(wg)), name
        assert np.array_equal(wg, row), (name, N, method, wg - row)


def test_univariate_state_at_the_largest_size(monkeypatch):
    """N = 8192 (m = 1): 128 rows, two row prefixes per lane in both families"""
    model = negbin_ar1(30, seed=5)
    wg, row = _both(monkeypatch, lambda: F.logLik_batch(model, 8192, 3, method="bsf", seed=2))
    assert np.array_equal(wg, row), wg - row


@pytest.mark.parametrize("N", [500, 4096])
def test_functor_models_same_bits_in_both_families(monkeypatch, N):
    """ssm_nlg growth model (two states): bootstrap filter, psi-APF and EKF particle filter; ssm_sde with Milstein sub-steps"""
    rng = np.random.default_rng(7)
    n = 40
    y = 50.0 + np.cumsum(rng.normal(0.5, 1.0, n)) + rng.normal(0, 1, n)
    growth = M.ssm_nlg(y, "nlg_growth", theta=np.array([0.0, np.log(0.05), 0.0]), known_params=[0.1, 500.0, 0.0, 50.0, 1.0, 100.0])
    for method in ("bsf", "psi", "ekf"):
        wg, row = _both(monkeypatch, lambda: F.logLik(growth, N, method=method, seed=5))
        assert np.isfinite(wg) and wg == row, (method, N, wg, row)


def test_degenerate_filter_stops_at_the_same_step_in_both_families(monkeypatch):
    """exp(signal) overflows (D = 800): every weight is -inf at the first time point, -Inf from both families
    (src/model_ssm_ung.cpp:678-682)"""
    import warnings
    y = np.array([1.0, 2.0, 1e6, 3.0, 1.0])
    model = M.ssm_ung(y, Z=1.0, T=1.0, R=1e-3, a1=0.0, P1=1e-6, distribution="poisson", D=800.0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        wg, row = _both(monkeypatch, lambda: F.logLik_batch(model, 2048, 2, method="bsf", seed=1))
    assert np.all(wg == -np.inf) and np.all(row == -np.inf)


@pytest.mark.parametrize("name,N", [("negbin_ar1", 4096), ("negbin_m4", 2000), ("poisson_slope_missing", 777), ("svm", 1500)])
def test_full_outputs_same_bits_in_both_families(monkeypatch, name, N):
    """bootstrap_filter / particle_smoother with every output: particle history, ancestor table, WEIGHTS (the row family writes them
    row-scaled and rescales them one launch later) and summaries equal bit for bit"""
    model = dict(small_zoo(), negbin_ar1=negbin_ar1(30, seed=3), negbin_m4=negbin_m4(30, seed=4))[name]
    wg, row = _both(monkeypatch, lambda: F.bootstrap_filter(model, N, seed=13, return_indices=True))
    for k in ("indices", "weights", "alpha", "at", "att", "Pt", "Ptt"):
        assert np.array_equal(wg[k], row[k]), (name, k)
    assert wg["logLik"] == row["logLik"]
    wg, row = _both(monkeypatch, lambda: F.particle_smoother(model, N, method="psi", seed=14))
    for k in ("alphahat", "Vt", "alpha"):
        assert np.array_equal(wg[k], row[k]), (name, k)


@pytest.mark.parametrize("name,N", [("negbin_ar1", 12000), ("poisson_slope_missing", 16384), ("negbin_m4", 9000)])
def test_row_family_beyond_the_workgroup_capacity_against_the_oracle(oracle, monkeypatch, name, N):
    """the row family takes filters of up to 16 384 particles (four row-prefix slots per lane): 129 ... 256 rows, against the oracle"""
    monkeypatch.setenv("BSSM_AMD_FORCE_PATH", "row")
    model = dict(small_zoo(), negbin_ar1=negbin_ar1(25, seed=3), negbin_m4=negbin_m4(25, seed=4))[name]
    c0 = FilterPlan.family_launches()
    out = F.bootstrap_filter(model, N, seed=21, return_indices=True)
    assert FilterPlan.family_launches()["row"] == c0["row"] + 1
    arr = M.model_arrays(model)
    ref = oracle.bsf_filter(arr, N, seed=21)
    assert np.array_equal(out["indices"], ref["indices"])
    assert abs(out["logLik"] - ref["logLik"]) <= 1e-9 * model.n
    assert np.allclose(out["weights"], ref["weights"], rtol=1e-8, atol=1e-13)
