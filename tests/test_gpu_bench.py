"""bench.py's multi-rank control flow on a one-GPU box: `--gpus N` outside a launcher must itself start N ranks (one per
GPU over RCCL in production; here BSSM_BENCH_BACKEND=gloo lets two ranks share the GPU) and print ONE line for the job."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(extra, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["BSSM_BENCH_BACKEND"] = "gloo"
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, cwd=ROOT, env=env, capture_output=True, text=True,
                        timeout=timeout)
    assert cp.returncode == 0, cp.stderr[-2000:]
    lines = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, cp.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_gpus_flag_starts_the_ranks_itself():
    out = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "1", "--replicas", "4", "--T", "64", "--pc-draws", "96",
                      "--no-cpu-baseline", "--no-dropin", "--no-live-pmc"])
    assert out["n_gpus"] == 2 and out["steps"] == 1
    assert out["communicator"]["ranks"] == 2 and out["communicator"]["world_size"] == 2
    assert out["communicator"]["backend"] == "gloo" and out["rccl_ranks"] == 0  # (RCCL only when the backend is nccl)
    assert out["communicator"]["distinct_gpus"] == 1                             # the two ranks shared this box's GPU
    assert out["value"] > 0 and out["scaling"] == "weak"
    for key, draws in (("post_correct", 96), ("post_correct_large", 8 * 96)):
        pc = out[key]
        assert "error" not in pc, pc
        assert pc["n_gpus"] == 2 and pc["scaling"] == "strong" and pc["finite_weights"]
        ranks = pc["per_rank_ms"]
        assert [r["rank"] for r in ranks] == [0, 1] and sum(r["draws"] for r in ranks) == draws
        for r in ranks:
            assert set(r) >= {"host_pack_ms", "setup_call_ms", "filter_ms", "local_ms", "wait_ms", "collective_ms"}
            assert r["filter_ms"] > 0 and r["local_ms"] >= r["filter_ms"] * 0.5
    sp = out["post_correct_single_process"]
    assert "error" not in sp, sp
    assert sp["devices"] == [0, 0] and sp["identical_weights"] and sp["gather"] == "device_copies"
    # the strong-scaling evidence at the top level (round 4): both jobs, the communicator's facts, the one-GPU reference
    sc = out["post_correct_scaling"]
    assert sc["scaling"] == "strong" and sc["n_gpus"] == 2 and sc["distinct_gpus"] == 1 and "r05_post_correct_n1.json" in sc["n1_source"]
    for job in ("C5", "C5_x8_draws"):
        assert sc[job]["seconds"] > 0 and sc[job]["particle_steps_per_s"] > 0 and sc[job]["seconds_n1"] > 0
        assert abs(sc[job]["speedup_vs_n1"] - sc[job]["seconds_n1"] / sc[job]["seconds"]) < 1e-12


def test_bench_eight_ranks_shard_the_post_correction_like_the_projection():
    """Multi-GPU readiness without the hardware (round-5 review, item 9): the driver's 8-GPU run must work unattended, so the EIGHT-rank
    control flow runs here with the ranks sharing this box's GPU (gloo): one line, the communicator's facts, every rank's shard timings,
    and each rank's share of the C5 job equal to what the one-GPU projection of the same job says a rank gets"""
    draws = 2340
    out = _run_bench(["--gpus", "8", "--steps", "1", "--warmup", "1", "--replicas", "2", "--T", "32", "--pc-draws", str(draws),
                      "--no-cpu-baseline", "--no-dropin", "--no-live-pmc", "--no-mng", "--no-regime-b", "--no-single", "--no-accuracy",
                      "--no-side-runs", "--no-clock", "--no-live-valu"], timeout=1500)
    assert out["n_gpus"] == 8 and out["communicator"]["ranks"] == 8 and out["communicator"]["world_size"] == 8
    assert out["communicator"]["backend"] == "gloo" and out["rccl_ranks"] == 0 and out["communicator"]["distinct_gpus"] == 1
    assert out["value"] > 0 and out["scaling"] == "weak" and out["config"]["parallelism"] == "replicas only x8"
    pc = out["post_correct"]
    assert "error" not in pc, pc
    assert pc["n_gpus"] == 8 and pc["scaling"] == "strong" and pc["finite_weights"]
    ranks = pc["per_rank_ms"]
    assert [r["rank"] for r in ranks] == list(range(8)) and sum(r["draws"] for r in ranks) == draws
    per_rank = -(-draws // 8)            # projected.C5.draws_per_rank at 8 ranks: ceil(2340 / 8) = 293
    assert max(r["draws"] for r in ranks) == per_rank and min(r["draws"] for r in ranks) >= per_rank - 8
    for r in ranks:
        assert r["filter_ms"] > 0 and r["collective_ms"] >= 0 and r["local_ms"] > 0
    sc = out["post_correct_scaling"]
    assert sc["scaling"] == "strong" and sc["n_gpus"] == 8 and sc["distinct_gpus"] == 1
    assert sc["C5"]["seconds"] > 0 and sc["C5"]["speedup_vs_n1"] > 0


def test_bench_single_rank_line_has_the_contract_fields():
    out = _run_bench(["--steps", "1", "--warmup", "1", "--replicas", "4", "--T", "64", "--pc-draws", "96", "--no-cpu-baseline",
                      "--no-dropin", "--no-mng", "--no-regime-b", "--no-single", "--no-live-valu"])
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 0 and out["communicator"]["ranks"] == 1
    for k in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in out
    assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "frac_by_hip_events"}
    # round 6: frac is the wall-clock figure of the timed region (value x 88 B / 8 TB/s), the HIP-event one sits beside it
    assert abs(out["roofline"]["frac"] - out["value"] * 88.0 / 8e12) < 1e-9 and out["roofline"]["frac_source"].startswith("wall clock")
    # traffic measured in the run (two rocprofv3 --pmc passes of one step of the same batch) wherever rocprofv3 exists
    import shutil
    if shutil.which("rocprofv3") or os.path.exists("/opt/rocm/bin/rocprofv3"):
        live = out["roofline"]["traffic_live"]
        assert out["roofline"]["traffic_source"].startswith("measured in this run") and out["roofline"]["traffic"] == live["hbm_bytes_per_launch"]
        assert 1.9 < live["read_factor_calibrated"] < 2.1 and live["read"] > 0 and live["written"] > 0
        assert 0.5 < out["roofline"]["traffic_over_algorithmic"] < 2.0
    assert out["post_correct"]["per_rank_ms"][0]["draws"] == 96
    # what `value` was quoted on -- the reference's weights layout exp(w - max_t) since round 6 -- and the same batch in the library's
    # factored layout / on the other observed series
    assert out["config"]["series"] == "informative" and "weights layout of `value`: reference" in out["config"]["workload"]
    assert out["config"]["weights_layout"].startswith("reference")
    for key in ("roofline_factored_layout", "other_series"):
        assert out[key]["value"] > 0 and 0 < out[key]["frac"] < 1
    assert out["roofline_factored_layout"]["weights_layout_factored"] is True and out["other_series"]["series"] == "model"
    assert out["post_correct_scaling"]["n1_source"] == "this run" and out["post_correct_scaling"]["C5"]["speedup_vs_n1"] == 1.0
    # the same batch under the contract's choices, inside `config` (round-4 review 6a)
    assert 0 < out["config"]["factored_layout_frac"] < 1 and 0 < out["config"]["contract_series_frac"] < 1
    # projected strong scaling: one rank's share of the job at 1 / 2 / 4 / 8 ranks, measured on this GPU (round-4 review 1c)
    pj = out["post_correct_scaling"]["projected"]
    assert pj["ranks"] == [1, 2, 4, 8] and "before the collective" in pj["what"]
    for job, total in (("C5", 96), ("C5_x8_draws", 768)):
        assert pj[job]["draws_per_rank"] == [-(-total // r) for r in pj["ranks"]]
        assert len(pj[job]["shard_ms"]) == 4 and all(v > 0 for v in pj[job]["shard_ms"]) and pj[job]["speedup"][0] == 1.0


def test_bench_accuracy_object_checks_every_pair_with_the_following_oracle():
    """the metric's second half on a small instance of the timed plan: every (seed, slot) pair re-run by the tie-following oracle"""
    out = _run_bench(["--steps", "1", "--warmup", "1", "--replicas", "4", "--T", "40", "--particles", "8192", "--no-dropin", "--no-mng",
                      "--no-regime-b", "--no-single", "--no-post-correct", "--no-side-runs", "--no-clock", "--no-live-pmc"])
    acc = out["accuracy"]
    assert acc["device_pairs"] == 8 and acc["oracle_checked_pairs"] == 8
    assert acc["unexplained_ancestor_differences"] == 0 and acc["ancestor_tables_identical_after_following"] == 8
    assert acc["loglik_rmse_vs_cpu_same_seeds"] <= 1e-9 * 40 and acc["weighted_state_means_max_rel_diff"] <= 1e-9
    assert acc["pairs_without_a_tie"] + acc["pairs_with_ties"] == 8 and acc["loglik_sd_of_the_estimator_across_streams"] > 0
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["cores"] == 1


def test_seasonal_post_correction_leg_of_the_big_states_object():
    """big_states.period_24.post_correct of the bench line (the psi set-up of 25 states on the device, the host set-up beside it), at a size that takes a second"""
    sys.path.insert(0, ROOT)
    import bench
    r = bench.seasonal_post_correct_leg(24, ns=24, nsim=32, threads=2)
    assert r["seconds"] > 0 and r["draws_per_s"] > 0
    assert "ung_smooth_wave_kernel<32>" in r["setup"] and "ung_factors_rows_kernel<32>" in r["setup"]
    assert r["eight_draws_device_s"] > 0 and r["eight_draws_host_setup_s"] > 0
    # (round 6: the null directions of the singular conditional covariances are zero in every implementation; 3.5e-7 at m = 53 before)
    assert r["max_abs_dloglik_device_vs_host_setup"] < 1e-9 * 72
    assert os.environ.get("BSSM_AMD_DEVICE_SETUP") is None


def test_mng_twenty_states_post_correction_leg_of_the_big_states_object():
    """big_states.mng_m20_post_correct of the bench line (round 6: the psi set-up of ssm_mng draws beyond 8 states on the device), at a size that takes a second"""
    sys.path.insert(0, ROOT)
    import bench
    r = bench.mng_big_post_correct_leg(m=20, p=4, n=20, ns=24, nsim=32, threads=2)
    assert r["seconds"] > 0 and r["draws_per_s"] > 0 and r["eight_draws_device_s"] > 0 and r["eight_draws_host_setup_s"] > 0
    assert r["max_abs_dloglik_device_vs_host_setup"] < 1e-9 * 20
