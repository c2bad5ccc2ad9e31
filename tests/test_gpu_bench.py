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
                      "--no-cpu-baseline", "--no-dropin"])
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


def test_bench_single_rank_line_has_the_contract_fields():
    out = _run_bench(["--steps", "1", "--warmup", "1", "--replicas", "4", "--T", "64", "--pc-draws", "96", "--no-cpu-baseline",
                      "--no-dropin", "--no-mng", "--no-regime-b", "--no-single"])
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 0 and out["communicator"]["ranks"] == 1
    for k in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in out
    assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert out["post_correct"]["per_rank_ms"][0]["draws"] == 96
