"""world_size-2 gloo test of the theta-sharded post-correction: sharding, the single all-gather phase and
the draw-order merge, on CPU.  The per-block compute is INJECTED from the oracle here (test infrastructure);
the product's own block runner goes through the HIP library and has no CPU path."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_block(arrs, modes, counts, first, n_stored, nsim, seed, is_type, output_type, n_threads, devices=None):
    """Stand-in for bssm_amd.post_correct._run_local_block built on the oracle (streams keyed by global index)."""
    from oracle import oracle as O
    cnt = len(arrs)
    n, m = arrs[0]["n"], arrs[0]["m"]
    weights, loglik = np.zeros(cnt), np.zeros(cnt)
    alpha = np.zeros((n + 1, m, cnt), order="F")
    ah = np.zeros((m, n + 1, cnt), order="F")
    vt = np.zeros((m, m, n + 1, cnt), order="F")
    for i in range(cnt):
        N = nsim * int(counts[i]) if is_type == 1 else nsim
        res = O.psi_filter(arrs[i], N, seed=seed, stream=first + i, is_mode=modes[:, i])
        loglik[i] = res["logLik"]
        weights[i] = np.exp(res["logLik"])
        if output_type != 3:
            sm = O.filter_smoother(res["alpha"], res["indices"])
            if output_type == 2:
                ah[:, :, i], vt[:, :, :, i] = O.summary(sm)
            else:
                import ctypes as C
                rng, _ = O.make_rng(seed=seed, stream=first + i)
                pick = min(int(O.lib().orc_rng_uniform(C.byref(rng), 3, 0, 0, 0) * N), N - 1)
                alpha[:, :, i] = sm[:, :, pick].T
    return dict(weights=weights, loglik=loglik, alpha=alpha, alphahat_i=ah, Vt_i=vt, degenerate=False)


def _make_job():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from bssm_amd import models as M
    from oracle import oracle as O
    rng = np.random.default_rng(4)
    y = rng.poisson(4.0, 30).astype(float)
    model = M.bsm_ng(y, sd_level=M.halfnormal(0.2, 1), sd_slope=M.halfnormal(0.02, 1), distribution="poisson",
                     P1=np.diag([1.0, 0.1]))
    ns = 7
    theta = np.log(np.column_stack([rng.uniform(0.1, 0.4, ns), rng.uniform(0.01, 0.05, ns)]))
    counts = np.array([3, 1, 2, 5, 1, 1, 4], dtype=np.uint32)
    from bssm_amd.post_correct import expand_models
    modes = np.stack([O.psi_setup(a)["mode"] for a in expand_models(model, theta)], axis=1)
    return model, dict(theta=theta, counts=counts, modes=modes, posterior=np.zeros(ns), output_type="summary")


def _worker(rank, world, port, outdir, is_type, output_type):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bssm_amd.post_correct as PC
    PC._run_local_block = _oracle_block
    model, mo = _make_job()
    mo["output_type"] = output_type
    out = PC.post_correct(model, mo, 12, threads=1, is_type=is_type, seed=5)
    np.savez(os.path.join(outdir, f"r{rank}.npz"), **{k: v for k, v in out.items()
                                                       if isinstance(v, np.ndarray) and v.dtype != object})
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("is_type,output_type", [("is2", "summary"), ("is1", "full"), ("is3", "theta")])
def test_two_rank_post_correct_equals_single_process(tmp_path, is_type, output_type):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), is_type, output_type), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    # single-process reference with the same injected block runner
    import bssm_amd.post_correct as PC
    old = PC._run_local_block
    PC._run_local_block = _oracle_block
    try:
        model, mo = _make_job()
        mo["output_type"] = output_type
        single = PC.post_correct(model, mo, 12, threads=1, is_type=is_type, seed=5)
    finally:
        PC._run_local_block = old
    for k in ("weights", "posterior", "loglik") + (("alphahat", "Vt") if output_type == "summary" else ()) + \
            (("alpha",) if output_type == "full" else ()):
        assert np.array_equal(r0[k], r1[k]), k                # every rank holds the full result
        assert np.array_equal(r0[k], single[k]), k            # and it does not depend on the number of ranks
    assert r0["shards"][0] == 0 and r0["shards"][-1] == r0["weights"].size and len(r0["shards"]) == 3
    assert np.all(np.isfinite(r0["weights"])) and r0["weights"].max() > 0


def test_shard_bounds_balance_by_work():
    from bssm_amd.post_correct import shard_bounds
    b = shard_bounds(np.ones(10), 4)
    assert list(b) == [0, 3, 5, 8, 10] or (b[0] == 0 and b[-1] == 10 and np.all(np.diff(b) >= 2))
    work = np.array([10, 1, 1, 1, 1, 1, 1, 1, 1, 2.0])
    b = shard_bounds(work, 2)
    assert b[0] == 0 and b[-1] == 10 and abs(work[:b[1]].sum() - work[b[1]:].sum()) <= 10
    assert list(shard_bounds(np.ones(2), 4)) == [0, 1, 1, 2, 2] or shard_bounds(np.ones(2), 4)[-1] == 2
    b = shard_bounds(np.ones(5), 1)
    assert list(b) == [0, 5]


def _oracle_replications(mdl, particles, n_rep, seed, stream0):
    """Stand-in for bssm_amd.post_correct._loglik_replications built on the oracle"""
    from bssm_amd import models as M
    from oracle import oracle as O
    arr = M.model_arrays(mdl)
    su = O.psi_setup(arr)
    return np.array([O.psi_filter(arr, particles, seed=seed, stream=stream0 + j, setup=su)["logLik"] for j in range(n_rep)])


def _suggest_worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bssm_amd.post_correct as PC
    PC._loglik_replications = _oracle_replications
    model, mo = _make_job()
    res = PC.suggest_N(model, mo["theta"][0], candidates=[4, 8, 16], replications=9, seed=3)
    np.savez(os.path.join(outdir, f"s{rank}.npz"), sd=res["results"]["sd"], N=np.array([-1 if res["N"] is None else res["N"]]))
    dist.destroy_process_group()


def test_two_rank_suggest_N_equals_single_process(tmp_path):
    """suggest_N (R/post_correction.R:84-130) shards its independent replications over the ranks: streams are
    keyed by (candidate, replication), so two ranks reproduce the single-process standard deviations"""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_suggest_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "s0.npz"), np.load(tmp_path / "s1.npz")
    assert np.array_equal(r0["sd"], r1["sd"]) and r0["N"][0] == r1["N"][0]
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from bssm_amd.post_correct import expand_models
    from bssm_amd import models as M
    model, mo = _make_job()
    arr = expand_models(model, mo["theta"][:1])[0]
    mdl = M.BssmModel(model.classes, **{**model, **{k: arr[k] for k in ("Z", "T", "R", "P1", "a1", "C", "D", "beta")}, "phi": arr["phi"]})
    for i, c in enumerate([4, 8, 16]):
        ll = _oracle_replications(mdl, c, 9, 3, i * 9)
        assert abs(r0["sd"][i] - ll.std(ddof=1)) <= 1e-12


def _mcmc_worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bssm_amd.post_correct as PC
    from bssm_amd import mcmc
    PC._run_local_block = _oracle_block
    model, _ = _make_job()
    out = mcmc.run_mcmc(model, 120, particles=8, burnin=40, mcmc_type="is2", output_type="theta", seed=8)
    np.savez(os.path.join(outdir, f"m{rank}.npz"), theta=out["theta"], counts=out["counts"], weights=out["weights"],
             posterior=out["posterior"])
    dist.destroy_process_group()


def test_two_rank_run_mcmc_is2_equals_single_process(tmp_path):
    """BASELINE config C5's shape: run_mcmc(mcmc_type = "is2") under torch.distributed -- every rank runs the (deterministic,
    host-side) approximate chain, the post-correction of its stored values shards over the ranks; the result is the same on
    every rank and equal to the single-process run.  The per-block filters are injected from the oracle here."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_mcmc_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "m0.npz"), np.load(tmp_path / "m1.npz")
    import bssm_amd.post_correct as PC
    from bssm_amd import mcmc
    old = PC._run_local_block
    PC._run_local_block = _oracle_block
    try:
        model, _ = _make_job()
        single = mcmc.run_mcmc(model, 120, particles=8, burnin=40, mcmc_type="is2", output_type="theta", seed=8)
    finally:
        PC._run_local_block = old
    for k in ("theta", "counts", "weights", "posterior"):
        assert np.array_equal(r0[k], r1[k]), k
        assert np.array_equal(r0[k], single[k]), k
    assert np.all(np.isfinite(r0["weights"])) and r0["weights"].size == r0["counts"].size > 5
