#!/usr/bin/env python
"""bench.py -- particle-steps/s of the bootstrap particle filter on MI355X.

Workload (BASELINE.json metric config, SURVEY.md 8d "C2-m4"): ssm_ung, negative-binomial observations,
m = k = 4 states, T = 1000 time points, N = 65 536 particles, full-output bootstrap filter (particle
history, ancestor table and weights all kept in HBM, as bootstrap_filter() returns them).  One "step" =
one complete filter pass (N*T = 65.5 M particle-steps) with the model tables already resident in HBM.

A single filter does not shard (resampling is a global exchange every time step), and on
its own it is latency-bound (1000 dependent launches with one wave per SIMD), so a step advances
`--replicas` independent filter calls (different Philox streams: repeated logLik / suggest_N / parallel
chains) together -- SURVEY.md 8d's regime B on the metric's own configuration; the single-call number
(regime A) is reported next to it as `single_filter`.  --gpus N runs the same batch on every GPU
("replicas only"), one process per GPU, and reports the aggregate.

Extra objects in the JSON line:
  roofline      dominant kernel (pf_grid_step_v2): algorithmic bytes per launch = N * (16 m + 24) B
                 / its average launch duration measured with HIP events on the launch stream
  cpu_baseline  the CPU oracle (single-threaded per filter, like the reference) timed on a bounded sample
  single_filter one filter call alone (latency regime)
  regime_b      512 independent N=2048 filters, one workgroup each, state resident in LDS (the shape IS
                post-correction uses)
  mng_c4, post_correct (+ post_correct_scaling.projected: one rank's share at 1 / 2 / 4 / 8 ranks, measured on this GPU),
  big_states (17 ... 64 state dimensions), dropin (the C ABI with host buffers), accuracy (same seeds against the CPU oracle)
"""
import argparse
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# fp64 VALU issue roof: 256 CUs x 4 SIMDs, one wave64 instruction per 4 cycles (16 lanes per clock) at 2.4 GHz
VALU_PEAK_WAVE_INSTR_S = 256 * 4 * 2.4e9 / 4
# wave-level VALU instructions per particle-step of the workgroup-resident kernels, from the PMC passes committed under
# profiles/r05_pmc_*.txt (SQ_INSTS_VALU per dispatch / particle-steps per dispatch; scripts/profile_r05.sh; regime B: profiles/r04_pmc_regb_v2.txt);
# fallbacks only: the run measures them live (live_valu_counts)
REGB_VALU_PER_PSTEP = 10595336946.0 / (512 * 2048 * 1000.0)   # pf_wg_kernel<4,4,2,false,1024> (profiles/r04_pmc_regb_v2.txt; the 512-thread kernel of rounds 3-4: 9.59e9, r04_pmc_regb.txt)
C5_VALU_PER_PSTEP = 16856741689.0 / (2304.0 * 4096 * 300.0)   # pf_wg_kernel<2,2,4,1,1024,1,exact> (nlg_growth, psi-APF), one dispatch of 2 304 of the 2 340 draws -- the other 36 take the row family (profiles/r05_pmc_C5.txt; round 4: 17.73e9 for 2 340)
C5_KERNEL = "pf_wg_kernel<2,2,4,1,1024,1,exact>"
HEAD_VALU_PER_PSTEP = 7874668.9 / (16 * 65536.0)                # pf_grid_step_v2<4,4,4,false,true,true>: 7.87 M per launch of 16 filters x 65 536 particles (profiles/r05_pmc_sq1_summary.txt; round 4: 7.81 M, round 3: 8.36 M)
C4_VALU_PER_PSTEP = 6658864.9 / (32 * 16384.0)                 # pf_grid_step_v2<8,8,2,true,...>: SQ_INSTS_VALU per launch of 32 filters x 16 384 particles (profiles/r05_pmc_C4.txt; round 4: 6.58 M, round 3: 7.14 M)


def valu_roofline(valu_per_pstep, psteps_per_s, kernel, clock=None, key=None):
    valu_source = "committed"
    if key is not None:
        valu_per_pstep, valu_source = valu_count(key, valu_per_pstep)
    a = valu_per_pstep * psteps_per_s
    out = {"bound": "valu_fp64_issue", "kernel": kernel, "achieved": a / 1e9, "peak": VALU_PEAK_WAVE_INSTR_S / 1e9,
           "unit": "G wave-instr/s", "frac": a / VALU_PEAK_WAVE_INSTR_S, "valu_wave_instr_per_particle_step": valu_per_pstep,
           "lane_instr_per_particle_step": valu_per_pstep * 64, "valu_source": valu_source,
           "source": "instruction count: rocprofv3 --pmc SQ_INSTS_VALU (profiles/r05_pmc_sq1_summary.txt, r05_pmc_C4.txt, r05_pmc_C5.txt, r04_pmc_regb_v2.txt); rate: measured live"}
    if clock and clock.get("sclk_mhz"):
        # the issue roof at the engine clock the kernel actually ran at (the socket power limit holds it under 2.4 GHz)
        out["frac_at_sampled_clock"] = a / (256 * 4 * clock["sclk_mhz"] * 1e6 / 4)
    return out


def clock_power_while(run_once, samples=3):
    """Engine clock (MHz) and socket power (W) read through rocm-smi while `run_once` repeats on another thread: the median of
    `samples` readings, the first one after a second of load.  Outside every timed region; None where rocm-smi is missing."""
    import shutil
    import subprocess
    import threading
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(exe):
        return None
    stop = threading.Event()

    def load():
        k = 0
        while not stop.is_set():
            run_once(k)
            k += 1

    th = threading.Thread(target=load, daemon=True)
    th.start()
    got = []
    try:
        time.sleep(1.0)
        for _ in range(samples):
            try:
                txt = subprocess.run([exe, "-d", os.environ.get("LOCAL_RANK", "0"), "--showclocks", "--showpower"],
                                     capture_output=True, text=True, timeout=20).stdout
            except Exception:  # noqa: BLE001
                continue
            ck = re.search(r"sclk clock level:[^(]*\((\d+)Mhz\)", txt)
            pw = re.search(r"Power \(W\):\s*([0-9.]+)", txt)
            if ck:
                got.append((float(ck.group(1)), float(pw.group(1)) if pw else float("nan")))
    finally:
        stop.set()
        th.join()
    if not got:
        return None
    got.sort()
    mid = got[len(got) // 2]
    return {"sclk_mhz": mid[0], "socket_power_w": mid[1], "samples": [list(g) for g in got],
            "source": "rocm-smi --showclocks --showpower, sampled while the same plan repeats on another thread (outside the timed region)"}


HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def pmc_traffic(R, N, ppt, kernel, factored):
    """HBM bytes per pf_grid_step launch from the PMC passes committed for THIS round's kernel (profiles/r05_pmc_traffic*.json:
    separate --pmc FETCH_SIZE / WRITE_SIZE runs of this script by scripts/profile_r03.sh, corrected as MI355X_MICROARCH.md
    prescribes and calibrated on calib_copy_kernel's known byte count in the same process); None unless the profile was
    taken on exactly this workload, kernel variant and weights layout (a counter pass cannot run inside a timed run)."""
    name = "r05_pmc_traffic.json" if factored else "r05_pmc_traffic_reference_layout.json"
    try:
        p = json.load(open(os.path.join(ROOT, "profiles", name)))
        same_kernel = p["kernel"].replace(" ", "").endswith(kernel.replace(" ", ""))
        if p["replicas"] == R and p["particles"] == N and p["particles_per_thread"] == ppt and same_kernel:
            return p["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def committed_traffic(kernel):
    """HBM bytes per launch of a kernel from the round's committed counter passes (profiles/r06_pmc_traffic_big_c4.json: separate --pmc FETCH_SIZE /
    WRITE_SIZE runs, FETCH factor calibrated on calib_copy_kernel); None when the profile has no such kernel"""
    try:
        p = json.load(open(os.path.join(ROOT, "profiles", "r06_pmc_traffic_big_c4.json")))
        e = p.get(kernel.replace(" ", ""))
        if e:
            return {"traffic": e["hbm_bytes_per_launch"], "traffic_over_algorithmic": e["hbm_bytes_per_launch"] / e["algorithmic_bytes_per_launch"],
                    "traffic_particles_per_launch": e["particles_per_launch"], "traffic_source": p["source"]}
    except (OSError, KeyError, ValueError):
        pass
    return None


def live_pmc_traffic(args, kernel, layout_factored, timeout=150):
    """HBM bytes per launch of the dominant kernel, MEASURED during this run: two short `rocprofv3 --pmc` passes of this very
    script (FETCH_SIZE, then WRITE_SIZE -- separate passes, as MI355X_MICROARCH.md prescribes; one timed step of the same batch, nothing
    else), the gfx950 FETCH_SIZE factor calibrated in the same process on calib_copy_kernel (2 GiB read + 2 GiB written).  Outside
    every timed region.  None when rocprofv3 is missing, a pass fails or takes too long (the committed profile value is used then)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    base = [sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "0", "--replicas", str(args.replicas), "--particles",
            str(args.particles), "--T", str(args.T), "--series", args.series, "--no-cpu-baseline", "--no-single", "--no-regime-b",
            "--no-post-correct", "--no-dropin", "--no-accuracy", "--no-side-runs", "--no-clock", "--no-mng", "--no-live-pmc"]
    if layout_factored:
        base.append("--factored-weights-layout")
    env = dict(os.environ, BSSM_BENCH_CALIB="1", TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    got = {}
    tmp = tempfile.mkdtemp(prefix="bssm_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out_dir = os.path.join(tmp, ctr)
            cp = subprocess.run([exe, "--pmc", ctr, "--output-format", "csv", "-d", out_dir, "--"] + base, cwd="/tmp", env=env,
                                capture_output=True, text=True, timeout=timeout)
            if cp.returncode != 0:
                return None
            acc = {}
            for path in glob.glob(out_dir + "/**/*counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(path)):
                    if row["Counter_Name"] != ctr:
                        continue
                    a = acc.setdefault(row["Kernel_Name"].split("(")[0], [0.0, 0])
                    a[0] += float(row["Counter_Value"]); a[1] += 1
            step = [k for k in acc if kernel.replace(" ", "") in k.replace(" ", "")]
            cal = [k for k in acc if "calib_copy" in k]
            if not step or not cal:
                return None
            got[ctr] = (acc[step[0]][0] / acc[step[0]][1], acc[cal[0]][0] / acc[cal[0]][1], acc[step[0]][1])
    except Exception:  # noqa: BLE001 -- a measurement aid: never fails the benchmark
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    known = float((1 << 28) * 8)
    rf = known / (got["FETCH_SIZE"][1] * 1024.0)
    wf = known / (got["WRITE_SIZE"][1] * 1024.0)
    wf_used = 1.0 if abs(wf - 1.0) < 0.05 else wf
    rd, wr = got["FETCH_SIZE"][0] * 1024.0 * rf, got["WRITE_SIZE"][0] * 1024.0 * wf_used
    return {"hbm_bytes_per_launch": rd + wr, "read": rd, "written": wr, "read_factor_calibrated": rf, "write_factor_measured": wf,
            "launches_sampled": int(min(got["FETCH_SIZE"][2], got["WRITE_SIZE"][2]))}


def live_valu_counts(args, timeout=240):
    """Wave-level VALU instructions per particle-step of the kernels the VALU rooflines of this line are quoted on, MEASURED
    during this run: ONE `rocprofv3 --pmc SQ_INSTS_VALU` pass of this very script (one step of the metric batch, regime B, C4, the
    C5 post-correction; nothing else), the count of a dispatch divided by the particle-steps it advanced (Grid_Size of the
    dispatch x particles per thread x time points per launch).  Outside every timed region.  None when rocprofv3 is missing or
    the pass fails: the constants committed with the round's profiles (above) are used then, and `valu_source` says which."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    base = [sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "0", "--replicas", str(args.replicas), "--particles",
            str(args.particles), "--T", str(args.T), "--series", args.series, "--no-cpu-baseline", "--no-single", "--no-dropin",
            "--no-accuracy", "--no-side-runs", "--no-clock", "--no-live-pmc", "--no-live-valu", "--pc-c5-only"]
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    tmp = tempfile.mkdtemp(prefix="bssm_valu_", dir="/tmp")
    acc = {}
    try:
        cp = subprocess.run([exe, "--pmc", "SQ_INSTS_VALU", "--output-format", "csv", "-d", tmp, "--"] + base, cwd="/tmp", env=env,
                            capture_output=True, text=True, timeout=timeout)
        if cp.returncode != 0:
            return None
        for path in glob.glob(tmp + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(path)):
                if row["Counter_Name"] != "SQ_INSTS_VALU":
                    continue
                name = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("bssm::", "").replace(" ", "")
                a = acc.setdefault(name, [0.0, 0.0, 0])   # instructions, work-items, dispatches
                a[0] += float(row["Counter_Value"]); a[1] += float(row["Grid_Size"]); a[2] += 1
    except Exception:  # noqa: BLE001 -- a measurement aid: never fails the benchmark
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)

    def per_pstep(kernel, particles_per_thread, steps_per_launch):
        a = acc.get(kernel)
        return a[0] / (a[1] * particles_per_thread * steps_per_launch) if a and a[1] > 0 else None
    got = {"head": per_pstep("pf_grid_step_v2<4,4,4,false,true,true>", 4, 1), "c4": per_pstep("pf_grid_step_v2<8,8,2,true,true,true>", 2, 1),
           "regb": per_pstep("pf_wg_kernel<4,4,2,0,1024,-1,false>", 2, args.T), "c5": per_pstep("pf_wg_kernel<2,2,4,1,1024,1,true>", 4, 300),
           "c5_row": per_pstep("pf_row_step<2,2,1,1>", 1, 1)}
    got["dispatches"] = {k: v[2] for k, v in acc.items() if k.startswith("pf_")}
    return got


LIVE_VALU = None   # filled once per run by live_valu_counts (rank 0, one GPU)


def valu_count(key, committed):
    """(instructions per particle-step, where the number comes from)"""
    v = (LIVE_VALU or {}).get(key)
    return (v, "live: rocprofv3 --pmc SQ_INSTS_VALU pass of this run") if v else (committed, "committed: profiles/r04_pmc_* (the live pass did not run)")


def _c5_job(ns):
    """ssm_nlg growth model, T = 300, m = 2; `ns` retained draws around theta-hat"""
    import numpy as np
    from bssm_amd import models as M
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
    theta = th0 + rng.normal(0, 0.05, (ns, 3))
    counts = rng.geometric(0.234, ns).astype(np.uint32)
    return model, theta, counts, n


def post_correct_phase(rank, world, dist, cdev="cuda", reps=3, cpu_baseline=False, ns=2340, label="C5"):
    """SURVEY.md 8d config C5 (north_star: IS post-correction sharded over the GPUs of a node): `ns` retained draws of
    the ssm_nlg growth model, psi-APF N = 4096, T = 300, split into contiguous blocks over the ranks (strong scaling of
    one job), log-weights all-gathered over RCCL.  Every rank takes part in every collective whatever happens locally
    (a local failure is flagged through an all-reduce first), so this phase cannot dead-lock the benchmark.

    per_rank_ms (what each number is, all wall-clock ms of the best repetition):
      host_pack_ms     per-draw packing on the caller's host threads (theta, a1, P1 -> one record per draw)
      setup_call_ms    the per-chunk device set-up calls: staging copy, uploads, fn_setup_kernel launch and, on calls of
                       more than 4 096 draws, the wait for that kernel (the pipeline paces itself on it)
      filter_ms        busy time of the thread that runs the filters (pf_wg_kernel launches + log-likelihood read-back);
                       overlaps the two above: the pipeline prepares chunk c + 1 while chunk c filters
      local_ms         wall time of this rank's whole block (= the pipeline's critical path)
      wait_ms          waiting for the slowest rank at the barrier in front of the collective (load imbalance)
      collective_ms    the all-reduce of the status flag + ONE all-gather of the padded log-weight blocks, after the barrier"""
    import numpy as np
    import torch
    from bssm_amd import filters as F, post_correct as PC
    nsim = 4096
    model, theta, counts, n = _c5_job(ns)
    bounds = PC.shard_bounds(np.ones(ns), world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    threads = max(1, min(32, (os.cpu_count() or 8) // max(1, world)))  # the host cores are shared by the ranks of the node
    times, splits, weights, err = [], [], None, ""
    for it in range(reps + 1):  # first pass warms the arena and the module
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loc_w = np.zeros(hi - lo)
        ok = 1.0
        split = [0.0] * 6  # host_pack, setup_call, filter, local, wait, collective
        try:
            if it == 0:  # the stored modes: here the mode at theta-hat for every draw (approximate_for_is re-linearises there)
                mode = F.psi_setup(model)["mode"].reshape(-1, order="F")
                modes = np.repeat(mode[:, None], hi - lo, axis=1)
            if hi > lo:
                loc = PC._run_local_block_nlg(model, theta[lo:hi], modes, counts[lo:hi], lo, ns, nsim, 1, 2, 3, threads)
                loc_w = np.log(loc["weights"])
                tm = loc["timing_ms"]
                split[0], split[1], split[2] = tm["host_pack"], tm["setup_call"], tm["filter"]
        except Exception as e:  # noqa: BLE001 -- reported, never raised past the collectives
            ok, err = 0.0, repr(e)[:200]
        split[3] = (time.perf_counter() - t0) * 1e3
        if dist is not None:
            tw = time.perf_counter()
            dist.barrier()
            tc = time.perf_counter()
            split[4] = (tc - tw) * 1e3
            flag = torch.tensor([ok], device=cdev, dtype=torch.float64)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = float(flag.item())
            mx = int(np.diff(bounds).max())
            pad = torch.zeros(mx, device=cdev, dtype=torch.float64)
            pad[:hi - lo] = torch.from_numpy(loc_w).to(cdev)
            outs = [torch.empty_like(pad) for _ in range(world)]
            dist.all_gather(outs, pad)
            weights = np.concatenate([outs[q].cpu().numpy()[:int(bounds[q + 1] - bounds[q])] for q in range(world)])
            split[5] = (time.perf_counter() - tc) * 1e3
        else:
            weights = loc_w
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([el] + split, device=cdev, dtype=torch.float64)
            allt = [torch.empty_like(tt) for _ in range(world)]
            dist.all_gather(allt, tt)
            el = max(float(x[0]) for x in allt)
            split_all = [[float(v) for v in x[1:]] for x in allt]
        else:
            split_all = [split]
        if ok < 1.0:
            return {"error": err or "a rank failed"}
        if it > 0:
            times.append(el)
            splits.append(split_all)
    best = int(np.argmin(times))
    extra = {}
    if cpu_baseline and world == 1:
        # the reference-equivalent CPU path of THIS phase: one psi-APF per draw, OpenMP static schedule over the draws on
        # all host cores (src/approx_mcmc.cpp:364-369), on a bounded sample of the same draws; also the parity check
        from oracle import oracle as O
        cores = os.cpu_count() or 1
        sample = min(ns, max(64, 2 * cores))
        t0 = time.perf_counter()
        ref = O.fn_is_logweights(model, theta[:sample], modes[:, :sample], nsim, seed=1, first_stream=0, n_threads=cores)
        ec = time.perf_counter() - t0
        extra["cpu_baseline"] = {"value": sample * float(nsim) * n / ec, "unit": "particle-steps/s", "cores": cores, "kind": "port",
                                 "sample": "%d of the %d draws, OpenMP schedule(static) over draws with %d threads, gcc -O2, %.1f s"
                                           % (sample, ns, cores, ec)}
        extra["parity_max_abs_dlogw_vs_oracle"] = float(np.max(np.abs(weights[:sample] - ref)))
    filt = max(v[2] for v in splits[best])
    return {**extra, "workload": "%s: ssm_nlg growth, T=300, m=2, %d retained draws, psi-APF N=%d, IS2 weights (output theta), draws split in "
                        "contiguous blocks over the ranks, log-weights all-gathered (RCCL)" % (label, ns, nsim),
            "scaling": "strong", "n_gpus": world, "seconds": times[best], "draws_per_s": ns / times[best],
            "value": ns * float(nsim) * n / times[best], "unit": "particle-steps/s", "host_threads_per_rank": threads,
            "roofline_valu": valu_roofline(C5_VALU_PER_PSTEP, (float(np.diff(bounds).max()) * world) * float(nsim) * n / (filt * 1e-3),
                                           C5_KERNEL, key="c5") if filt > 0 else None,
            "per_rank_ms": [{"rank": q, "draws": int(bounds[q + 1] - bounds[q]), "host_pack_ms": round(v[0], 2),
                             "setup_call_ms": round(v[1], 2), "filter_ms": round(v[2], 2), "local_ms": round(v[3], 2),
                             "wait_ms": round(v[4], 2), "collective_ms": round(v[5], 2)}
                            for q, v in enumerate(splits[best])],
            "per_rank_note": "host_pack / setup_call run on the caller's threads and overlap filter_ms (the filter thread) inside a rank; "
                             "local_ms = the rank's whole block; wait_ms = barrier in front of the collective (waiting for the slowest "
                             "rank); collective_ms = status all-reduce + one all-gather after that barrier",
            "finite_weights": bool(np.all(np.isfinite(weights))), "checksum_logw": float(np.sum(weights))}


_SINGLE_PROCESS_SNIPPET = r"""
import json, os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import bench
from bssm_amd import filters as F, post_correct as PC
ns, nsim = int(sys.argv[1]), 4096
devs = json.loads(sys.argv[2])
model, theta, counts, n = bench._c5_job(ns)
mode = F.psi_setup(model)["mode"].reshape(-1, order="F")
modes = np.repeat(mode[:, None], ns, axis=1)
threads = max(1, min(64, (os.cpu_count() or 8)))
res = {}
for devices in (None, devs):
    best, tm = 1e9, None
    for it in range(3):
        t0 = time.perf_counter()
        loc = PC._run_local_block_nlg(model, theta, modes, counts, 0, ns, nsim, 1, 2, 3, threads, devices=devices)
        el = time.perf_counter() - t0
        if it and el < best:
            best, tm = el, loc["timing_ms"]
    res["one_device" if devices is None else "all_devices"] = {"seconds": best, "timing": tm, "checksum_logw": float(np.sum(np.log(loc["weights"])))}
print("RESULT " + json.dumps(res))
"""


def single_process_multi_device(ns, devices=None):
    """The in-library sharding (bssm_is_correction_nlg_multi: one host thread + hipSetDevice per GPU, one RCCL all-gather
    from ncclCommInitAll) in a child process with a time limit so that a collective that cannot start never blocks the
    benchmark.  devices = HIP ordinals of the shards (the same ordinal several times = logical shards on one GPU, exchanged
    by device copies); None = every GPU this process can see, skipped when that is one."""
    import subprocess
    import sys
    import torch
    nd = torch.cuda.device_count()
    if devices is None:
        if nd < 2:
            return {"skipped": "one visible GPU", "visible_gpus": nd}
        devices = list(range(nd))
    try:
        cp = subprocess.run([sys.executable, "-c", _SINGLE_PROCESS_SNIPPET, str(ns), json.dumps(list(devices))], capture_output=True,
                            text=True, timeout=150, cwd=os.path.dirname(os.path.abspath(__file__)))
    except subprocess.TimeoutExpired:
        return {"error": "timeout (150 s)", "visible_gpus": nd}
    for line in cp.stdout.splitlines():
        if line.startswith("RESULT "):
            r = json.loads(line[7:])
            r["visible_gpus"] = nd
            r["devices"] = list(devices)
            r["draws"] = ns
            a, b = r["one_device"], r["all_devices"]
            r["gather"] = (b.get("timing") or {}).get("gather")
            r["speedup"] = a["seconds"] / b["seconds"]
            r["identical_weights"] = a["checksum_logw"] == b["checksum_logw"]
            return r
    return {"error": (cp.stderr or cp.stdout)[-300:], "visible_gpus": nd}


def mng_post_correct_leg(ns=512, n=200, nsim=256, threads=8):
    """One psi post-correction call over `ns` theta-draws of the C4-shaped ssm_mng model (T = n), device set-up (the default)."""
    import numpy as np
    from bssm_amd import filters as F, models as M, post_correct as PC
    from models_zoo import mng_c4
    base = mng_c4(n)
    m, k = int(base.m), np.shape(base["R"])[1]
    R0 = np.array(base["R"], dtype=float).reshape(m, k)
    model = M.BssmModel(base.classes, **{**base, "update_fn": (lambda theta: dict(R=R0 * np.exp(theta[0]))), "theta": np.zeros(1)})
    rng = np.random.default_rng(5)
    theta = rng.normal(0, 0.1, (ns, 1))
    counts = np.ones(ns, dtype=np.uint32)
    mode = F.psi_setup(model)["mode"].reshape(-1, order="F")
    modes = np.repeat(mode[:, None], ns, axis=1)
    arrs = PC.expand_models(model, theta)
    best, r = 1e9, None
    for _ in range(3):
        t0 = time.perf_counter()
        r = PC._run_local_block(arrs, modes, counts, 0, ns, nsim, 1, 2, 3, threads)
        best = min(best, time.perf_counter() - t0)
    assert np.all(np.isfinite(r["loglik"]))
    return {"workload": "ssm_mng C4 shape (m=8, k=8, p=4), T=%d, %d theta-draws, psi-APF N=%d, is2, output theta; one bssm_is_correction call, "
                        "%d host threads for ingesting / packing the draws" % (n, ns, nsim, threads),
            "seconds": best, "draws_per_s": ns / best,
            "setup": "device: mng_laplace_kernel, mng_setup_kernel<8, rows> (eight lanes per draw), mng_factors_kernel, mng_tables_kernel",
            "checksum_loglik": float(np.sum(r["loglik"])),
            "round_history_s": {"round 2 (one kernel, one lane per draw)": 0.26, "host set-up, 32 threads": 0.10}}


def seasonal_post_correct_leg(period, ns=256, nsim=64, threads=8):
    """One psi post-correction call over `ns` theta-draws of bsm_ng(period = 24 / 52) -- 25 / 53 states: the set-up (Gaussian approximation,
    smoother, conditional factors, tables) on the device with the states padded to 32 / 64 (lg_smoother_wave.h, lg_factors_rows.h); the host
    set-up timed beside it on 8 draws"""
    import numpy as np
    from bssm_amd import filters as F, models as M, post_correct as PC
    rng = np.random.default_rng(5)
    n, m = 3 * period if period == 24 else 2 * period, period + 1
    y = rng.poisson(np.exp(1.6 + 0.4 * np.sin(2 * np.pi * np.arange(n) / period))).astype(float)
    y[7] = np.nan
    model = M.bsm_ng(y, sd_level=M.halfnormal(0.1, 1), sd_slope=M.halfnormal(0.02, 1), sd_seasonal=M.halfnormal(0.05, 1), period=period,
                     distribution="poisson", P1=np.diag([1.0, 0.1] + [0.5] * (period - 1)), a1=np.array([1.6] + [0.0] * period))
    mode = np.asarray(F.psi_setup(model)["mode"]).reshape(-1, 1)

    def job(k):
        theta = np.column_stack([rng.uniform(0.05, 0.2, k), rng.uniform(0.01, 0.03, k), rng.uniform(0.02, 0.08, k)])
        return dict(theta=theta, counts=np.ones(k, dtype=np.uint32), modes=np.repeat(mode, k, axis=1) + rng.normal(0, 0.03, (n, k)), output_type=3)

    def timed(mo, reps):
        best, r = 1e9, None
        for _ in range(reps):
            t0 = time.perf_counter()
            r = PC.post_correct(model, mo, nsim, threads=threads, is_type="is2", seed=6)
            best = min(best, time.perf_counter() - t0)
        return best, r
    sec, r = timed(job(ns), 2)
    assert np.all(np.isfinite(r["loglik"]))
    small = job(8)
    dev8, r8 = timed(small, 2)
    prev = os.environ.get("BSSM_AMD_DEVICE_SETUP")
    os.environ["BSSM_AMD_DEVICE_SETUP"] = "0"
    try:
        host8, h8 = timed(small, 1)
    finally:
        if prev is None:
            del os.environ["BSSM_AMD_DEVICE_SETUP"]
        else:
            os.environ["BSSM_AMD_DEVICE_SETUP"] = prev
    dll = float(np.max(np.abs(r8["loglik"] - h8["loglik"])))
    assert dll <= 1e-9 * n, "device and host psi set-up differ by %.3g in a log-likelihood (bar: 1e-9 T, BASELINE.md section 3)" % dll
    return {"workload": "bsm_ng(period = %d): m = %d, T = %d, %d theta-draws, psi-APF N = %d, is2, output theta; one bssm_is_correction call, %d host threads"
                        % (period, m, n, ns, nsim, threads),
            "seconds": sec, "draws_per_s": ns / sec,
            "library_split_ms": {k: round(float(v), 3) for k, v in (r.get("timing_ms") or {}).items()},   # (of the last timed call: the C ABI call's own wall and its parts; the rest of `seconds` is the Python mirror)
            "setup": "device: ung_setup_kernel<%d> (Laplace step, tables), ung_smooth_wave_kernel<%d> (products on the matrix cores), ung_factors_thin_kernel<%d> (time points >= 1) + ung_factors_rows_kernel<%d> (t = 0), setup_tables_big_kernel<%d>" % ((32 if m <= 32 else 64,) * 5),
            "eight_draws_device_s": dev8, "eight_draws_host_setup_s": host8,
            "max_abs_dloglik_device_vs_host_setup": float(np.max(np.abs(r8["loglik"] - h8["loglik"])))}


def mng_big_post_correct_leg(m=20, p=4, n=100, ns=512, nsim=64, threads=8):
    """Round 6: one psi post-correction call over `ns` theta-draws of an ssm_mng model with m = 20 states (padded to 32) and p = 4 series: the
    set-up on the device (mng_laplace_kernel, mng_smooth_wave_kernel<32> over the n p virtual steps, mng_factors_rows_kernel<32>, mng_tables_kernel<32>;
    src/model_ssm_mlg.cpp:90-137, 310-396), the host set-up timed beside it on 8 draws"""
    import numpy as np
    from bssm_amd import models as M, post_correct as PC
    from bssm_amd.filters import psi_setup
    rng = np.random.default_rng(120)
    k = 3
    Q = np.linalg.qr(rng.normal(size=(m, m)))[0]
    Z = rng.normal(0, 0.4, (p, m))
    dists = ["poisson", "negative binomial", "gaussian", "poisson"][:p]
    phis = [1.0, 5.0, 0.6, 1.0][:p]
    y = np.column_stack([rng.poisson(3, n) if d == "poisson" else (rng.negative_binomial(5, 0.6, n) if d == "negative binomial" else rng.normal(0, 0.8, n))
                         for d in dists]).astype(float)
    R0 = 0.15 * rng.normal(size=(m, k))
    base = M.ssm_mng(y, Z, 0.9 * Q, R0, a1=np.zeros(m), P1=np.diag(0.05 + 0.002 * np.arange(m)), distribution=dists, phi=phis)
    model = M.BssmModel(base.classes, **{**base, "update_fn": (lambda th: dict(R=R0 * np.exp(th[0]))), "theta": np.zeros(1)})
    mode = np.asarray(psi_setup(model)["mode"]).reshape(-1, 1, order="F")

    def job(q):
        return dict(theta=rng.normal(0, 0.1, (q, 1)), counts=np.ones(q, dtype=np.uint32), modes=np.repeat(mode, q, axis=1) + rng.normal(0, 0.03, (n * p, q)), output_type=3)

    def timed(mo, reps):
        best, r = 1e9, None
        for _ in range(reps):
            t0 = time.perf_counter()
            r = PC.post_correct(model, mo, nsim, threads=threads, is_type="is2", seed=6)
            best = min(best, time.perf_counter() - t0)
        return best, r
    sec, r = timed(job(ns), 2)
    assert np.all(np.isfinite(r["loglik"]))
    small = job(8)
    dev8, r8 = timed(small, 2)
    prev = os.environ.get("BSSM_AMD_DEVICE_SETUP")
    os.environ["BSSM_AMD_DEVICE_SETUP"] = "0"
    try:
        host8, h8 = timed(small, 1)
    finally:
        if prev is None:
            del os.environ["BSSM_AMD_DEVICE_SETUP"]
        else:
            os.environ["BSSM_AMD_DEVICE_SETUP"] = prev
    dll = float(np.max(np.abs(r8["loglik"] - h8["loglik"])))
    assert dll <= 1e-9 * n, "device and host psi set-up differ by %.3g in a log-likelihood (bar: 1e-9 T, BASELINE.md section 3)" % dll
    return {"workload": "ssm_mng: m = %d, p = %d, k = %d, T = %d, %d theta-draws, psi-APF N = %d, is2, output theta; one bssm_is_correction call, %d host threads"
                        % (m, p, k, n, ns, nsim, threads),
            "seconds": sec, "draws_per_s": ns / sec, "eight_draws_device_s": dev8, "eight_draws_host_setup_s": host8,
            "max_abs_dloglik_device_vs_host_setup": float(np.max(np.abs(r8["loglik"] - h8["loglik"])))}


def accuracy_leg(model, N, T, R, m, fw, stream, max_pairs=64):
    """The metric's second half: log-likelihood RMSE (and weighted state means, states, weights, ancestors) against the
    reference-equivalent CPU path on identical seeds, on the very variant that was timed: the same plan shape (N, T, R
    filters in flight, two launch sequences), two passes with fresh seeds = 2 R (seed, batch slot) pairs, EVERY one of
    them re-run by the oracle (one N x T filter, ~8 s of one core, per host thread).

    The oracle runs in tie-following mode (oracle/bssm_oracle.c): it resamples exactly as the reference does and compares
    its ancestors with the device's; where they differ it adopts the device's choice only if that is the adjacent particle
    and the slot's target lies within tie_tolerance(N) strata of the cumulative weight between the two (either answer is a
    correctly rounded evaluation of the same comparison) -- anything else is counted as unexplained and breaks the
    comparison.  The oracle is then the reference recursion along the device's own path and everything is compared at the
    fp64 bar on every pair.  For the pairs that had a tie, a free-running oracle is run as well: what its log-likelihood
    differs by is Monte-Carlo noise of two equally valid particle systems, reported separately."""
    import numpy as np
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from bssm_amd import models as M
    from bssm_amd.plan import FilterPlan
    from oracle import oracle as O
    arr = M.model_arrays(model)
    acc_seeds = (20011, 20012)
    keys = [(sd, f) for sd in acc_seeds for f in range(R)][:max_pairs] if max_pairs >= 2 * R else \
        [(sd, f) for sd in acc_seeds for f in sorted(set([0, max(0, R // 2 - 1), R // 2, R - 1]))][:max(1, max_pairs)]
    try:
        avail_gb = int(re.search(r"MemAvailable:\s+(\d+)", open("/proc/meminfo").read()).group(1)) / 1e6
    except Exception:  # noqa: BLE001
        avail_gb = 64.0
    per_pair_gb = 2.6 * (8.0 * ((m + 1) * (T + 1) + T / 2.0) * N) / 1e9   # device copy + oracle outputs + temporaries
    workers = int(max(1, min(len(keys), (os.cpu_count() or 1) // 2, (0.6 * avail_gb) // per_pair_gb)))
    pa = FilterPlan(model, N, method="bsf", keep_history=True, force="grid", n_rep=R, factored_weights=fw)

    def one(job):
        sd, f, got, gll = job
        with O.following(got["indices"], N) as fo:
            ref = O.bsf_filter(arr, N, seed=sd, stream=f)
        rec = {"seed": sd, "slot": f, "dloglik": gll - float(ref["logLik"]), "ties": fo.ties, "unexplained": fo.unexplained,
               "tables_identical_after_following": bool(np.array_equal(got["indices"], ref["indices"].T))}
        w = got["weights"][:T]
        att_gpu = np.einsum("tmn,tn->tm", got["alpha"][:T], w / w.sum(axis=1, keepdims=True))   # weighted filtered means, T x m
        rw = ref["weights"][:, :T]
        att_ref = np.einsum("mtn,nt->tm", ref["alpha"][:, :T, :], rw / rw.sum(axis=0))
        rec["att_max_rel_diff"] = float(np.max(np.abs(att_gpu - att_ref) / (1.0 + np.abs(att_ref))))
        ra = np.transpose(ref["alpha"], (1, 0, 2))
        rec["alpha_max_abs_diff_over_scale"] = float(np.max(np.abs(got["alpha"] - ra)) / max(1.0, float(np.abs(ra).max())))
        rec["weights_max_rel_diff"] = float(np.max(np.abs(got["weights"] - ref["weights"].T) / (1e-300 + np.abs(ref["weights"].T))))
        if fo.ties:
            t0, j0 = fo.first_tie
            rec.update(first_tie_t=t0, first_tie_slot=j0, max_margin_strata=fo.max_margin)
            del ref
            free = O.bsf_filter(arr, N, seed=sd, stream=f)     # the oracle left alone: diverges at the first tie
            neq = (got["indices"] != free["indices"].T).sum(axis=1)
            rec["free_running_dloglik"] = gll - float(free["logLik"])
            rec["free_running_differing_ancestors_by_step_from_first_tie"] = [int(v) for v in neq[t0:t0 + 5]]
            rec["free_running_differing_entries"] = int(neq.sum())
        return rec

    recs, dev_ll = [], {}
    with ThreadPoolExecutor(max_workers=workers) as ex:
        for sd in acc_seeds:
            mine = [k for k in keys if k[0] == sd]
            if not mine:
                continue
            pa.run(sd, stream0=0, hip_stream=stream)
            llv = pa.loglik(stream)
            dev_ll[sd] = [float(v) for v in llv]
            for g in range(0, len(mine), workers):
                jobs = [(sd, f, pa.fetch(f, hip_stream=stream), float(llv[f])) for _, f in mine[g:g + workers]]
                recs += list(ex.map(one, jobs))
                del jobs
    acc_kernel, acc_seqs, acc_layout = FilterPlan.last_kernel(), pa.launch_sequences(), pa.weights_layout()
    pa.close()
    d = np.asarray([r["dloglik"] for r in recs])
    tied = [r for r in recs if r["ties"]]
    free = np.asarray([r.get("free_running_dloglik", r["dloglik"]) for r in recs])
    sd_est = float(np.mean([np.std(v, ddof=1) for v in dev_ll.values() if len(v) > 1])) if dev_ll else None
    return {"loglik_rmse_vs_cpu_same_seeds": float(np.sqrt(np.mean(d * d))), "max_abs_diff": float(np.max(np.abs(d))),
            "oracle_mode": "tie-following (adopts the device's ancestor only at explained last-ulp ties)",
            "device_pairs": len(acc_seeds) * R, "oracle_checked_pairs": len(recs), "oracle_threads": workers,
            "pairs_without_a_tie": len(recs) - len(tied), "pairs_with_ties": len(tied), "ties_total": int(sum(r["ties"] for r in recs)),
            "unexplained_ancestor_differences": int(sum(r["unexplained"] for r in recs)),
            "ancestor_tables_identical_after_following": int(sum(r["tables_identical_after_following"] for r in recs)),
            "tie_tolerance_strata": O.tie_tolerance(N), "max_tie_margin_strata": max([r["max_margin_strata"] for r in tied], default=0.0),
            "weighted_state_means_max_rel_diff": max(r["att_max_rel_diff"] for r in recs),
            "states_max_abs_diff_over_scale": max(r["alpha_max_abs_diff_over_scale"] for r in recs),
            "weights_max_rel_diff": max(r["weights_max_rel_diff"] for r in recs),
            "loglik_rmse_free_running_oracle": float(np.sqrt(np.mean(free * free))),
            "loglik_sd_of_the_estimator_across_streams": sd_est,
            "tied_pairs": [{k: r[k] for k in ("seed", "slot", "ties", "first_tie_t", "first_tie_slot", "max_margin_strata", "dloglik",
                                               "free_running_dloglik", "free_running_differing_ancestors_by_step_from_first_tie")} for r in tied],
            "seeds": list(acc_seeds), "N": N, "T": T, "m": m, "filters_in_flight": R, "kernel": acc_kernel, "launch_sequences": acc_seqs,
            "weights_layout_factored": acc_layout["factored"],
            "note": "identical Philox streams on both sides; they differ in libm vs device elementary functions (<= 2 ulp) and in "
                    "reduction shape.  A slot's stratified target lands within rounding distance of a cumulative weight about once per "
                    "1e9 particle-steps; two free-running systems then differ from that step on by Monte-Carlo noise of the estimator "
                    "(loglik_sd_of_the_estimator_across_streams; free_running_* fields) -- or not at all on a series whose weights "
                    "are all 1 - O(1e-9).  loglik_rmse_vs_cpu_same_seeds is against the oracle that follows the device through "
                    "those ties and is held to the fp64 bar on every pair; loglik_rmse_free_running_oracle leaves the oracle alone"}


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(n_gpus):
    """`python bench.py --gpus N` outside a launcher: start N ranks of this script, one per GPU, under torch.distributed.run
    (exactly the driver's command line) and pass rank 0's JSON line through.  OMP_NUM_THREADS: the launcher would set 1;
    the ranks of a node share its cores instead (the library's own host threads are sized per rank in post_correct_phase)."""
    import subprocess
    import torch
    backend = os.environ.get("BSSM_BENCH_BACKEND", "nccl")
    nd = torch.cuda.device_count()
    if backend == "nccl" and nd < n_gpus:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (one rank per GPU over RCCL; BSSM_BENCH_BACKEND=gloo lets "
                         "ranks share a GPU for control-flow tests only)" % (n_gpus, nd))
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n_gpus)))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["BSSM_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--particles", type=int, default=65536)
    ap.add_argument("--replicas", type=int, default=32,
                    help="independent filters (different Philox streams) in flight per step; 1 = latency regime")
    ap.add_argument("--T", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-regime-b", action="store_true")
    ap.add_argument("--no-single", action="store_true")
    ap.add_argument("--no-post-correct", action="store_true")
    ap.add_argument("--no-dropin", action="store_true")
    ap.add_argument("--no-accuracy", action="store_true")
    ap.add_argument("--accuracy-pairs", type=int, default=64, help="(seed, slot) pairs of the timed plan the oracle re-runs (64 = all of two passes)")
    ap.add_argument("--series", choices=["informative", "model"], default="informative",
                    help="the observed series of the metric model: counts that stay informative (default), or simulated from the "
                         "model itself (rounds 1-3: its integrated-random-walk level drifts off, 98 %% zeros, weights ~ 1)")
    ap.add_argument("--factored-weights-layout", action="store_true",
                    help="quote `value` on the library's factored weights layout (opt-in for device-resident plans) instead of the reference's "
                         "exp(w - max_t), which is what the drop-in entry points return and what `value` is quoted on since round 6")
    ap.add_argument("--reference-weights-layout", action="store_true", help="(the default since round 6; kept so that older command lines still run)")
    ap.add_argument("--no-mng", action="store_true")
    ap.add_argument("--no-live-pmc", action="store_true", help="take roofline.traffic from the committed profile instead of two rocprofv3 --pmc passes of this run")
    ap.add_argument("--no-side-runs", action="store_true", help="skip the reference-layout / other-series runs of the same batch")
    ap.add_argument("--no-clock", action="store_true", help="skip the rocm-smi clock / power samples under load")
    ap.add_argument("--pc-draws", type=int, default=2340, help="retained draws of the post_correct job (C5: 2340)")
    ap.add_argument("--no-live-valu", action="store_true", help="take the VALU instruction counts of the roofline_valu objects from the committed profiles instead of one rocprofv3 --pmc SQ_INSTS_VALU pass of this run")
    ap.add_argument("--pc-c5-only", action="store_true", help="post_correct leg: the C5 job alone (the counter pass of live_valu_counts)")
    ap.add_argument("--no-projection", action="store_true", help="skip post_correct_scaling.projected (one rank's share of the job at 1 / 2 / 4 / 8 ranks, measured on this GPU)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # not under a launcher: become one (N ranks, one per GPU); the ranks print the line
        raise SystemExit(launch_ranks(args.gpus))

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but the launcher started %d rank(s); reporting n_gpus = %d" % (args.gpus, world, world), file=sys.stderr)
    if world > 1:
        torch.set_num_threads(max(1, (os.cpu_count() or 8) // world))  # (torch.distributed.run exports OMP_NUM_THREADS=1)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the particle filters have no CPU path)")
    # BSSM_BENCH_BACKEND=gloo is a test hook: several ranks sharing one GPU (collectives on CPU tensors) to exercise the
    # multi-rank control flow on a single-GPU box; the driver's runs use nccl (= RCCL), one rank per GPU
    backend = os.environ.get("BSSM_BENCH_BACKEND", "nccl")
    local_dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_dev))
        else:
            dist.init_process_group(backend)
    cdev = "cuda" if backend == "nccl" else "cpu"  # where the tensors of a collective live
    comm = {"backend": "none", "ranks": 1, "distinct_gpus": 1}
    if dist is not None:
        # what the communicator itself says: one all-reduce of ones over it (= ranks that took part in a collective), and
        # the PCI bus ids of the ranks' devices (distinct_gpus < ranks means ranks share a GPU: the gloo test hook only)
        one = torch.ones(1, device=cdev, dtype=torch.float64)
        dist.all_reduce(one)
        props = torch.cuda.get_device_properties(local_dev)
        ident = "%s/%s/%s" % (getattr(props, "pci_bus_id", local_dev), getattr(props, "pci_device_id", 0), getattr(props, "uuid", local_dev))
        ids = [None] * world
        dist.all_gather_object(ids, ident)
        comm = {"backend": "rccl" if dist.get_backend() == "nccl" else dist.get_backend(), "ranks": int(round(float(one.item()))),
                "world_size": dist.get_world_size(), "distinct_gpus": len(set(ids))}

    from bssm_amd.plan import FilterPlan
    from models_zoo import negbin_m4, negbin_m4_informative

    global LIVE_VALU
    if rank == 0 and world == 1 and not args.no_live_valu and not args.no_live_pmc:
        # (before this process holds device memory: the counter pass is a child process running the same legs once)
        LIVE_VALU = live_valu_counts(args)

    N, T, m = args.particles, args.T, 4
    # The metric model on counts that stay informative (default since round 4), or on a series simulated
    # from the model itself -- rounds 1-3: the model's level is an integrated random walk, on seed 103 the signal sits at
    # -15 ... -38 after ~100 steps, 98 % of the counts are 0, every weight is 1 - O(1e-9) and resampling is the identity map.
    # Same matrices, same kernel, same bytes; what differs is what the ancestor search and the gathers see.
    make_model = negbin_m4_informative if args.series == "informative" else negbin_m4
    model = make_model(T, seed=103)
    R = args.replicas
    # Round 6: `value` is quoted on the weights layout the reference interface returns, exp(w - max_t) (what bssm_bsf hands back).  The
    # library's factored layout (opt-in for device-resident callers, include/bssm_amd.h BSSM_PLAN_FACTORED_WEIGHTS: block-scaled
    # weight x one factor per (time point, particle block); bssm_plan_fetch multiplies it out) is timed beside it as
    # roofline_factored_layout; --factored-weights-layout swaps the two.
    fw = args.factored_weights_layout and not args.reference_weights_layout
    plan = FilterPlan(model, N, method="bsf", keep_history=True, force="grid", n_rep=R, factored_weights=fw)
    info = plan.info()
    stream = torch.cuda.current_stream().cuda_stream

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(args.warmup):
        plan.run(1000 + w, stream0=rank * R, hip_stream=stream)
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        plan.run(1 + s, stream0=rank * R, hip_stream=stream)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ll = plan.loglik(stream)
    assert np.all(np.isfinite(ll)), ll
    layout = plan.weights_layout()

    psteps_per_pass = float(N) * T * R
    value = world * psteps_per_pass * args.steps / elapsed
    ms_per_step = elapsed / args.steps * 1e3

    out = {
        "metric": "particle-steps/sec (N*T), bootstrap filter, ssm_ung negbin T=1000 m=4 N=65536",
        "value": value, "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic", "rccl_ranks": comm["ranks"] if comm["backend"] == "rccl" else 0, "communicator": comm,
        "config": {"workload": "C2-m4: ssm_ung negative binomial, m=k=4, T=%d, N=%d, full-output bootstrap filter "
                               "(alpha, indices, weights kept in HBM); one step = %d independent filter calls "
                               "(different Philox streams) advanced together; --gpus N = the same batch on each GPU; "
                               "observed series: %s; weights layout of `value`: %s" % (T, N, R,
                               "informative counts (ESS ~0.76 N, every step resamples for real)" if args.series == "informative"
                               else "simulated from the model (98 % zeros, weights ~ 1, resampling ~ identity)",
                               "factored (library layout, opt-in for device-resident plans); the reference's exp(w - max_t) layout "
                               "is timed beside it as roofline_reference_layout" if fw else "reference: exp(w - max_t), what the drop-in entry points "
                               "return (the library's factored layout is timed beside it as roofline_factored_layout)"),
                   "series": args.series, "filters_in_flight": R, "weights_layout": ("factored: exp(w - max_block) written once + exp(max_block - max_t) per (t, block of %d)"
                                                               % layout["block_particles"]) if layout["factored"] else "reference: exp(w - max_t)",
                   "kernel_shape": info["path"], "blocks_per_launch": info["blocks"],
                   "threads": info["threads"], "launches_per_step": T + 3, "parallelism": "replicas only x%d" % world},
    }

    if rank == 0:
        # ---- roofline of the dominant kernel: HIP events around sampled pf_grid_step launches
        k_ms, k_n = plan.time_kernel(77, sample_every=16, hip_stream=stream)
        # a batch this size is stepped as `seqs` interleaved launch sequences over disjoint halves of its filters (two
        # streams, so that one half's step tail runs under the other's body): ONE launch covers R / seqs filters and
        # `seqs` launches are in flight at any time
        seqs = max(1, plan.launch_sequences())
        bytes_per_launch = float(R) / seqs * N * info["bytes_per_particle_step"]
        ppt = max(1, -(-N // (info["blocks"] // info["n_filters"] * info["threads"])))  # particles per thread of the step kernel
        achieved_events = seqs * bytes_per_launch / (k_ms * 1e-3) / 1e9
        # Round 6 (review of round 5, item 5): `achieved` / `frac` are the WALL-CLOCK figures of the timed region -- algorithmic bytes of
        # all its launches over its elapsed time, i.e. bytes per launch over the wall time one launch accounts for (wall_per_launch_us);
        # the HIP-event launch duration gives achieved_by_hip_events / frac_by_hip_events beside them (the two launch sequences overlap
        # by slightly less than 2.0, so the event figure is ~1 % higher)
        achieved = (value / world) * info["bytes_per_particle_step"] / 1e9
        committed = pmc_traffic(R // seqs, N, ppt, plan.last_kernel(), layout["factored"])
        out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS, "frac_source": "wall clock of the timed region",
                           "achieved_by_hip_events": achieved_events, "frac_by_hip_events": achieved_events / HBM_PEAK_GBS, "traffic": committed,
                           "traffic_source": "profiles/r05_pmc_traffic%s.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this script at this kernel)" % ("" if layout["factored"] else "_reference_layout"),
                           "traffic_committed_profile": committed,
                           "kernel": plan.last_kernel(), "avg_launch_us": k_ms * 1e3, "sampled_launches": k_n,
                           "concurrent_launches": seqs, "filters_per_launch": R // seqs,
                           "wall_per_launch_us": ms_per_step * 1e3 / (T + 1) / seqs,
                           "algorithmic_bytes_per_launch": bytes_per_launch,
                           "achieved_note": "achieved = algorithmic bytes of the timed region / its wall time; achieved_by_hip_events = "
                                            "concurrent_launches x algorithmic_bytes_per_launch / avg_launch_us (the launches of the two sequences "
                                            "overlap in time, each lasting avg_launch_us)",
                           "frac_of_measured_copy_6290GBs": achieved / 6290.0}
        if world == 1 and not args.no_clock:
            ck = clock_power_while(lambda k: (plan.run(5000 + k, stream0=0, hip_stream=stream), torch.cuda.synchronize()))
            torch.cuda.synchronize()
            out["roofline"]["clock_power_under_load"] = ck
            out["roofline"]["valu"] = valu_roofline(HEAD_VALU_PER_PSTEP, seqs * (R // seqs) * float(N) / (k_ms * 1e-3),
                                                    "pf_grid_step_v2<4,4,4,false,true,true>", ck, key="head") if ppt == 4 and m == 4 and (layout["factored"] or (LIVE_VALU or {}).get("head")) else None
    head_kernel, head_factored = (FilterPlan.last_kernel(), layout["factored"]) if rank == 0 else ("", True)
    plan.close()
    if rank == 0 and world == 1 and not args.no_live_pmc and "roofline" in out:
        from bssm_amd import _lib as _Lp
        _Lp.lib().bssm_amd_trim_cache()   # the child processes need the memory this process has cached
        live = live_pmc_traffic(args, head_kernel, head_factored)
        if live:
            out["roofline"]["traffic"] = live["hbm_bytes_per_launch"]
            out["roofline"]["traffic_source"] = ("measured in this run: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of one timed step of the "
                                                 "same batch, FETCH factor calibrated in-process on calib_copy_kernel")
            out["roofline"]["traffic_live"] = live
            out["roofline"]["traffic_over_algorithmic"] = live["hbm_bytes_per_launch"] / out["roofline"]["algorithmic_bytes_per_launch"]

    def side_run(mdl, factored):
        """the same batch on another (series, weights layout): a short timed run + the kernel's launch duration"""
        pl = FilterPlan(mdl, N, method="bsf", keep_history=True, force="grid", n_rep=R, factored_weights=factored)
        pl.run(999, stream0=0, hip_stream=stream); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for q in range(5):
            pl.run(50 + q, stream0=0, hip_stream=stream)
        torch.cuda.synchronize()
        e = (time.perf_counter() - t1) / 5
        kms, _ = pl.time_kernel(77, sample_every=16, hip_stream=stream)
        sq = max(1, pl.launch_sequences())
        ach = sq * (float(R) / sq * N * pl.info()["bytes_per_particle_step"]) / (kms * 1e-3) / 1e9
        res = {"value": float(N) * T * R / e, "unit": "particle-steps/s", "ms_per_step": e * 1e3, "steps": 5, "kernel": FilterPlan.last_kernel(),
               "avg_launch_us": kms * 1e3, "achieved": ach, "frac": ach / HBM_PEAK_GBS, "frac_by_wall_clock": float(N) * T * R / e * 88.0 / 8e12,
               "weights_layout_factored": pl.weights_layout()["factored"]}
        pl.close()
        return res

    if rank == 0 and world == 1 and not args.no_side_runs:
        if fw:   # the layout the drop-in entry points return, next to the factored one `value` is quoted on
            out["roofline_reference_layout"] = side_run(model, False)
        else:    # the library's factored layout (opt-in for device-resident plans), next to the reference one `value` is quoted on
            out["roofline_factored_layout"] = side_run(model, True)
        other = negbin_m4 if args.series == "informative" else negbin_m4_informative
        out["other_series"] = {"series": "model" if args.series == "informative" else "informative", **side_run(other(T, seed=103), fw),
                               "note": "the same model matrices and plan on the other observed series (see config.workload); rounds 1-3 "
                                       "quoted `value` on the model-simulated series"}
        # the same batch under the CONTRACT's choices, in `config` so that the record is self-contained (round-4 review 6a): the
        # reference's weights layout (what the drop-in entry points return) and the SURVEY 8d series simulated from the model
        if "roofline_reference_layout" in out:
            out["config"]["reference_layout_frac"] = out["roofline_reference_layout"]["frac_by_wall_clock"]
        if "roofline_factored_layout" in out:
            out["config"]["factored_layout_frac"] = out["roofline_factored_layout"]["frac_by_wall_clock"]
        out["config"]["contract_series_frac" if args.series == "informative" else "informative_series_frac"] = out["other_series"]["frac_by_wall_clock"]
        out["config"]["fracs_note"] = ("wall-clock roofline fraction of the same 32-filter batch (5 timed passes each): factored_layout_frac = the library's factored weights "
                                       "layout (opt-in, device-resident plans), contract_series_frac = y simulated from the model (SURVEY.md 8d seed 103: 98 % zeros); "
                                       "88 B per particle-step; the HIP-event figures are in the side objects")

    if rank == 0 and world == 1 and R != 1 and not args.no_single:
        # ---- regime A: ONE filter, latency-bound (1000 dependent launches, one wave per SIMD)
        p1 = FilterPlan(model, N, method="bsf", keep_history=True, force="grid", n_rep=1)
        p1.run(3, hip_stream=stream); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(5):
            p1.run(4 + s, hip_stream=stream)
        torch.cuda.synchronize()
        e1 = (time.perf_counter() - t0) / 5
        k1_ms, _ = p1.time_kernel(78, sample_every=16, hip_stream=stream)
        out["single_filter"] = {"workload": "one bootstrap_filter call, same model, N=%d, T=%d, full output" % (N, T),
                                "value": float(N) * T / e1, "unit": "particle-steps/s", "ms_per_call": e1 * 1e3,
                                "avg_launch_us": k1_ms * 1e3,
                                "algorithmic_GBs": N * info["bytes_per_particle_step"] / (k1_ms * 1e-3) / 1e9}
        p1.close()

    if rank == 0 and world == 1 and not args.no_regime_b:
        # ---- regime B: many independent filters, one workgroup each (post-correction shape)
        nb, fb = 2048, 512
        pb = FilterPlan(model, nb, method="bsf", n_rep=fb, keep_history=False, force="wg")
        pb.run(5, hip_stream=stream); torch.cuda.synchronize()
        reps = 3
        t0 = time.perf_counter()
        for s in range(reps):
            pb.run(6 + s, hip_stream=stream)
        torch.cuda.synchronize()
        eb = (time.perf_counter() - t0) / reps
        kb_ms, _ = pb.time_kernel(9, hip_stream=stream)
        ib = pb.info()
        ps = float(nb) * fb * T
        out["regime_b"] = {"workload": "%d independent bootstrap filters, N=%d, T=%d, m=4, log-likelihood only "
                                       "(one workgroup per filter, state in LDS)" % (fb, nb, T),
                           "value": ps / eb, "unit": "particle-steps/s", "ms_per_pass": eb * 1e3,
                           "kernel": pb.last_kernel(), "kernel_ms": kb_ms,
                           "algorithmic_GBs": ps * ib["bytes_per_particle_step"] / (kb_ms * 1e-3) / 1e9,
                           "note": "state never leaves LDS: HBM traffic ~0, bound by fp64 VALU/transcendentals",
                           "roofline_valu": valu_roofline(REGB_VALU_PER_PSTEP, ps / (kb_ms * 1e-3), pb.last_kernel(), key="regb")}
        pb.close()

    if rank == 0 and world == 1 and not args.no_regime_b and not args.no_mng:
        # ---- BASELINE configs[3]: ssm_mng m = 8, p = 4, T = 2000, N = 16384 -- the fp64-MFMA transition path
        # (pf_grid_step_v2<8,8,2,true,...>), full output; 16 m + 24 = 152 algorithmic bytes per particle-step.  64 filters in
        # flight: a 16 384-particle filter is 32 workgroups at two particles per lane, and the step kernels of this shape reach
        # their steady rate from ~48 filters on (profiles/r03_c4_fill.txt: 32 -> 0.42, 48 -> 0.51, 64 -> 0.52, 80 -> 0.52 of
        # 8 TB/s); 160 GB of outputs stay resident, so the cached blocks of the earlier legs go back to the driver first
        from bssm_amd import _lib as _L
        from models_zoo import mng_c4
        _L.lib().bssm_amd_trim_cache()
        nm, fm, Tm = 16384, 64, 2000
        pm = FilterPlan(mng_c4(Tm), nm, method="bsf", keep_history=True, force="grid", n_rep=fm, factored_weights=fw)
        pm.run(5, hip_stream=stream); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(2):
            pm.run(6 + s, hip_stream=stream)
        torch.cuda.synchronize()
        em = (time.perf_counter() - t0) / 2
        psm = float(nm) * fm * Tm
        out["mng_c4"] = {"workload": "ssm_mng m=8 k=8 p=4 (poisson, poisson, negbin, gaussian), T=%d, N=%d, bootstrap filter, full output, "
                                     "%d filters in flight" % (Tm, nm, fm),
                         "value": psm / em, "unit": "particle-steps/s", "ms_per_pass": em * 1e3, "kernel": pm.last_kernel(),
                         "weights_layout_factored": pm.weights_layout()["factored"], "launch_sequences": pm.launch_sequences(),
                         "algorithmic_GBs": psm * 152 / em / 1e9, "frac_of_8TBs": psm * 152 / em / 8e12,
                         "roofline_valu": valu_roofline(C4_VALU_PER_PSTEP, psm / em, "pf_grid_step_v2<8,8,2,true,true,true>", key="c4")}
        ctm = committed_traffic(pm.last_kernel())
        if ctm:   # (round 6: the C4 kernel's HBM bytes by counters, a committed profile of this kernel at this launch shape when the batch is the default one)
            out["mng_c4"]["roofline"] = {"bound": "hbm", "achieved": psm * 152 / em / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": psm * 152 / em / 8e12, **ctm}
        if not args.no_clock:
            ckm = clock_power_while(lambda k: (pm.run(5000 + k, hip_stream=stream), torch.cuda.synchronize()), samples=2)
            torch.cuda.synchronize()
            out["mng_c4"]["clock_power_under_load"] = ckm
            out["mng_c4"]["roofline_valu"] = valu_roofline(C4_VALU_PER_PSTEP, psm / em, "pf_grid_step_v2<8,8,2,true,true,true>", ckm, key="c4")
        pm.close()
        _L.lib().bssm_amd_trim_cache()

    if rank == 0 and world == 1 and not args.no_regime_b and not args.no_mng:
        # ---- round 5: linear-state models of 17 ... 64 state dimensions (bsm_ng with a seasonal component of period 24 / 52:
        # 25 / 53 states, R/models.R:995-997) on the fp64 matrix cores in fragment-native layout (bssm_amd/csrc/pf_big.h);
        # roofline = the matrix peak by the flops the v_mfma_f64_16x16x4_f64 sequence executes on the padded tiles
        try:
            from bssm_amd import models as Mb
            big = {}
            for period in (24, 52):
                Tb, mb = 200, 2 + period - 1
                rngb = np.random.default_rng(1)
                yb = rngb.poisson(np.exp(1.5 + 0.4 * np.sin(2 * np.pi * np.arange(Tb) / period))).astype(float)
                mdl = Mb.bsm_ng(yb, sd_level=0.1, sd_slope=0.01, sd_seasonal=0.05, period=period, distribution="poisson",
                                a1=np.r_[1.5, 0.0, np.zeros(period - 1)], P1=np.diag(np.r_[1.0, 0.01, np.full(period - 1, 0.05)]))
                MPb = 32 if mb <= 32 else 64
                Nb, Fb = 1024, 256
                pbg = FilterPlan(mdl, Nb, method="bsf", n_rep=Fb, keep_history=False)
                pbg.run(5, hip_stream=stream); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for s in range(3):
                    pbg.run(6 + s, hip_stream=stream)
                torch.cuda.synchronize()
                eb = (time.perf_counter() - t0) / 3
                flops = 2.0 * MPb * ((mb + 3) // 4 * 4 + 4)   # [T R] tile: MP rows x (m, k = 3 rounded up to fours) columns
                psb = float(Nb) * Fb * Tb / eb
                big["period_%d" % period] = {"workload": "bsm_ng(period = %d): m = %d states, k = 3, T = %d, bootstrap filter, N = %d x %d filters, log-likelihood only"
                                                         % (period, mb, Tb, Nb, Fb),
                                             "value": psb, "unit": "particle-steps/s", "ms_per_pass": eb * 1e3, "kernel": pbg.last_kernel(),
                                             "roofline": {"bound": "mfma", "achieved": psb * flops / 1e12, "peak": 78.6, "unit": "TFLOP/s",
                                                          "frac": psb * flops / 78.6e12, "flops_per_particle_step_on_the_tiles": flops}}
                if "row" in big["period_%d" % period]["kernel"]:
                    # the row shape keeps the particle system in HBM: SURVEY 8(d)'s log-likelihood-only figure, 16 m + 16 bytes per particle-step
                    # (ancestor gather + new state, weights exchange) -- at these dimensions that roof is the nearer one
                    hb = psb * (16.0 * mb + 16.0)
                    big["period_%d" % period]["roofline_hbm"] = {"bound": "hbm", "achieved": hb / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": hb / 8e12,
                                                                 "bytes_per_particle_step": 16.0 * mb + 16.0, **(committed_traffic(pbg.last_kernel()) or {"traffic": None})}
                pbg.close()
                try:
                    big["period_%d" % period]["post_correct"] = seasonal_post_correct_leg(period)
                except Exception as e:  # noqa: BLE001
                    big["period_%d" % period]["post_correct"] = {"error": repr(e)[:200]}
            try:
                big["mng_m20_post_correct"] = mng_big_post_correct_leg()
            except Exception as e:  # noqa: BLE001
                big["mng_m20_post_correct"] = {"error": repr(e)[:200]}
            out["big_states"] = big
        except Exception as e:  # noqa: BLE001
            out["big_states"] = {"error": repr(e)[:200]}
        from bssm_amd import _lib as _Lb
        _Lb.lib().bssm_amd_trim_cache()   # (the set-up of 256 draws of 53 states leaves 3.5 GB of work arrays cached: the legs below start from a clean arena)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # N = 1 only: the other ranks would idle behind it
        # ---- CPU baseline: the oracle (reference loop structure, one thread per filter) on a bounded sample
        from bssm_amd import models as M
        from oracle import oracle as O
        Ts = T  # the whole single-filter workload: ~10 s of one core
        sample = make_model(Ts, seed=103)
        arr = M.model_arrays(sample)
        O.bsf_filter(arr, 1024, seed=1)  # warm
        t0 = time.perf_counter()
        O.bsf_filter(arr, N, seed=1)
        ec = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": float(N) * Ts / ec, "unit": "particle-steps/s", "cores": 1, "kind": "port",
                               "sample": "one full filter call of the benchmark model: N=%d, T=%d, 1 thread (the reference has no "
                                         "intra-filter parallelism), gcc -O2, %.1f s" % (N, Ts, ec)}

    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.no_accuracy:
        out["accuracy"] = accuracy_leg(model, N, T, R, m, fw, stream, args.accuracy_pairs)

    if rank == 0 and not args.no_dropin:
        # ---- the drop-in boundary as R sees it: host buffers in, host buffers out (PCIe-inclusive; never `value`).
        # Reference contract: src/R_bootstrap_filter.cpp:136-155 (outputs are ordinary R arrays).
        try:
            from bssm_amd import filters as Fd, models as Md

            first = {}

            def best_of(fn, reps=3, tag=None):
                # the call alone: releasing the previous result (2.6 GB of pageable host memory, ~90 ms of munmap) is the
                # caller's garbage collection, not the entry point, and stays outside the timed region
                t1 = time.perf_counter(); res = fn(); first[tag] = time.perf_counter() - t1
                bt = 1e9
                for _ in range(reps):
                    del res
                    t1 = time.perf_counter(); res = fn(); bt = min(bt, time.perf_counter() - t1)
                return bt

            # the floor of such a call: its outputs over PCIe at the rate THIS box copies device memory into pinned host memory
            gib = 1 << 30
            dsrc = torch.empty(gib, dtype=torch.uint8, device=cdev)
            hdst = torch.empty(gib, dtype=torch.uint8, pin_memory=True)
            hdst.copy_(dsrc, non_blocking=True); torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                hdst.copy_(dsrc, non_blocking=True)
            torch.cuda.synchronize()
            d2h_gbs = 3 * gib / (time.perf_counter() - t1) / 1e9
            del dsrc, hdst
            e1 = best_of(lambda: Fd.bootstrap_filter(model, N, seed=1), tag="c2m4")
            yx = np.loadtxt(os.path.join(ROOT, "tests", "golden", "exchange.txt"))
            svm = Md.svm(yx, rho=Md.uniform(0.98, -0.999, 0.999), sd_ar=Md.halfnormal(0.2, 5), sigma=Md.halfnormal(1.0, 2))
            e3 = best_of(lambda: Fd.particle_smoother(svm, 32768, method="psi", seed=1))
            out["dropin"] = {
                "note": "one call through the C ABI with host (pageable numpy) buffers: model upload, filter, and every output "
                        "(particle cube m x (n+1) x N, weights, summaries) copied back; output slabs stream out while the filter steps",
                "pinned_d2h_GBs": d2h_gbs,
                "C2-m4_bootstrap_filter": {"ms_per_call": e1 * 1e3, "value": N * float(T) / e1, "unit": "particle-steps/s",
                                           "output_GB": (m * (T + 1) * N + (T + 1) * N) * 8 / 1e9,
                                           "pcie_floor_ms": (m * (T + 1) * N + (T + 1) * N) * 8 / 1e9 / d2h_gbs * 1e3,
                                           "first_call_ms": first["c2m4"] * 1e3,
                                           "floor_note": "pcie_floor_ms = output bytes / the pinned device-to-host rate measured in this run (1 GiB copies); "
                                                         "the entry point writes into the caller's PAGEABLE arrays, as R's are.  first_call_ms = the first call of "
                                                         "this entry point in this process (device arena already warm from the legs above; pinned staging and the "
                                                         "2.6 GB of fresh pageable output pages are not)"},
                "C3_psi_particle_smoother": {"ms_per_call": e3 * 1e3, "value": 32768 * float(len(yx)) / e3, "unit": "particle-steps/s",
                                             "workload": "svm on the exchange series (n = 945), psi-APF N = 32768, back-trace + summary"}}
        except Exception as e:  # noqa: BLE001
            out["dropin"] = {"error": repr(e)[:200]}

    if not args.no_post_correct:
        # ---- the sharded half of the path (all ranks): IS post-correction, SURVEY.md 8e
        try:
            pc = post_correct_phase(rank, world, dist, cdev, cpu_baseline=not args.no_cpu_baseline, ns=args.pc_draws)
        except Exception as e:  # noqa: BLE001
            pc = {"error": repr(e)[:200]}
        if rank == 0:
            out["post_correct"] = pc
        # the same job eight times larger (what a 160 000-iteration chain retains): long enough for a scaling curve
        try:
            pc8 = {"skipped": "--pc-c5-only"} if args.pc_c5_only else post_correct_phase(rank, world, dist, cdev, reps=2, ns=8 * args.pc_draws, label="C5 x8 draws")
        except Exception as e:  # noqa: BLE001
            pc8 = {"error": repr(e)[:200]}
        if rank == 0:
            out["post_correct_large"] = pc8
            # the strong-scaling evidence of the path that shards (north_star: "near-linear post_correct scaling to 8 GPUs"),
            # at the top level: the whole job's wall time at this N against the one-GPU time of the SAME job (measured in this
            # run when N = 1; otherwise the figure stored by the round's one-GPU run, profiles/r05_post_correct_n1.json)
            try:
                n1 = json.load(open(os.path.join(ROOT, "profiles", "r04_post_correct_n1.json")))
            except (OSError, ValueError):
                n1 = {}
            sc = {"scaling": "strong", "n_gpus": world, "rccl_ranks": comm["ranks"] if comm["backend"] == "rccl" else 0,
                  "distinct_gpus": comm["distinct_gpus"], "n1_source": "this run" if world == 1 else "profiles/r05_post_correct_n1.json (one MI355X, same job)"}
            for key, obj, name in (("C5", pc, "post_correct"), ("C5_x8_draws", pc8, "post_correct_large")):
                if isinstance(obj, dict) and "seconds" in obj:
                    base = obj["seconds"] if world == 1 else n1.get(name, {}).get("seconds")
                    sc[key] = {"seconds": obj["seconds"], "seconds_n1": base, "speedup_vs_n1": (base / obj["seconds"]) if base else None,
                               "particle_steps_per_s": obj["value"]}
            if world == 1 and not args.no_projection and not args.pc_c5_only:
                # PROJECTED strong scaling, measured on this one GPU (no multi-GPU node is available to the builder; the driver's
                # SCALE run is the real curve): what ONE rank of an N-rank job does -- ceil(draws / N) draws, streams keyed by the
                # global draw index -- timed as a whole call of the library, BEFORE the collective (one all-gather of <= 150 KB).
                # A rank's share of a job this size is 1-10 rounds of 256 workgroup-resident filters; a sparsely filled last
                # round goes through the row family (pf_row_step.h), bit-identical, which is what keeps the 8-rank share near 1 / 8.
                proj = {"what": "projected, one rank's share of the job on one MI355X, before the collective", "ranks": [1, 2, 4, 8]}
                for key, total in (("C5", args.pc_draws), ("C5_x8_draws", 8 * args.pc_draws)):
                    ms = []
                    for wr in proj["ranks"]:
                        r = post_correct_phase(0, 1, None, cdev, reps=2 if total > 5000 else 3, ns=-(-total // wr), label="%s / %d ranks" % (key, wr))
                        ms.append(r["seconds"] * 1e3)
                    proj[key] = {"draws_per_rank": [-(-total // wr) for wr in proj["ranks"]], "shard_ms": [round(v, 3) for v in ms],
                                 "speedup": [round(ms[0] / v, 3) for v in ms]}
                sc["projected"] = proj
            out["post_correct_scaling"] = sc
        if rank == 0 and world == 1:
            try:
                out["post_correct_single_process"] = single_process_multi_device(8 * args.pc_draws)
            except Exception as e:  # noqa: BLE001
                out["post_correct_single_process"] = {"error": repr(e)[:200]}

    if rank == 0 and world == 1 and not args.no_post_correct and not args.no_mng:
        # ---- IS post-correction of ssm_mng draws of the C4 shape (m = 8, p = 4): the psi set-up (Laplace step, Kalman filter +
        # covariance smoother over n p virtual steps, conditional factors, tables) runs on the device for every draw -- the
        # smoother row-distributed over eight lanes per draw -- in front of small filters; SURVEY.md 8f rank 1
        try:
            out["post_correct_mng"] = mng_post_correct_leg()
        except Exception as e:  # noqa: BLE001
            out["post_correct_mng"] = {"error": repr(e)[:200]}

    if rank == 0 and os.environ.get("BSSM_BENCH_CALIB"):
        # profiling runs only (scripts/profile_round.sh): one kernel of known traffic (2 GiB read + 2 GiB written) so that
        # the PMC FETCH_SIZE / WRITE_SIZE factors can be calibrated inside the same process
        import ctypes
        from bssm_amd import _lib
        _lib.check(_lib.lib().bssm_debug_stream_copy(ctypes.c_uint64(1 << 28)))

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and world > 1 and not args.no_post_correct:
        # the same job sharded INSIDE the library by one process (what an R caller gets) over the GPUs the ranks used; the
        # other ranks have left their GPUs by now (logical shards on one GPU when the ranks shared it: gloo test hook)
        try:
            from bssm_amd import _lib
            _lib.lib().bssm_amd_trim_cache()
            devs = list(range(world)) if comm["distinct_gpus"] == world else [local_dev] * world
            out["post_correct_single_process"] = single_process_multi_device(8 * args.pc_draws, devs)
        except Exception as e:  # noqa: BLE001
            out["post_correct_single_process"] = {"error": repr(e)[:200]}
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
