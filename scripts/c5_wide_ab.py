"""C5 post-correction (2 340 draws, psi-APF N = 4096) by workgroup width of the LDS-resident filter: BSSM_AMD_WG_WIDE=0|1"""
import os, sys, time, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
    import numpy as np
    import bench
    from bssm_amd import filters as F, post_correct as PC
    ns = 2340
    model, theta, counts, n = bench._c5_job(ns)
    mode = F.psi_setup(model)["mode"].reshape(-1, order="F")
    modes = np.repeat(mode[:, None], ns, axis=1)
    best, tm = 1e9, None
    for it in range(4):
        t0 = time.perf_counter()
        loc = PC._run_local_block_nlg(model, theta, modes, counts, 0, ns, 4096, 1, 2, 3, 32)
        el = time.perf_counter() - t0
        if it and el < best: best, tm = el, loc["timing_ms"]
    print("WIDE=%s: %.4f s  device %.1f ms  checksum %.10f" % (os.environ.get("BSSM_AMD_WG_WIDE", "default"), best, tm["device"], float(np.sum(np.log(loc["weights"])))), flush=True)
else:
    for rnd in range(2):
        for w in ("1", "0"):
            env = dict(os.environ); env["BSSM_AMD_WG_WIDE"] = w
            subprocess.run([sys.executable, __file__, "child"], env=env)
