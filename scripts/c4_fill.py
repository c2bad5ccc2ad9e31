"""C4 (ssm_mng m = 8, p = 4, N = 16 384, T = 2000, full output): filters in flight x particles per lane x launch sequences."""
import sys, os, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
    from bssm_amd.plan import FilterPlan
    from models_zoo import mng_c4
    mdl = mng_c4(2000)
    for R in [int(x) for x in sys.argv[2].split(",")]:
        p = FilterPlan(mdl, 16384, method="bsf", force="grid", keep_history=True, n_rep=R)
        p.run(1); p.loglik()
        ms = min(p.time(2, 2) for _ in range(2))
        v = R * 16384 * 2000 / ms * 1e3
        print(f"R={R}: {v:.3e} p-steps/s  {v*152/8e12:.3f} of 8 TB/s  {FilterPlan.last_kernel()} seqs={p.launch_sequences()}", flush=True)
        p.close()
else:
    rs = sys.argv[1] if len(sys.argv) > 1 else "32,48,64,80"
    for ppt, streams in (("", ""), ("1", ""), ("2", "1"), ("4", "")):
        env = dict(os.environ)
        if ppt: env["BSSM_AMD_GRID_PPT"] = ppt
        if streams: env["BSSM_AMD_GRID_STREAMS"] = streams
        print("== BSSM_AMD_GRID_PPT=%s BSSM_AMD_GRID_STREAMS=%s" % (ppt or "default", streams or "default"), flush=True)
        subprocess.run([sys.executable, __file__, "child", rs], env=env)
