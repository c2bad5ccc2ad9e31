"""C4 (ssm_mng m = 8, p = 4, N = 16 384, T = 2000, full output) by filters in flight x particles per lane."""
import sys, os, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
    from bssm_amd.plan import FilterPlan
    from models_zoo import mng_c4
    mdl = mng_c4(2000)
    for R in (8, 16, 24, 32, 48):
        p = FilterPlan(mdl, 16384, method="bsf", force="grid", keep_history=True, n_rep=R)
        p.run(1); p.loglik()
        ms = min(p.time(2, 3) for _ in range(2))
        print(f"R={R}: {R*16384*2000/ms*1e3:.3e} {FilterPlan.last_kernel()}", flush=True)
        p.close()
else:
    for ppt in ("", "1", "2"):
        env = dict(os.environ)
        if ppt: env["BSSM_AMD_GRID_PPT"] = ppt
        print("== BSSM_AMD_GRID_PPT=" + (ppt or "default"), flush=True)
        subprocess.run([sys.executable, __file__, "child"], env=env)
