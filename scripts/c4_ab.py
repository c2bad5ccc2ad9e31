"""A/B probe for the C4 step kernel (scripts/ab_script.sh "scripts/c4_ab.py"): 64 filters in flight at 2 and 4 particles per lane"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from bssm_amd.plan import FilterPlan
    from models_zoo import mng_c4
    p = FilterPlan(mng_c4(2000), 16384, method="bsf", force="grid", keep_history=True, n_rep=int(sys.argv[2]))
    p.run(1); p.loglik()
    ms = min(p.time(9, 2) for _ in range(2))
    print("ppt=%s R=%s %.3e %s" % (os.environ.get("BSSM_AMD_GRID_PPT", "-"), sys.argv[2], int(sys.argv[2]) * 16384 * 2000 / ms * 1e3, FilterPlan.last_kernel()), flush=True)
else:
    for ppt, R in (("2", "64"), ("4", "64"), ("4", "80")):
        subprocess.run([sys.executable, __file__, "child", R], env=dict(os.environ, BSSM_AMD_GRID_PPT=ppt))
