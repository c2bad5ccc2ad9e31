"""Launch sequences (BSSM_AMD_GRID_STREAMS = 2 / 3 / 4) for the bench batch and C4: python scripts/streams_ab.py"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from bssm_amd.plan import FilterPlan
    from models_zoo import mng_c4, negbin_m4
    which, R = sys.argv[2], int(sys.argv[3])
    if which == "c4":
        p = FilterPlan(mng_c4(2000), 16384, method="bsf", force="grid", keep_history=True, n_rep=R, factored_weights=True); ps = R * 16384 * 2000
    else:
        p = FilterPlan(negbin_m4(1000, seed=103), 65536, method="bsf", force="grid", keep_history=True, n_rep=R, factored_weights=True); ps = R * 65536 * 1000
    p.run(1); p.loglik()
    ms = min(p.time(9, 3) for _ in range(2))
    print("%s R=%d streams=%s: %.3e  seqs=%d" % (which, R, os.environ.get("BSSM_AMD_GRID_STREAMS", "-"), ps / ms * 1e3, p.launch_sequences()), flush=True)
else:
    for rnd in range(2):
        for which, R in (("c2", 32), ("c2", 36), ("c4", 64), ("c4", 66)):
            for st in ("2", "3", "4"):
                subprocess.run([sys.executable, __file__, "child", which, str(R)], env=dict(os.environ, BSSM_AMD_GRID_STREAMS=st))
