"""C5 post-correction (2 340 draws, psi-APF N = 4096, T = 300, output theta) -- the timing bench.py's post_correct object
reports, alone: python scripts/c5_quick.py [draws] [reps]"""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import bench
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 2340
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
r = bench.post_correct_phase(0, 1, None, reps=reps, ns=ns)
print(json.dumps({k: r[k] for k in ("seconds", "value", "per_rank_ms", "checksum_logw")}), "valu_frac", r["roofline_valu"]["frac"])
