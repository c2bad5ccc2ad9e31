"""C5 post-correction by shard size: what one rank of an N-GPU run does (2340 / N draws), one GPU each -- the strong-scaling
curve of the job as far as a one-GPU box can show it; `old` = the chunking rule of round 2 (at least four chunks per call):
python scripts/c5_shard_probe.py"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import bench
base = None
for world in (1, 2, 4, 8):
    ns = -(-2340 // world)
    row = []
    for rule in ("new", "old"):
        if rule == "old" and ns < 2048: os.environ["BSSM_AMD_IS_CHUNK"] = str(max(64, ((ns + 3) // 4 + 63) // 64 * 64))
        else: os.environ.pop("BSSM_AMD_IS_CHUNK", None)
        r = bench.post_correct_phase(0, 1, None, reps=4, ns=ns)
        row.append((rule, r["seconds"] * 1e3))
    os.environ.pop("BSSM_AMD_IS_CHUNK", None)
    base = base or row[0][1]
    print("ranks %d: %4d draws per rank | " % (world, ns) + " | ".join("%s %.2f ms (x%.2f)" % (k, v, base / v) for k, v in row), flush=True)
