"""C5 post-correction by shard size, round 4: what one rank of an N-GPU run does (2340 / N draws) on one GPU, with every draw
workgroup-resident (the default: results independent of the shard layout) and with the opt-in tail split (BSSM_AMD_TAIL_SPLIT=1)
-> profiles/r04_c5_shard_probe.txt"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import bench
base = {}
for world in (1, 2, 4, 8):
    ns = -(-2340 // world)
    row = []
    for rule in ("default", "tail_split"):
        if rule == "tail_split": os.environ["BSSM_AMD_TAIL_SPLIT"] = "1"
        else: os.environ.pop("BSSM_AMD_TAIL_SPLIT", None)
        r = bench.post_correct_phase(0, 1, None, reps=4, ns=ns)
        base.setdefault(rule, r["seconds"] * 1e3)
        row.append("%s %.2f ms (x%.2f of one GPU)" % (rule, r["seconds"] * 1e3, base[rule] / (r["seconds"] * 1e3)))
    os.environ.pop("BSSM_AMD_TAIL_SPLIT", None)
    print("ranks %d: %4d draws per rank | " % (world, ns) + " | ".join(row), flush=True)
