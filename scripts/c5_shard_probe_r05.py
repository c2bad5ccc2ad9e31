"""C5 post-correction by shard size, round 5: what ONE rank of an N-GPU run does (ceil(draws / N) draws) on one GPU, before the
collective -- projected strong scaling of the job.  default = the last round of a shard goes through the row family when it is
sparsely filled (bit-identical to the workgroup-resident kernel: pf_canon.h) and, since late round 5, steps BESIDE the last full round;
behind = BSSM_AMD_TAIL_CONCURRENT=0 (the tail behind that round: the first half of round 5); no_split = BSSM_AMD_TAIL_SPLIT=0 (every draw
workgroup-resident, the round-4 default).  Two jobs: C5 as sized (2 340 draws) and eight times the draws (18 720).
-> profiles/r05_c5_shard_probe.txt"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import bench
for total, label in ((2340, "C5"), (18720, "C5x8")):
    base = {}
    for world in (1, 2, 4, 8):
        ns = -(-total // world)
        row = []
        for rule in ("default", "behind", "no_split"):
            os.environ.pop("BSSM_AMD_TAIL_SPLIT", None); os.environ.pop("BSSM_AMD_TAIL_CONCURRENT", None)
            if rule == "no_split": os.environ["BSSM_AMD_TAIL_SPLIT"] = "0"
            if rule == "behind": os.environ["BSSM_AMD_TAIL_CONCURRENT"] = "0"
            r = bench.post_correct_phase(0, 1, None, reps=4 if total < 5000 else 2, ns=ns)
            base.setdefault(rule, r["seconds"] * 1e3)
            row.append("%s %.2f ms (x%.2f of one GPU)" % (rule, r["seconds"] * 1e3, base[rule] / (r["seconds"] * 1e3)))
        os.environ.pop("BSSM_AMD_TAIL_SPLIT", None); os.environ.pop("BSSM_AMD_TAIL_CONCURRENT", None)
        print("%s ranks %d: %5d draws per rank | " % (label, world, ns) + " | ".join(row), flush=True)
