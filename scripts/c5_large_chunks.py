"""C5 x 8 draws (18 720) by chunk size of the post-correction pipeline (BSSM_AMD_IS_BIG_CHUNK): python scripts/c5_large_chunks.py"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import bench
for rnd in range(2):
    for ch in ("512", "1024", "2048", "4096", "9360"):
        os.environ["BSSM_AMD_IS_BIG_CHUNK"] = ch
        r = bench.post_correct_phase(0, 1, None, reps=2, ns=18720)
        print("chunk %5s: %.1f ms  %s" % (ch, r["seconds"] * 1e3, r["per_rank_ms"][0]), flush=True)
