"""Throughput of the row family (pf_row_step.h) against the workgroup-resident kernel on the C5 post-correction shape (psi-APF,
N = 4096, T = 300, m = 2) as a function of the number of draws: BSSM_AMD_FORCE_PATH=row sends every draw of a call through the
row family, =wg through pf_wg_kernel -> profiles/r05_row_family_probe.txt"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import bench
for ns in [int(a) for a in sys.argv[1:]] or [37, 64, 128, 256, 293, 512, 549, 1024]:
    row = []
    for path in ("wg", "row"):
        os.environ["BSSM_AMD_FORCE_PATH"] = path
        r = bench.post_correct_phase(0, 1, None, reps=3, ns=ns)
        row.append("%s %.2f ms" % (path, r["seconds"] * 1e3))
    os.environ.pop("BSSM_AMD_FORCE_PATH")
    print("%5d draws | " % ns + " | ".join(row), flush=True)
