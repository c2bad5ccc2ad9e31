"""Derive HBM bytes per pf_grid_step launch from the two PMC passes of scripts/profile_round.sh.
Usage: python scripts/pmc_traffic.py gpurun_out/prof REPLICAS PARTICLES T M > profiles/rNN_pmc_traffic.json
FETCH_SIZE / WRITE_SIZE are reported in KB per dispatch; on gfx950 FETCH_SIZE counts 64 B per 128-B request
(MI355X_MICROARCH.md, HBM section), so the read factor is calibrated on weights_finalize_kernel, whose traffic is
known exactly (bench.py launches it once when BSSM_BENCH_CALIB is set): calib_copy_kernel reads 2 GiB and writes 2 GiB."""
import ast
import json
import sys

d, R, N, T, m = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])


def read(path):
    out = {}
    for line in open(path):
        if "{" not in line:
            continue
        name = line[:line.index("{")].strip()
        vals = ast.literal_eval(line[line.index("{"):line.rindex("}") + 1])
        out[name] = vals
    return out


f, w = read(d + "/pmc_fetch.summary.txt"), read(d + "/pmc_write.summary.txt")
step = [k for k in f if "pf_grid_step" in k][0]
fin = [k for k in f if "calib_copy" in k][0]
known = (1 << 28) * 8
rf = known / (f[fin]["FETCH_SIZE"] * 1024.0)
wf = known / (w[fin]["WRITE_SIZE"] * 1024.0)
wf_used = 1.0 if abs(wf - 1.0) < 0.05 else wf
rd = f[step]["FETCH_SIZE"] * 1024.0 * rf
wr = w[step]["WRITE_SIZE"] * 1024.0 * wf_used
print(json.dumps({
    "replicas": R, "particles": N, "kernel": step, "particles_per_thread": int(step.split(",")[2].strip(" >")) if step.count(",") >= 2 else 1,
    "command": "scripts/profile_round.sh (separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of bench.py --steps 1 --warmup 0)",
    "FETCH_SIZE_KB_per_launch": f[step]["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": w[step]["WRITE_SIZE"],
    "calibration": {"kernel": fin, "known_bytes_each_way": known, "read_factor": rf, "write_factor_measured": wf,
                    "write_factor_used": wf_used,
                    "note": "gfx950 FETCH_SIZE reports half the fetched bytes (MI355X_MICROARCH.md, HBM section)"},
    "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
    "algorithmic_bytes_per_launch": R * N * (16 * m + 24)}, indent=1))
