"""Highest VGPR index referenced per window of instructions of one kernel: python scripts/vgpr_profile.py file.s name-substring [window]"""
import re
import sys
lines = open(sys.argv[1]).read().split("\n")
sym = sys.argv[2]
win = int(sys.argv[3]) if len(sys.argv) > 3 else 100
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sym in l and l.rstrip().endswith(l.split(":")[0] + ":") or (l.startswith("_Z") and sym in l and ": " in l))
cnt = 0; mx = 0; out = []; mark = ""
for l in lines[start + 1:]:
    if ".end_amdhsa_kernel" in l or l.startswith(".Lfunc_end"):
        break
    s = l.strip()
    if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
        continue
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", s):
        hi = int(m.group(2)) if m.group(2) else int(m.group(3))
        mx = max(mx, hi)
    op = s.split()[0]
    if op in ("s_barrier",): mark += "B"
    if op.startswith("v_mad_u64_u32"): mark += "p"
    if op.startswith("global_load"): mark += "l"
    if op.startswith("global_store"): mark += "s"
    if op.startswith("scratch_"): mark += "!"
    if op.startswith("v_rsq"): mark += "q"
    cnt += 1
    if cnt % win == 0:
        out.append((cnt, mx, mark)); mx = 0; mark = ""
for c, m, k in out:
    print(f"{c:6d} maxv {m:4d} {k[:120]}")
