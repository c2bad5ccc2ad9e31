"""Static VALU / SALU / memory instruction counts between the V2MARK comments of a -DBSSM_V2_MARKS listing:
python scripts/isa_phases.py file.s kernel-substring"""
import collections
import sys
lines = open(sys.argv[1]).read().split("\n")
sym = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sym in l and ":" in l)
cur = "prologue"; order = [cur]; c = collections.defaultdict(collections.Counter)
for l in lines[start + 1:]:
    if ".end_amdhsa_kernel" in l: break
    s = l.strip()
    if "V2MARK" in s:
        cur = s.split("V2MARK")[1].strip()
        if cur not in order: order.append(cur)
        continue
    if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"): continue
    op = s.split()[0]
    k = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem"
    c[cur][k] += 1
    if op.startswith("v_readlane") or op.startswith("v_writelane"): c[cur]["lane"] += 1
    if op.startswith("v_mov"): c[cur]["mov"] += 1
for k in order:
    print(f"{k:10s}", dict(c[k]))
