"""Static instruction mix of one kernel in a hipcc -S listing: python scripts/isa_count.py file.s <mangled-substring>"""
import collections
import re
import sys

path, sym = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sym in l and l.rstrip().split(":")[0].endswith(l.split(":")[0]))
mix = collections.Counter()
ops = collections.Counter()
for l in lines[start + 1:]:
    if l.startswith("\t.end_amdhsa_kernel") or l.startswith(".Lfunc_end"):
        break
    s = l.strip()
    if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
        continue
    op = s.split()[0]
    ops[op] += 1
    if op.startswith("v_"):
        mix["valu"] += 1
    elif op.startswith("s_"):
        mix["salu" if not op.startswith(("s_load", "s_buffer", "s_waitcnt", "s_barrier", "s_nop")) else op.split("_")[1]] += 1
    elif op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        mix["vmem"] += 1
    elif op.startswith("ds_"):
        mix["lds"] += 1
    else:
        mix["other"] += 1
print(dict(mix))
for k, v in ops.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
    print(f"{v:6d} {k}")
for l in lines[start:]:
    if re.search(r"(NumVgprs|NumSgprs|ScratchSize|Occupancy|sgpr_spill|vgpr_spill|NumAgprs)", l):
        print(l.strip())
    if l.startswith(".Lfunc_end"):
        pass
    if "; -- End function" in l:
        break
