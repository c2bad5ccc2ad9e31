"""Register / LDS / spill summary of the kernels in a hipcc -S listing: python scripts/kernel_regs.py file.s [name-substring]"""
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
meta = s[s.index("amdhsa.kernels:"):]
for blk in meta.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if pat not in name:
        continue
    def g(k):
        m = re.search(r"\." + k + r":\s+(\d+)", blk)
        return m.group(1) if m else "?"
    print(name[:70], "vgpr", g("vgpr_count"), "vspill", g("vgpr_spill_count"), "sgpr", g("sgpr_count"), "sspill", g("sgpr_spill_count"),
          "lds", g("group_segment_fixed_size"), "scratch", g("private_segment_fixed_size"))
