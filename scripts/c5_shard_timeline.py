"""Timeline of ONE 293-draw post-correction call from a rocprofv3 kernel trace (the last call of scripts/c5_quick.py): when the set-up, the
last full round (pf_wg_kernel / pf_wg_kernel_lean) and the tail's row steps start and end, relative to the set-up's start.
  rocprofv3 --kernel-trace --output-format csv -d DIR -- python scripts/c5_quick.py 293 4;  python scripts/c5_shard_timeline.py DIR/.../*_kernel_trace.csv"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Stream_Id"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
wg = [r for r in rows if "pf_wg_kernel" in r[2]]
s0, e0 = wg[-1][0], wg[-1][1]
setup = [r for r in rows if "fn_setup_kernel" in r[2] and r[0] < s0][-4:]
t0 = setup[0][0]
steps = [r for r in rows if "pf_row_step" in r[2] and r[0] > t0]
us = lambda x: (x - t0) / 1e3
print("set-up kernels            %8.1f ... %8.1f us" % (us(setup[0][0]), us(max(r[1] for r in setup))))
print("%-24s  %8.1f ... %8.1f us  (%s)" % ("last full round", us(s0), us(e0), wg[-1][2].split("(")[0].replace("void bssm::", "")))
for sid in sorted(set(r[3] for r in steps)):
    v = [r for r in steps if r[3] == sid]
    dur = sum(r[1] - r[0] for r in v) / len(v) / 1e3
    in_round = [r for r in v if r[0] < e0]
    print("tail steps, stream %-6s %8.1f ... %8.1f us  %d launches, mean duration %.2f us, mean period %.2f us; %d of them started before the round ended"
          % (sid, us(v[0][0]), us(v[-1][1]), len(v), dur, (v[-1][0] - v[0][0]) / max(1, len(v) - 1) / 1e3, len(in_round)))
last = max(r[1] for r in rows if r[0] >= t0)
print("last kernel of the call ends at %8.1f us" % us(last))
