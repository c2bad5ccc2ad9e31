"""Per-queue cadence of a kernel in a rocprofv3 kernel trace: python scripts/trace_gaps.py kernel_trace.csv name-substring
-> per queue: launches, mean duration, mean gap to the next launch of the same queue, mean period (start to start)"""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
byq = collections.defaultdict(list)
for r in rows:
    byq[(r["Queue_Id"], r["Stream_Id"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Grid_Size_X"], r["Grid_Size_Y"], r["VGPR_Count"]))
for q, v in sorted(byq.items()):
    v.sort()
    dur = [e - s for s, e, *_ in v]
    gaps = [v[i + 1][0] - v[i][1] for i in range(len(v) - 1) if v[i + 1][0] - v[i][1] < 50000]
    per = [v[i + 1][0] - v[i][0] for i in range(len(v) - 1) if v[i + 1][0] - v[i][0] < 50000]
    print("queue/stream %s: %d launches grid %sx%s vgpr %s | duration %.2f us | gap %.2f us | period %.2f us" % (
        q, len(v), v[0][2], v[0][3], v[0][4], sum(dur) / len(dur) / 1e3, sum(gaps) / max(1, len(gaps)) / 1e3, sum(per) / max(1, len(per)) / 1e3))
