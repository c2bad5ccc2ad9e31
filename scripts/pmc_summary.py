"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per dispatch for each kernel."""
import sys, glob, csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0][:60]
        a = acc[k][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
for k, cs in sorted(acc.items()):
    print(k, {c: round(v[0] / v[1], 1) for c, v in cs.items()}, "dispatches", max(v[1] for v in cs.values()))
