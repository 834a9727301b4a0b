"""sum rocprofv3 counter_collection.csv per (kernel, counter): prints mean value per dispatch"""
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        k = row.get("Kernel_Name", "?").split("(")[0][:60]
        if "rans" not in k and "tans" not in k and "range" not in k and "aec" not in k and "cp_" not in k:
            continue
        key = (k, row.get("Counter_Name"))
        acc[key][0] += float(row.get("Counter_Value", 0))
        acc[key][1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print(f"{k:60s} {c:28s} mean/dispatch={v / n:.6g}  dispatches={n}")
