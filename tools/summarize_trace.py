"""per-kernel duration summary from rocprofv3 kernel_trace.csv"""
import csv, sys, collections
acc = collections.defaultdict(list)
extra = {}
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        k = row["Kernel_Name"].split("(")[0][:70]
        acc[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
        extra[k] = (row.get("VGPR_Count"), row.get("SGPR_Count"), row.get("LDS_Block_Size"), row.get("Workgroup_Size"), row.get("Grid_Size"))
with open(sys.argv[2], "w") as out:
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        out.write(f"{k:70s} calls={len(v):4d} avg_us={sum(v)/len(v):10.2f} min_us={min(v):10.2f} max_us={max(v):10.2f} vgpr/sgpr/lds/wg/grid={extra[k]}\n")
