"""Summarise a rocprofv3 --pmc counter_collection csv: per kernel name (template arguments kept) and counter, the number of
dispatches, the mean and the sum.  For the forest margin kernels the dispatches are additionally listed in launch order."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
agg = defaultdict(lambda: [0, 0.0])
seq = defaultdict(dict)
for r in rows:
    name = r["Kernel_Name"].split("(")[0][:90]
    key = (name, r["Counter_Name"])
    agg[key][0] += 1
    agg[key][1] += float(r["Counter_Value"])
    if "k_forest_margin" in name:
        seq[(int(r["Dispatch_Id"]), name)][r["Counter_Name"]] = float(r["Counter_Value"])
print("kernel,counter,dispatches,mean,sum")
for (name, c), (n, s) in sorted(agg.items()):
    print(f"{name},{c},{n},{s / n:.6g},{s:.6g}")
print()
print("# forest margin dispatches in launch order")
for (d, name), cs in sorted(seq.items()):
    print(d, name, " ".join(f"{k}={v:.6g}" for k, v in sorted(cs.items())))
