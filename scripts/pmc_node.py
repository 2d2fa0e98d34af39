"""Counters of the big k_forest_screen_node dispatches (deep levels of the 10M build) from a rocprofv3 --pmc csv."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.OrderedDict()
for r in rows:
    if sys.argv[2] not in r["Kernel_Name"]:
        continue
    key = (r["Dispatch_Id"], r["Grid_Size"])
    by.setdefault(key, {})[r["Counter_Name"]] = by.setdefault(key, {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for (d, g), c in by.items():
    if int(g) < 256 * 50000:
        continue
    print(d, "grid", g, " ".join(f"{k}={v:.4g}" for k, v in sorted(c.items())))
