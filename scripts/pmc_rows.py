"""Counters of the row-major / dense margin dispatches (levels 0-9 of the 10M x 100-tree build) from a rocprofv3 --pmc csv:
one line per dispatch in launch order — kernel (template arguments kept), grid, every counter summed over its instances."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2:] or ["k_forest_screen_rows", "k_forest_dense_screen", "k_forest_exact_pairs"]
by = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"]
    if not any(w in name for w in want):
        continue
    short = re.sub(r"\(.*", "", name).replace("ah::", "").replace("void ", "")
    key = (int(r["Dispatch_Id"]), short, r["Grid_Size"])
    by.setdefault(key, {})
    by[key][r["Counter_Name"]] = by[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for (d, name, g), c in sorted(by.items()):
    print(d, name, "grid", g, " ".join(f"{k}={v:.5g}" for k, v in sorted(c.items())))
