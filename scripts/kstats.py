"""rocprofv3 kernel_stats csv -> one short line per kernel: name (template arguments kept, parameters dropped), calls, average us."""
import csv
import re
import sys

rows = list(csv.reader(open(sys.argv[1])))
want = sys.argv[2:]
for r in rows[1:]:
    name = re.sub(r"\(.*", "", r[0]).replace("ah::", "").replace("void ", "")
    if want and not any(w in name for w in want):
        continue
    calls, total, avg = r[-7], r[-6], r[-5]
    print(f"{name:48s} calls {calls:>5s}  avg {float(avg) / 1000:10.1f} us  total {float(total) / 1e6:9.2f} ms")
