#!/bin/bash
OUT=gpurun_out/r06p; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python scripts/exp_latency.py 1 200 > $OUT/lat.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r06p/kt/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 150 calls: sequences descend_multi -> leaf_tiles16 -> search_select
seq = [(r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ah::", "").replace("ah::", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
gaps = collections.defaultdict(list)
for i in range(1, len(seq)):
    a, b = seq[i - 1], seq[i]
    if a[0] == "k_descend_multi" and b[0] == "k_leaf_tiles16":
        gaps["descend->tiles gap"].append(b[1] - a[2]); gaps["descend dur"].append(a[2] - a[1])
    if a[0] == "k_leaf_tiles16" and b[0] == "k_search_select_screened":
        gaps["tiles->select gap"].append(b[1] - a[2]); gaps["tiles dur"].append(a[2] - a[1]); gaps["select dur"].append(b[2] - b[1])
    if a[0] == "k_search_select_screened" and b[0] == "k_descend_multi":
        gaps["select->next descend (host turnaround)"].append(b[1] - a[2])
for k, v in gaps.items():
    v = sorted(v)[len(v) // 10: -len(v) // 10 or None]
    print(f"{k}: median {sorted(v)[len(v)//2] / 1000:.2f} us  mean {sum(v) / len(v) / 1000:.2f} us  n={len(v)}")
names = collections.Counter(s[0] for s in seq[-40:])
print(names)
PY
grep "^nq=" $OUT/lat.log | tail -1
rm -rf $OUT/kt
