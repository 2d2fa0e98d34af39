#!/bin/bash
# copy streams with a priority of their own (AH_READBACK_PRIORITY): the second dataset of a process, then the whole default bench line twice (1 / 0)
OUT=gpurun_out/r06t; mkdir -p $OUT
for P in 1 0; do AH_READBACK_PRIORITY=$P timeout 300 python scripts/exp_second_dataset.py keep 2>&1 | tail -1 | sed "s/^/AH_READBACK_PRIORITY=$P /"; done
for P in 1 0; do
AH_READBACK_PRIORITY=$P timeout 1200 python bench.py > $OUT/benchfull.json 2> $OUT/benchfull.err
echo "== whole bench line, AH_READBACK_PRIORITY=$P"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06t/benchfull.json").read().strip().split("\n")[-1])
b = d["build_10m"]
print("main", b["seconds_samples"], b["seconds_after_device"], "stream", b["stream"]["seconds"], "share", b["share_13"]["seconds"], "cold", b["cold"].get("total_s"))
for n in ("normal", "clustered"):
    print(n, b[n]["seconds_samples"], b[n]["seconds_device"], b[n]["seconds_after_device"])
r = d["rerank"]; print("rerank", {k: round(v["queries_per_s"]) for k, v in r.items() if isinstance(v, dict) and "queries_per_s" in v})
s = d["search"]; print("search", {k: round(v["queries_per_s"]) for k, v in s.items() if isinstance(v, dict) and "queries_per_s" in v})
PY
done
