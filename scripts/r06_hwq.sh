#!/bin/bash
# hardware queues per priority (GPU_MAX_HW_QUEUES, the runtime's own switch; default 4): the whole default bench line with 8
OUT=gpurun_out/r06t; mkdir -p $OUT
for Q in 8; do
GPU_MAX_HW_QUEUES=$Q timeout 1200 python bench.py > $OUT/hwq.json 2> $OUT/hwq.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r06t/hwq.json").read().strip().split("\n")[-1])
b = d["build_10m"]; r = d["rerank"]; s = d["search"]
print("GPU_MAX_HW_QUEUES=$Q value", d["value"], d["roofline"]["frac"], "build_1m", d["build"]["seconds"])
print("main", b["seconds_samples"], b["seconds_after_device"], "stream", b["stream"]["seconds"], "share", b["share_13"]["seconds"], "cold", b["cold"].get("total_s"))
for n in ("normal", "clustered"):
    print(n, b[n]["seconds_samples"], b[n]["seconds_after_device"])
print("rerank", {k: round(v["queries_per_s"]) for k, v in r.items() if isinstance(v, dict) and "queries_per_s" in v})
print("search", {k: round(v["queries_per_s"]) for k, v in s.items() if isinstance(v, dict) and "queries_per_s" in v})
l = s["latency"]; print("latency", {k: (round(v["p50_us"]), round(v["p99_us"])) for k, v in l.items() if isinstance(v, dict) and "p50_us" in v})
print("bq", d["bq_scan"]["roofline"]["frac"])
PY
done
