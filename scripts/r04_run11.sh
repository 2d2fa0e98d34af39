#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04k
mkdir -p $OUT
timeout 400 python scripts/fuzz_gpu.py 150 51 > $OUT/fuzz_51.log 2>&1; echo "fuzz 51 rc=$?"; tail -2 $OUT/fuzz_51.log
AH_SCREEN_VERIFY=1 timeout 300 python scripts/fuzz_gpu.py 90 52 > $OUT/fuzz_52.log 2>&1; echo "fuzz 52 rc=$?"; tail -2 $OUT/fuzz_52.log
AH_TIMING=1 timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
j = json.load(open("gpurun_out/r04k/bench.json"))
b = j["build_10m"]
print("value", j["value"], "frac", j["roofline"]["frac"])
print("build_10m", b["seconds"], b["seconds_samples"], "dev", b["seconds_device"], "setup", b["seconds_setup"], "after", b["seconds_after_device"], b["identical"])
print("stream", {k: v for k, v in b["stream"].items() if k != "sink"})
print("cold", {k: v for k, v in b["cold"].items() if k != "workload"})
print("share", b["share_13"]["seconds"], b["share_13"]["speedup_100_trees_over_share"], b["share_13"]["speedup_from_device_seconds"])
print("normal", b["normal"]["seconds"], "f32", b["f32_only"]["seconds"])
s = j["search"]
print("search", {k: round(v["queries_per_s"]) for k, v in s.items() if isinstance(v, dict) and "queries_per_s" in v}, s["verified"])
r = j["rerank"]
print("rerank", {k: round(v["queries_per_s"]) for k, v in r.items() if isinstance(v, dict) and "queries_per_s" in v})
PY
