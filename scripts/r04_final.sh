#!/bin/bash
# what the driver runs at round end, in its order: the GPU suite, smoke(), the default bench line
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04z
mkdir -p $OUT
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -2 $OUT/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log | cut -c1-300
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
j = json.load(open("gpurun_out/r04z/bench.json"))
b = j["build_10m"]
print("value", j["value"], "frac", j["roofline"]["frac"], "traffic", j["roofline"]["traffic"])
print("build_10m", b["seconds"], b["seconds_samples"], "dev", b["seconds_device"], "setup", b["seconds_setup"], "after", b["seconds_after_device"], b["identical"])
print("stream", b["stream"]["seconds"], b["stream"]["seconds_samples"])
print("cold", {k: v for k, v in b["cold"].items() if k != "workload"})
print("share", b["share_13"]["seconds"], b["share_13"]["seconds_samples"], b["share_13"]["speedup_100_trees_over_share"])
print("normal", b["normal"]["seconds"], "f32", b["f32_only"]["seconds"])
s = j["search"]
print("search", {k: round(v["queries_per_s"]) for k, v in s.items() if isinstance(v, dict) and "queries_per_s" in v}, s["verified"])
r = j["rerank"]
print("rerank", {k: round(v["queries_per_s"]) for k, v in r.items() if isinstance(v, dict) and "queries_per_s" in v})
print("build 1M", j["build"]["seconds"])
PY
