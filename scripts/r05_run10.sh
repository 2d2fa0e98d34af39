#!/bin/bash
# round 5, call 10: the judged bench line as the driver runs it
set -x
OUT=gpurun_out/r05j
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench.json
tail -5 $OUT/bench.err
python - <<'PY'
import json
j = json.load(open("gpurun_out/r05j/bench.json"))
b = j["build_10m"]
print("value", j["value"], "frac", j["roofline"]["frac"], "traffic", j["roofline"]["traffic"], j["roofline"]["kernel_ms"])
print("build_10m", b["seconds"], b["seconds_samples"], "dev", b["seconds_device"], b["identical"], "after", b["seconds_after_device"])
print("stream", b["stream"]["seconds"], b["stream"]["seconds_after_device"], "share", b["share_13"]["seconds"], b["share_13"]["seconds_device"], b["share_13"]["speedup_100_trees_over_share"], "normal", b["normal"]["seconds"], "f32", b["f32_only"]["seconds"])
print("union", j.get("build_10m_union"))
s = j["search"]; print("search", {k: round(v["queries_per_s"]) for k, v in s.items() if isinstance(v, dict) and "queries_per_s" in v}, s["verified"])
l = s["latency"]; print("latency", {k: (round(v["p50_us"]), round(v["p99_us"])) for k, v in l.items() if isinstance(v, dict) and "p50_us" in v}, l["cpu_one_core"])
r = j["rerank"]; print("rerank", {k: round(v["queries_per_s"]) for k, v in r.items() if isinstance(v, dict) and "queries_per_s" in v}, r["roofline"]["frac"], r["roofline_screened"]["frac"])
print("bq", j["bq_scan"]["roofline"]["frac"], "read", j["roofline"]["measured_read_only_gb_per_s"])
c = j["cpu_baseline"]; print("cpu", c["value"], c.get("build_seconds_config_1"))
PY
