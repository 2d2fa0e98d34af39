#!/bin/bash
# the round's last state as the first process on a box: the default bench line -> profiles/r04_bench.json, the per-level timing
# of the 13-tree share -> profiles/r04_levels_timing_13trees.txt, then two more seeds of the randomised sweep
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04z
mkdir -p $OUT
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; echo "bench rc=$?"; grep real $OUT/bench.time
AH_TIMING=2 python scripts/exp_build.py 10000000 13 2 2>&1 | tail -19 > $OUT/r04_levels_timing_13trees.txt
timeout 300 python scripts/fuzz_gpu.py 100 65 > $OUT/fuzz_65.log 2>&1; echo "fuzz 65 rc=$?"; tail -1 $OUT/fuzz_65.log | cut -c1-300
python - <<'PY'
import json
j = json.load(open("gpurun_out/r04z/bench.json"))
b = j["build_10m"]
print("value", j["value"], "frac", j["roofline"]["frac"], "traffic", j["roofline"]["traffic"], j["roofline"]["traffic_source"][:40])
print("build_10m", b["seconds"], b["seconds_samples"], "dev", b["seconds_device"], b["identical"], "cold", {k: v for k, v in b["cold"].items() if k.endswith("_s")})
print("stream", b["stream"]["seconds"], "share", b["share_13"]["seconds"], b["share_13"]["speedup_100_trees_over_share"], "normal", b["normal"]["seconds"], "f32", b["f32_only"]["seconds"])
s = j["search"]; print("search", {k: round(v["queries_per_s"]) for k, v in s.items() if isinstance(v, dict) and "queries_per_s" in v}, s["verified"])
r = j["rerank"]; print("rerank", {k: round(v["queries_per_s"]) for k, v in r.items() if isinstance(v, dict) and "queries_per_s" in v}, r["roofline"]["frac"], r["roofline"]["traffic"])
print("bq", j["bq_scan"]["roofline"]["frac"], j["bq_scan"]["roofline"]["traffic"])
PY
