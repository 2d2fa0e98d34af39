#!/bin/bash
# after the last harness change (median of five passes in the caller legs): the default bench line -> profiles/r04_bench.json,
# and two more seeds of the randomised sweep (the second with the screens' verify mode on)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04z
mkdir -p $OUT
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; echo "bench rc=$?"; grep real $OUT/bench.time
timeout 400 python scripts/fuzz_gpu.py 150 63 > $OUT/fuzz_63.log 2>&1; echo "fuzz 63 rc=$?"; tail -1 $OUT/fuzz_63.log | cut -c1-300
AH_SCREEN_VERIFY=1 timeout 400 python scripts/fuzz_gpu.py 150 64 > $OUT/fuzz_64.log 2>&1; echo "fuzz 64 rc=$?"; tail -1 $OUT/fuzz_64.log | cut -c1-300
python - <<'PY'
import json
j = json.load(open("gpurun_out/r04z/bench.json"))
b = j["build_10m"]
print("value", j["value"], "frac", j["roofline"]["frac"], "traffic", j["roofline"]["traffic"], j["roofline"]["traffic_source"][:40])
print("build_10m", b["seconds"], b["seconds_samples"], "dev", b["seconds_device"], b["identical"], "cold", b["cold"]["first_build_s"], b["cold"]["total_s"])
print("stream", b["stream"]["seconds"], "share", b["share_13"]["seconds"], b["share_13"]["speedup_100_trees_over_share"], "normal", b["normal"]["seconds"], "f32", b["f32_only"]["seconds"])
s = j["search"]; print("search", {k: round(v["queries_per_s"]) for k, v in s.items() if isinstance(v, dict) and "queries_per_s" in v}, s["verified"])
r = j["rerank"]; print("rerank", {k: round(v["queries_per_s"]) for k, v in r.items() if isinstance(v, dict) and "queries_per_s" in v}, r["roofline"]["frac"], r["roofline"]["traffic"])
print("bq", j["bq_scan"]["roofline"]["frac"], j["bq_scan"]["roofline"]["traffic"])
PY
