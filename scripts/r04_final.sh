#!/bin/bash
# what the driver runs at round end, in its order (GPU suite, smoke(), the default bench line -> profiles/r04_bench.json),
# then the randomised sweep with the round's final kernels
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04z
mkdir -p $OUT
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -2 $OUT/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log | cut -c1-300
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; echo "bench rc=$?"; grep real $OUT/bench.time
timeout 400 python scripts/fuzz_gpu.py 150 61 > $OUT/fuzz_61.log 2>&1; echo "fuzz 61 rc=$?"; tail -1 $OUT/fuzz_61.log | cut -c1-300
AH_SCREEN_VERIFY=1 timeout 400 python scripts/fuzz_gpu.py 150 62 > $OUT/fuzz_62.log 2>&1; echo "fuzz 62 rc=$?"; tail -1 $OUT/fuzz_62.log | cut -c1-300
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
