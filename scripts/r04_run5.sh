#!/bin/bash
# round 4, GPU call 5: the certified top-k screen of the tile re-rank: every search test, then the search leg of the bench
mkdir -p gpurun_out/r04e
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_index.py tests/test_gpu_search_scale.py tests/test_gpu_multi_device.py tests/test_gpu_staging.py -q -m gpu -k "search or index or multi or device or shim or golden or snapshot" > gpurun_out/r04e/tests.log 2>&1
echo "tests rc=$?"; tail -15 gpurun_out/r04e/tests.log
timeout 600 python - > gpurun_out/r04e/search_bench.json 2> gpurun_out/r04e/search_bench.err <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
out = {}
from arroy_amd import _lib
out["screen_on"] = bench.extra_search(0)
with _lib.tuning(AH_SEARCH_SCREEN=0):
    out["screen_off"] = bench.extra_search(0)
print(json.dumps(out))
PY
echo "search bench rc=$?"
python - <<'PY'
import json
j = json.load(open("gpurun_out/r04e/search_bench.json"))
for k, v in j.items():
    print(k, {a: (round(b["queries_per_s"]) if isinstance(b, dict) and "queries_per_s" in b else b) for a, b in v.items() if a.startswith("callers") or a == "verified"})
    print("   ", {a: v["stats"][a] for a in ("queries", "rerank_screened", "screen_survivors", "rerank_tiles", "rerank_sorted", "fallback_chunks")})
PY
