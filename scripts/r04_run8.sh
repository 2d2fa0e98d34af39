#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04h
mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_schedules.py::test_baseline_config_3_all_100_trees_take_the_checked_paths > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -8 $OUT/tests.log
timeout 600 python - > $OUT/rerank.json 2> $OUT/rerank.err <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
print(json.dumps(bench.extra_c4(0)))
PY
python - <<'PY'
import json
j = json.load(open("gpurun_out/r04h/rerank.json"))
for k, v in j.items():
    if isinstance(v, dict) and "queries_per_s" in v:
        print(k, round(v["queries_per_s"]), {a: round(b, 1) for a, b in v.items() if a.endswith("gb_per_s")})
PY
