#!/bin/bash
# round 5, call 6: fault-injection tests, the virtual N > 1 bench path, the latency leg (search only)
set -x
OUT=gpurun_out/r05f
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_faults.py -x -q -m gpu 2>&1 | tail -30
timeout 1500 python -m pytest tests/test_gpu_multi_device.py -x -q -m gpu 2>&1 | tail -30
timeout 900 python - <<'PY' 2>&1 | tail -40
import json, sys
sys.argv = ["bench.py"]
import bench
r = bench.extra_search(0)
print(json.dumps({k: r[k] for k in ("latency", "callers_1", "callers_1_distinct_items", "verified")}, indent=1))
PY
