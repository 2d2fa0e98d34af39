#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04n
mkdir -p $OUT
# a process that takes and releases a lot of HBM right before: the state in which fresh allocations were slow
python scripts/exp_build.py 10000000 100 1 > /dev/null 2>&1
AH_TIMING=1 timeout 900 python bench.py --no-extra > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
grep "batch setup\|batch of 100" $OUT/bench.err | head -4 | cut -c1-250
python - <<'PY'
import json
j = json.load(open("gpurun_out/r04n/bench.json"))
b = j["build_10m"]
print("build_10m", b["seconds"], b["seconds_samples"], "dev", b["seconds_device"], "setup", b["seconds_setup"])
print("cold", {k: v for k, v in b["cold"].items() if k != "workload"})
print("share", b["share_13"]["seconds_samples"])
PY
python scripts/exp_build.py 10000000 100 1 > /dev/null 2>&1
AH_TIMING=1 timeout 900 python bench.py --no-extra --no-cpu > $OUT/bench2.json 2> $OUT/bench2.err; echo "bench2 rc=$?"
python - <<'PY'
import json
j = json.load(open("gpurun_out/r04n/bench2.json"))
b = j["build_10m"]
print("build_10m", b["seconds"], b["seconds_samples"], "dev", b["seconds_device"], "setup", b["seconds_setup"])
print("cold", {k: v for k, v in b["cold"].items() if k != "workload"})
PY
timeout 600 python -m pytest tests/test_gpu_staging.py tests/test_gpu_stream.py -q -m gpu -x 2>&1 | tail -2
