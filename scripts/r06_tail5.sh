#!/bin/bash
# the build_10m leg inside bench.py against the same build from scripts/exp_build.py, on one box: where the device time differs
OUT=gpurun_out/r06t; mkdir -p $OUT
AH_TIMING=3 timeout 900 python bench.py --steps 5 --warmup 1 --no-cpu --no-extra --no-live-pmc --no-e2e > $OUT/bench5.json 2> $OUT/bench5.err
grep -n "batch of 100 trees\|tail:" $OUT/bench5.err | cut -c1-260 | head -20
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06t/bench5.json").read().strip().split("\n")[-1])
b = d["build_10m"]
print("bench", b["seconds_samples"], b["seconds_device"], b["seconds_after_device"], b.get("level_by_level"))
PY
AH_TIMING=3 timeout 300 python scripts/exp_build.py 10000000 100 3 > $OUT/exp5.log 2>&1
grep "batch of 100 trees\|^{" $OUT/exp5.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('exp wall %.4f device %.4f after %.4f' % (d['wall'], d['seconds_device'], d['seconds_after_device']))
    else: print(l.strip()[:200])
"
