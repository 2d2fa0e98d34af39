#!/bin/bash
# the build legs of bench.py with the library's own timing lines, in order
OUT=gpurun_out/r06t; mkdir -p $OUT
AH_TIMING=1 timeout 900 python bench.py --steps 5 --warmup 1 --no-cpu --no-extra --no-live-pmc --no-e2e > $OUT/benchlegs.json 2> $OUT/benchlegs.err
grep "batch of 100 trees\|streamed batch" $OUT/benchlegs.err | cut -c1-250
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06t/benchlegs.json").read().strip().split("\n")[-1])
b = d["build_10m"]
print("main", b["seconds_samples"], b["seconds_after_device"])
for n in ("normal", "clustered"):
    print(n, b[n]["seconds_samples"], b[n]["seconds_device"], b[n]["seconds_after_device"])
PY
free -g | head -2
