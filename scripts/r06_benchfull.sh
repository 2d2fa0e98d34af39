#!/bin/bash
# which leg of the default bench line slows the ~N(0,1) build behind it: one flag removed at a time
OUT=gpurun_out/r06t; mkdir -p $OUT
for FLAG in "--no-e2e" "--no-cpu"; do
AH_TIMING=1 timeout 1200 python bench.py $FLAG > $OUT/benchfull.json 2> $OUT/benchfull.err
echo "== $FLAG"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06t/benchfull.json").read().strip().split("\n")[-1])
b = d["build_10m"]
print("main", b["seconds_samples"], b["seconds_after_device"])
for n in ("normal", "clustered"):
    print(n, b[n]["seconds_samples"], b[n]["seconds_device"], b[n]["seconds_after_device"])
PY
done
