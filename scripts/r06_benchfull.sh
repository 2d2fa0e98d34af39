#!/bin/bash
# the default bench line with and without the NUMA placement of the build's output path (AH_NUMA): the ~N(0,1) build behind the CPU legs
OUT=gpurun_out/r06t; mkdir -p $OUT
python - <<'PY'
import ctypes, sys
sys.path.insert(0, ".")
from arroy_amd import _lib
L = _lib.lib()
print("device 0 pci / numa:", open("/proc/self/status").read().split("Cpus_allowed_list:")[1].split()[0])
PY
for i in 0; do cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr '\n' ' '; echo; done
for NU in 1 0 1; do
AH_NUMA=$NU timeout 1200 python bench.py > $OUT/benchfull.json 2> $OUT/benchfull.err
echo "== AH_NUMA=$NU"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06t/benchfull.json").read().strip().split("\n")[-1])
b = d["build_10m"]
print("main", b["seconds_samples"], b["seconds_after_device"], "stream", b["stream"]["seconds"], "share", b["share_13"]["seconds"], "cold", b["cold"].get("total_s"))
for n in ("normal", "clustered"):
    print(n, b[n]["seconds_samples"], b[n]["seconds_device"], b[n]["seconds_after_device"])
print("rerank", d["rerank"]["callers_1"]["queries_per_s"], "search", d["search"]["callers_1_distinct_items"]["queries_per_s"])
PY
done
