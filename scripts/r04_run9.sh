#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04i
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_index.py -q -m gpu -x -k "rerank or top_k or topk or full_size_properties_1m_x_1536" > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -4 $OUT/tests.log
timeout 600 python - > $OUT/rerank.json 2> $OUT/rerank.err <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
print(json.dumps(bench.extra_c4(0)))
PY
python - <<'PY'
import json
j = json.load(open("gpurun_out/r04i/rerank.json"))
for k, v in j.items():
    if isinstance(v, dict) and "queries_per_s" in v:
        print(k, round(v["queries_per_s"]), {a: round(b, 1) for a, b in v.items() if a.endswith("gb_per_s")})
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python - > $OUT/prof.log 2>&1 <<'PY'
import sys
sys.path.insert(0, ".")
import bench
bench.extra_c4(0)
PY
python scripts/kstats.py $OUT/kt/kt_kernel_stats.csv k_pairs k_search_select k_queries k_batch k_prepare | head -12
rm -rf $OUT/kt
