#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04j
mkdir -p $OUT
for g in 1 2 3 4; do
AH_RERANK_GROUPS=$g timeout 600 python - > $OUT/rerank_$g.json 2> $OUT/rerank_$g.err <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
print(json.dumps(bench.extra_c4(0)))
PY
python - $g <<'PY'
import json, sys
j = json.load(open(f"gpurun_out/r04j/rerank_{sys.argv[1]}.json"))
print("groups", sys.argv[1], {k: round(v["queries_per_s"]) for k, v in j.items() if isinstance(v, dict) and "queries_per_s" in v})
PY
done
