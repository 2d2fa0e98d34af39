#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/final
python bench.py --steps 50 --warmup 5 2>gpurun_out/final/bench.err | tail -1 > gpurun_out/final/r03_bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/final/r03_bench.json'))
r=d['rerank']
print(d['value'], d['roofline']['traffic'], d['build_10m']['seconds'], d['build_10m']['identical'], {k:round(v['queries_per_s']) for k,v in r.items() if isinstance(v,dict) and 'queries_per_s' in v}, r['roofline']['traffic'], d['search']['callers_1']['queries_per_s'])
PY
