#!/bin/bash
# GPU call: suite + the search-side bench legs (re-rank / 1-bit scan / search) after the pipelined gather
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03d
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-build --no-cpu 2>$OUT/bench.err | tail -1 > $OUT/bench_search.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03d/bench_search.json'))
r=d['rerank']
print({k:(round(v['queries_per_s']),round(v.get('gb_per_s',v.get('effective_gb_per_s',0)))) for k,v in r.items() if isinstance(v,dict) and 'queries_per_s' in v})
print('search', {k:round(v['queries_per_s']) for k,v in d['search'].items() if isinstance(v,dict)})
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python bench.py --steps 20 --warmup 5 --no-build --no-cpu > $OUT/kt.log 2>&1
cp $OUT/kt/kt_kernel_stats.csv $OUT/kernel_stats_search.csv; rm -rf $OUT/kt
head -12 $OUT/kernel_stats_search.csv | cut -c1-160
