#!/bin/bash
# last call of the round: the whole suite, the smoke, the judged bench line (with the PMC traffic files in place)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/final
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
python bench.py --steps 50 --warmup 5 2>$OUT/bench.err | tail -1 > $OUT/r03_bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/final/r03_bench.json'))
print(d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['build_10m']['seconds'], d['build_10m']['identical'], d['build_10m']['share_13']['seconds'], d['build_10m']['normal']['seconds'], d['rerank']['roofline']['traffic'], d['bq_scan']['roofline']['traffic'])
PY
