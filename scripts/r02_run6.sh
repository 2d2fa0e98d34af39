#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02i
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_staging.py -x -q > $OUT/pytest_staging.log 2>&1
AH_TIMING=1 timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu --no-build --no-extra --extra staging > $OUT/staging.json 2> $OUT/staging.err
AH_TIMING=1 AH_STAGE_THREADS=1 timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu --no-build --no-extra --extra staging > $OUT/staging1.json 2> $OUT/staging1.err
for i in 1 2 3; do timeout 300 python scripts/exp_build.py 1000000 50 3 | python -c "import sys,json; [print(json.loads(l)['wall'], json.loads(l)['seconds_device']) for l in sys.stdin]"; done > $OUT/b1m.txt 2>&1
timeout 300 python scripts/exp_build.py 10000000 100 2 | cut -c1-200 > $OUT/b10m.txt 2>&1
tail -n 4 $OUT/pytest_staging.log; cat $OUT/staging.err $OUT/staging1.err | grep "\[ah\]"; python -c "import json; print(json.load(open('$OUT/staging.json'))['extra'])"; cat $OUT/b1m.txt $OUT/b10m.txt
