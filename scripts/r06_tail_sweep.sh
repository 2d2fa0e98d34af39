#!/bin/bash
OUT=gpurun_out/r06t; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_tail_groups.py tests/test_gpu_stream.py -x -q -m gpu > $OUT/tests_sweep.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests_sweep.log
for CFG in "4 100" "4 76" "5 76" "5 100" "6 100" "6 85"; do
  set -- $CFG
  echo "== groups $1 ratio $2"
  AH_BUILD_TAIL_GROUPS=$1 AH_BUILD_TAIL_RATIO=$2 AH_TIMING=1 timeout 300 python scripts/exp_build.py 10000000 100 4 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('wall %.4f total %.4f device %.4f after %.4f groups %d' % (d['wall'], d['seconds_total'], d['seconds_device'], d['seconds_after_device'], d['tail_groups']))
" | tail -3
done
echo "== stream"
for R in 100 76; do AH_BUILD_TAIL_RATIO=$R timeout 300 python scripts/exp_stream.py 10000000 100 3 2>&1 | tail -2; done
echo "== 13 trees"
for R in 100 76; do AH_BUILD_TAIL_RATIO=$R timeout 300 python scripts/exp_build.py 10000000 13 4 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('wall %.4f total %.4f device %.4f after %.4f groups %d' % (d['wall'], d['seconds_total'], d['seconds_device'], d['seconds_after_device'], d['tail_groups']))
" | tail -2; done
