#!/bin/bash
OUT=gpurun_out/r06t; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_tail_groups.py tests/test_gpu_stream.py -x -q -m gpu > $OUT/tests3.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests3.log
for CFG in "0 2" "4 2" "5 2" "6 2" "8 2"; do
  set -- $CFG
  echo "== groups $1 node_items $2"
  AH_BUILD_TAIL_GROUPS=$1 AH_BUILD_TAIL_NODE_ITEMS=$2 AH_TIMING=1 timeout 300 python scripts/exp_build.py 10000000 100 4 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('wall %.4f total %.4f device %.4f after %.4f groups %d' % (d['wall'], d['seconds_total'], d['seconds_device'], d['seconds_after_device'], d['tail_groups']))
    elif 'batch of' in l: print(l.strip()[60:230])
"
done
echo "== 13 trees"
for G in 0 4; do AH_BUILD_TAIL_GROUPS=$G timeout 300 python scripts/exp_build.py 10000000 13 4 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('wall %.4f total %.4f device %.4f after %.4f groups %d' % (d['wall'], d['seconds_total'], d['seconds_device'], d['seconds_after_device'], d['tail_groups']))
"; done
