#!/bin/bash
OUT=gpurun_out/r06t; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_tail_groups.py -x -q -m gpu > $OUT/tests2.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests2.log
for CFG in "4 2" "4 4" "4 16" "3 2" "5 2" "6 2"; do
  set -- $CFG
  echo "== groups $1 node_items $2"
  AH_BUILD_TAIL_GROUPS=$1 AH_BUILD_TAIL_NODE_ITEMS=$2 AH_TIMING=2 timeout 300 python scripts/exp_build.py 10000000 100 2 2>&1 | python -c "
import sys, json
lines = sys.stdin.read().split('\n')
# the second build only
k = [i for i, l in enumerate(lines) if l.startswith('{')]
for l in lines[k[0] + 1:]:
    if l.startswith('{'):
        d = json.loads(l); print('wall %.4f total %.4f device %.4f after %.4f groups %d' % (d['wall'], d['seconds_total'], d['seconds_device'], d['seconds_after_device'], d['tail_groups']))
    elif 'batch of' in l or 'tail:' in l: print(l.strip()[:200])
    elif 'level 1' in l and 'level 1:' not in l: print(l.strip()[:150])
"
done
