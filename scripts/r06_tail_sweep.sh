#!/bin/bash
# groups x ratio of the build's tail (AH_BUILD_TAIL_GROUPS, AH_BUILD_TAIL_RATIO), 10M x 768 x 100 trees, the last three of four builds each
for CFG in "5 76" "6 76" "6 85" "4 76" "7 80" "5 76"; do
  set -- $CFG
  echo "== groups $1 ratio $2"
  AH_BUILD_TAIL_GROUPS=$1 AH_BUILD_TAIL_RATIO=$2 timeout 300 python scripts/exp_build.py 10000000 100 4 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('wall %.4f total %.4f device %.4f after %.4f groups %d' % (d['wall'], d['seconds_total'], d['seconds_device'], d['seconds_after_device'], d['tail_groups']))
" | tail -3
done
