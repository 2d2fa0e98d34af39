#!/bin/bash
for G in 0 5 3; do echo "== clustered groups $G"; AH_BUILD_TAIL_GROUPS=$G AH_TIMING=1 timeout 300 python scripts/exp_build.py 10000000 100 3 768 4 2>&1 | grep "tail:\|^{" | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('wall %.4f device %.4f after %.4f groups %d levels %d' % (d['wall'], d['seconds_device'], d['seconds_after_device'], d['tail_groups'], d['levels']))
    else: print(l.strip()[:150])
" | tail -3; done
echo "== low rank"; for G in 0 5; do AH_BUILD_TAIL_GROUPS=$G timeout 300 python scripts/exp_build.py 10000000 100 3 768 5 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('wall %.4f device %.4f after %.4f groups %d levels %d' % (d['wall'], d['seconds_device'], d['seconds_after_device'], d['tail_groups'], d['levels']))
" | tail -2; done
echo "== normal"; for G in 0 5; do AH_BUILD_TAIL_GROUPS=$G timeout 300 python scripts/exp_build.py 10000000 100 3 768 2 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('wall %.4f device %.4f after %.4f groups %d levels %d' % (d['wall'], d['seconds_device'], d['seconds_after_device'], d['tail_groups'], d['levels']))
" | tail -2; done
