#!/bin/bash
# the tail in groups of trees: parity tests, then the 10M x 768 x 100-tree build with 0 / 4 / 8 groups (materialised and streamed)
OUT=gpurun_out/r06t; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_tail_groups.py tests/test_gpu_stream.py -x -q -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -5 $OUT/tests.log
for G in 0 4 8; do
  echo "== groups $G"
  AH_BUILD_TAIL_GROUPS=$G AH_TIMING=1 timeout 300 python scripts/exp_build.py 10000000 100 3 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('wall %.4f total %.4f device %.4f after %.4f groups %d' % (d['wall'], d['seconds_total'], d['seconds_device'], d['seconds_after_device'], d['tail_groups']))
    elif 'batch of' in l or 'tail:' in l: print(l.strip()[:260])
"
  AH_BUILD_TAIL_GROUPS=$G timeout 300 python scripts/exp_stream.py 10000000 100 3 2>&1 | tail -3
done
