#!/bin/bash
OUT=gpurun_out/r06t; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_tail_groups.py tests/test_gpu_stream.py tests/test_gpu_faults.py -x -q -m gpu > $OUT/tests_host.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests_host.log
AH_TIMING=3 timeout 300 python scripts/exp_build.py 10000000 100 4 > $OUT/exp6.log 2>&1
grep "^{" $OUT/exp6.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('exp wall %.4f device %.4f after %.4f' % (d['wall'], d['seconds_device'], d['seconds_after_device']))
"
grep -n "level 1[34]\|host:" $OUT/exp6.log | tail -20 | cut -c1-170
for G in 0 5; do AH_BUILD_TAIL_GROUPS=$G timeout 300 python scripts/exp_build.py 10000000 100 4 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('groups %d wall %.4f device %.4f after %.4f' % (d['tail_groups'], d['wall'], d['seconds_device'], d['seconds_after_device']))
" | tail -3; done
timeout 300 python scripts/exp_stream.py 10000000 100 3 2>&1 | tail -2
timeout 300 python scripts/exp_build.py 10000000 13 4 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('13 trees: wall %.4f device %.4f after %.4f' % (d['wall'], d['seconds_device'], d['seconds_after_device']))
" | tail -2
