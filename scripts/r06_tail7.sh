#!/bin/bash
OUT=gpurun_out/r06t; mkdir -p $OUT
AH_TIMING=1 timeout 300 python scripts/exp_build.py 10000000 100 4 2>&1 | grep "ah\] build:\|^{" | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('wall %.4f device %.4f after %.4f' % (d['wall'], d['seconds_device'], d['seconds_after_device']))
    else: print(l.strip()[:200])
"
timeout 300 python scripts/exp_stream.py 10000000 100 3 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_tail_groups.py tests/test_gpu_stream.py tests/test_gpu_faults.py tests/test_gpu_staging.py -x -q -m gpu > $OUT/tests7.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests7.log
