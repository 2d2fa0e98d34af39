#!/bin/bash
OUT=gpurun_out/r06s; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_stream.py -x -q -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
AH_TIMING=3 timeout 300 python scripts/exp_stream.py 10000000 100 3 2>&1 | grep "level 1[234]\|host: \|streamed batch\|wall" | tail -16 | cut -c1-200
