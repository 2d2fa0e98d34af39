#!/bin/bash
OUT=gpurun_out/r06fz; mkdir -p $OUT
AH_SCREEN_VERIFY=1 timeout 400 python scripts/fuzz_gpu.py 200 98 > $OUT/fuzz98.log 2>&1; echo "fuzz 98 rc=$?"; tail -1 $OUT/fuzz98.log | cut -c1-200
timeout 400 python scripts/fuzz_gpu.py 200 99 > $OUT/fuzz99.log 2>&1; echo "fuzz 99 rc=$?"; tail -1 $OUT/fuzz99.log | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_small_calls.py -x -q -m gpu 2>&1 | tail -2
