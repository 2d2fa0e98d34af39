#!/bin/bash
OUT=gpurun_out/r06x; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_small_calls.py -x -q -m gpu > $OUT/tests.log 2>&1; echo "rc=$?"; tail -12 $OUT/tests.log
