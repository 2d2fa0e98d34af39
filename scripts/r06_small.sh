#!/bin/bash
# The small-submission tests, the C shims and the remaining files after the first failure of r06_suite.sh.
OUT=gpurun_out/r06c; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_small_calls.py tests/test_gpu_staging.py tests/test_gpu_stream.py tests/test_gpu_structured.py -x -q -m gpu --durations=8 > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -22 $OUT/tests.log
