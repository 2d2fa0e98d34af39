#!/bin/bash
OUT=gpurun_out/r06y; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_search_scale.py -x -q -m gpu > $OUT/tests.log 2>&1; echo "rc=$?"; tail -12 $OUT/tests.log
