#!/bin/bash
mkdir -p gpurun_out/r04d
AH_TIMING=1 timeout 900 python -m pytest tests/test_gpu_stream.py -x -q -m gpu > gpurun_out/r04d/stream_tests.log 2>&1
echo "stream tests rc=$?"; tail -30 gpurun_out/r04d/stream_tests.log
