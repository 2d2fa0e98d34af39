#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r03t
AH_LIB_PATH=/root/repo/arroy_amd/libarroy_hip_head.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "equal_keys_across_the_cut" > gpurun_out/r03t/pytest_old.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03t/pytest_old.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "equal_keys" > gpurun_out/r03t/pytest_new.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03t/pytest_new.log
