#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r03x
timeout 300 python scripts/fuzz_gpu.py 170 2026 > gpurun_out/r03x/fuzz_2026.log 2>&1; echo "rc $?" >> gpurun_out/r03x/fuzz_2026.log
