#!/bin/bash
# round 3, step o: randomised parity sweep with the new search paths
cd /root/repo
OUT=gpurun_out/r03o
mkdir -p $OUT
timeout 400 python scripts/fuzz_gpu.py 150 31 > $OUT/fuzz_31.log 2>&1; echo "rc $?" >> $OUT/fuzz_31.log
timeout 400 python scripts/fuzz_gpu.py 150 77 > $OUT/fuzz_77.log 2>&1; echo "rc $?" >> $OUT/fuzz_77.log
