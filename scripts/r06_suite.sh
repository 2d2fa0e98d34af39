#!/bin/bash
# The whole -m gpu suite + a fuzz sweep that draws the structured distributions.  Usage: gpurun --timeout 2400 -- bash scripts/r06_suite.sh
OUT=gpurun_out/r06b; mkdir -p $OUT
timeout 1500 python -m pytest tests -x -q -m gpu --durations=12 > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -22 $OUT/tests.log
for cfg in "150 81" "150 82 AH_SCREEN_VERIFY=1"; do
  set -- $cfg
  env $3 timeout 400 python scripts/fuzz_gpu.py $1 $2 > $OUT/fuzz_$2.log 2>&1; echo "fuzz seed $2 ($3) rc=$?"; tail -1 $OUT/fuzz_$2.log | cut -c1-300
done
