#!/bin/bash
# the whole -m gpu suite only
OUT=gpurun_out/r06suite; mkdir -p $OUT
timeout 1700 python -m pytest tests -q -m gpu --durations=12 > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -25 $OUT/tests.log
