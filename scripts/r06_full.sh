#!/bin/bash
# Round 6: the whole -m gpu suite, then the round's evidence (scripts/collect_profiles_r06.sh).
# Usage: gpurun --timeout 3000 -- bash scripts/r06_full.sh
OUT=gpurun_out/r06final; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu --durations=12 > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -8 $OUT/tests.log
timeout 1500 bash scripts/collect_profiles_r06.sh > $OUT/collect.log 2>&1
echo "collect rc=$?"; tail -25 $OUT/collect.log
