#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04m
mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -4 $OUT/tests.log | cut -c1-300
timeout 300 python scripts/fuzz_gpu.py 100 51 > $OUT/fuzz_51.log 2>&1; echo "fuzz 51 rc=$?"; tail -2 $OUT/fuzz_51.log | cut -c1-600
AH_SCREEN_VERIFY=1 timeout 300 python scripts/fuzz_gpu.py 80 52 > $OUT/fuzz_52.log 2>&1; echo "fuzz 52 rc=$?"; tail -2 $OUT/fuzz_52.log | cut -c1-600
