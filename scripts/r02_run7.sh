#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02k
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_staging.py -x -q > $OUT/pytest_staging.log 2>&1
timeout 400 python scripts/fuzz_gpu.py 150 11 > $OUT/fuzz11.log 2>&1
AH_SCREEN_VERIFY=1 timeout 400 python scripts/fuzz_gpu.py 120 12 > $OUT/fuzz12_verify.log 2>&1
tail -n 3 $OUT/pytest_staging.log $OUT/fuzz11.log $OUT/fuzz12_verify.log
