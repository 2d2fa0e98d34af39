#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03g
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
AH_TIMING=3 timeout 300 python scripts/exp_build.py 10000000 100 3 > $OUT/host_t100.txt 2>&1
grep -A1 "level 1[34]" $OUT/host_t100.txt | tail -8
grep batch $OUT/host_t100.txt
grep -o '"wall": [0-9.]*' $OUT/host_t100.txt
AH_TIMING=1 timeout 300 python scripts/exp_build.py 10000000 13 3 2>&1 | grep -E "batch|wall" | grep -o "batch.*\|\"wall\": [0-9.]*"
