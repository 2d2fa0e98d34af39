#!/bin/bash
# same-box A/B: two-phase read-back of the ids on / off
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03h
mkdir -p $OUT
for rep in 1 2; do
for e in 67108864 1099511627776; do
  AH_EARLY_IDS_MIN=$e AH_TIMING=1 timeout 300 python scripts/exp_build.py 10000000 100 3 > $OUT/t100_e${e}_${rep}.txt 2>&1
  echo "== early_min=$e"; grep -o "batch of.*\|\"wall\": [0-9.]*" $OUT/t100_e${e}_${rep}.txt | tail -4
done
done
timeout 600 python -m pytest tests/test_gpu_schedules.py -m gpu -x -q -k "two_phase or tunable" 2>&1 | tail -3
