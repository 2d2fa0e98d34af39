#!/bin/bash
# k_forest_screen_rows8: parity, then the per-level A/B on the 10M x 100-tree build.
OUT=gpurun_out/r06o; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_margin_modes.py -x -q -m gpu -k "int8_copies or several_tiles or every_margin_mode" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -5 $OUT/tests.log
for cfg in "AH_ROWS8=1" "AH_ROWS8=0"; do
  echo "== $cfg"
  env $cfg AH_TIMING=2 timeout 300 python scripts/exp_build.py 10000000 100 2 2>&1 | grep "level  [5-9]:\|level 1[0-4]:\|batch of\|seconds_total" | tail -12 | cut -c1-200
done
