#!/bin/bash
# GPU call: suite (+ the 128x128 wave-tile dense variant under the tunable), A/B of that variant on the 10M builds
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03e
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
AH_DENSE_WAVE128=1 timeout 600 python -m pytest tests/test_gpu_margin_modes.py -m gpu -x -q -k "dense_mfma or bound_holds" > $OUT/pytest_wave128.log 2>&1; echo "rc $?" >> $OUT/pytest_wave128.log
tail -2 $OUT/pytest_wave128.log
for w in 1 0; do
  AH_DENSE_WAVE128=$w AH_TIMING=2 timeout 300 python scripts/exp_build.py 10000000 100 2 > $OUT/levels_t100_w$w.txt 2>&1
  grep -h "level  [0-5]" $OUT/levels_t100_w$w.txt | tail -6
done
AH_DENSE_WAVE128=1 AH_DENSE=1 AH_TIMING=2 timeout 300 python scripts/exp_build.py 10000000 100 1 > $OUT/levels_t100_w1_forced.txt 2>&1
AH_DENSE_WAVE128=0 AH_DENSE=1 AH_TIMING=2 timeout 300 python scripts/exp_build.py 10000000 100 1 > $OUT/levels_t100_w0_forced.txt 2>&1
grep -h "level  [5-8]" $OUT/levels_t100_w1_forced.txt $OUT/levels_t100_w0_forced.txt
