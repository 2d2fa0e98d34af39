#!/bin/bash
# First GPU call of round 3: the whole -m gpu suite, per-level timings of the 10M builds (uniform / ~N(0,1) / outlier
# dimensions; 100 trees and the 13-tree share), the bench line.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03a
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
for spec in "100 1" "13 1" "100 2" "100 3"; do
  set -- $spec
  AH_TIMING=2 timeout 300 python scripts/exp_build.py 10000000 $1 2 768 $2 > $OUT/levels_t$1_d$2.txt 2>&1
done
timeout 900 python bench.py --steps 20 --warmup 5 2>$OUT/bench.err | tail -1 > $OUT/bench.json
tail -c 600 $OUT/bench.json
