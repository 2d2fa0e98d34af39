#!/bin/bash
# GPU call: suite + per-level timings + kernel stats after the dense-epilogue / create_split changes
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03c
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
for t in 100 13; do
  AH_TIMING=2 timeout 300 python scripts/exp_build.py 10000000 $t 2 > $OUT/levels_t$t.txt 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt$t -o kt -- python scripts/exp_build.py 10000000 $t 2 > $OUT/kt$t.log 2>&1
  cp $OUT/kt$t/kt_kernel_stats.csv $OUT/kernel_stats_t$t.csv
  python scripts/level_trace.py $OUT/kt$t/kt_kernel_trace.csv > $OUT/level_trace_t$t.txt 2>&1
  rm -rf $OUT/kt$t
done
grep -h "level  [0-5]\|batch" $OUT/levels_t100.txt | tail -7; grep -h "batch" $OUT/levels_t13.txt | tail -1
