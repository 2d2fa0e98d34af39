#!/bin/bash
# GPU call: suite, A/B of the pipelined int8 stage, kernel-trace of the 100-tree and 13-tree builds
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03b
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
for pre in 1 0; do
  AH_NODE_PREFETCH=$pre AH_TIMING=2 timeout 300 python scripts/exp_build.py 10000000 100 2 > $OUT/levels_t100_pre$pre.txt 2>&1
  AH_NODE_PREFETCH=$pre AH_TIMING=2 timeout 300 python scripts/exp_build.py 10000000 13 2 > $OUT/levels_t13_pre$pre.txt 2>&1
done
for t in 100 13; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt$t -o kt -- python scripts/exp_build.py 10000000 $t 2 > $OUT/kt$t.log 2>&1
  cp $OUT/kt$t/kt_kernel_stats.csv $OUT/kernel_stats_t$t.csv
  python scripts/level_trace.py $OUT/kt$t/kt_kernel_trace.csv > $OUT/level_trace_t$t.txt 2>&1
  rm -rf $OUT/kt$t
done
grep -h "level 1[0-3]" $OUT/levels_t100_pre1.txt | tail -4; grep -h "level 1[0-3]" $OUT/levels_t100_pre0.txt | tail -4
