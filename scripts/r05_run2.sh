#!/bin/bash
# round 5, call 2: narrow dense kernel on 16x16x32 tiles — parity, A/B, kernel trace of the 13-tree share
set -x
OUT=gpurun_out/r05b
mkdir -p $OUT
python -m pytest tests/test_gpu_margin_modes.py tests/test_gpu_schedules.py -x -q -m gpu -k "dense or coverage or launch_map" > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
timeout 900 python scripts/r05_dense.py 10000000 768 old,narrow256,narrow256_nostream,narrow512 > $OUT/ab.jsonl 2> $OUT/ab_levels.txt
cat $OUT/ab.jsonl
grep -E "===|level  [0-5]:" $OUT/ab_levels.txt | head -80
cd /tmp && export TMPDIR=/tmp
for v in 0 -1; do
  AH_DENSE_NARROW=$v rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o t -- python $GRAFT_REPO_ROOT/scripts/exp_build.py 10000000 13 2 > /dev/null 2>&1
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  cp $f $GRAFT_REPO_ROOT/$OUT/kernel_stats_13trees_narrow$v.csv
  python $GRAFT_REPO_ROOT/scripts/kstats.py $f 2>/dev/null | head -14
done
