#!/bin/bash
cd /root/repo
OUT=gpurun_out/r03r
mkdir -p $OUT
for keep in 0.5 0.1; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python scripts/exp_search.py 5 64 $keep > $OUT/kt_$keep.log 2>&1
  f=$(find $OUT/kt -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && grep -v "k_forest\|k_dense\|k_rows\|k_shadow\|k_col\|k_dim\|k_next\|k_build\|k_decode" "$f" | head -16 > $OUT/kernel_stats_$keep.csv
  rm -rf $OUT/kt
  AH_DEBUG=1 timeout 300 python scripts/exp_search.py 1 64 $keep 2>&1 | grep "search tiles" | head -1 > $OUT/stats_$keep.txt
done
