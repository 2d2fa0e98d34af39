#!/bin/bash
# round 3, step j: the leaf-tile re-rank of the search leg: parity, then kernel times with and without
cd /root/repo
OUT=gpurun_out/r03j
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_index.py tests/test_gpu_reference_snapshots.py -m gpu -x -q -k "search or index or snapshot" > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
for b in 1 0; do
  AH_SEARCH_TILES=$b timeout 600 python scripts/exp_search.py 10 > $OUT/plain_$b.log 2>&1
  AH_SEARCH_TILES=$b timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$b -o kt -- python scripts/exp_search.py 5 > $OUT/search_$b.log 2>&1
  f=$(find $OUT/kt_$b -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && grep -v "k_forest\|k_dense\|k_rows\|k_shadow\|k_col\|k_dim" "$f" | head -25 > $OUT/kernel_stats_$b.csv
  rm -rf $OUT/kt_$b
done
