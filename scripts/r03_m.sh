#!/bin/bash
# round 3, step m: the wave-per-query descent: parity, then the search leg with and without
cd /root/repo
OUT=gpurun_out/r03m
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_index.py tests/test_gpu_reference_snapshots.py -m gpu -x -q -k "search or index or snapshot" > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
for w in 1 0; do
  AH_SEARCH_WAVE=$w timeout 300 python scripts/exp_search.py 10 2>/dev/null | tail -1 > $OUT/search_wave$w.json
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python scripts/exp_search.py 5 > $OUT/kt.log 2>&1
f=$(find $OUT/kt -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && grep "k_leaf\|k_descend\|k_search\|k_flag\|k_visit" "$f" > $OUT/kernel_stats.csv
rm -rf $OUT/kt
