#!/bin/bash
# round 3, step q: candidate filters through the wave descent and the leaf tiles
cd /root/repo
OUT=gpurun_out/r03q
mkdir -p $OUT
rm -f $OUT/filter.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_index.py -m gpu -x -q -k "search or index or filter" > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
timeout 200 python scripts/fuzz_gpu.py 60 5 > $OUT/fuzz.log 2>&1; echo "rc $?" >> $OUT/fuzz.log
for keep in 0 0.5 0.1 0.02; do
  for cfg in "1 1" "0 0"; do
    set -- $cfg
    echo "keep $keep tiles $1 wave $2: $(AH_SEARCH_TILES=$1 AH_SEARCH_WAVE=$2 timeout 300 python scripts/exp_search.py 6 64 $keep 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["queries_per_s"]), d["checksum"])')" >> $OUT/filter.txt
  done
done
for keep in 0.5 0.1; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python scripts/exp_search.py 5 64 $keep > $OUT/kt_$keep.log 2>&1
  f=$(find $OUT/kt -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && grep "k_descend\|k_leaf\|k_search\|k_flag\|k_filter" "$f" | head -16 > $OUT/kernel_stats_$keep.csv
  rm -rf $OUT/kt
done
