#!/bin/bash
# round 3, step k: variants of the leaf-tile kernel (temporary switches AH_X_VARIANT / AH_X_BLOCKS), HBM traffic
cd /root/repo
OUT=gpurun_out/r03k
mkdir -p $OUT
rm -f $OUT/variants.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "search" > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
AH_DEBUG=1 timeout 300 python scripts/exp_search.py 1 2>&1 | grep "search tiles" | head -1 > $OUT/stats.txt
for v in 0 1 2; do
  for b in 1024 4096; do
    echo "variant $v blocks $b: $(AH_X_VARIANT=$v AH_X_BLOCKS=$b timeout 300 python scripts/exp_search.py 8 2>/dev/null | tail -1 | cut -c1-80)" >> $OUT/variants.txt
  done
done
for c in FETCH_SIZE; do
  AH_X_VARIANT=0 timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- python scripts/exp_search.py 2 > $OUT/pmc_$c.log 2>&1
  f=$(find $OUT/pmc_$c -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python scripts/pmc_summary.py "$f" | grep "k_leaf_tiles\|k_descend\|k_search_select\|k_flag" > $OUT/pmc_$c.txt
  rm -rf $OUT/pmc_$c
done
AH_X_VARIANT=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python scripts/exp_search.py 5 > $OUT/kt.log 2>&1
f=$(find $OUT/kt -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && grep "k_leaf\|k_descend\|k_search\|k_flag\|k_visit" "$f" > $OUT/kernel_stats.csv
rm -rf $OUT/kt
