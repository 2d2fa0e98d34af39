#!/bin/bash
# round 5, call 11: wide margin loads in the descent: parity (search tests) + latency + kernel times
set -x
OUT=gpurun_out/r05k
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 2400 python -m pytest tests/test_gpu_index.py tests/test_gpu_search_scale.py tests/test_gpu_parity.py -q -m gpu -k "search or index or nns or reader or route" 2>&1 | tail -5
for nq in 1 1000; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python scripts/exp_latency.py $nq 100 > $OUT/lat_$nq.log 2>&1
  tail -2 $OUT/lat_$nq.log | cut -c1-200
  python scripts/kstats.py $OUT/kt/kt_kernel_stats.csv k_descend k_leaf_tiles k_search_select | head -6
  rm -rf $OUT/kt
done
python scripts/exp_latency.py 1 300 2>&1 | tail -2 | cut -c1-150
python scripts/exp_latency.py 8 300 2>&1 | tail -2 | cut -c1-150
python scripts/exp_latency.py 64 300 2>&1 | tail -2 | cut -c1-150
