#!/bin/bash
# round 5, call 8: block descent (nq small) parity + latency; fault + virtual tests
set -x
OUT=gpurun_out/r05h
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_faults.py -x -q -m gpu 2>&1 | tail -30
timeout 1500 python -m pytest tests/test_gpu_multi_device.py -x -q -m gpu 2>&1 | tail -30
timeout 2400 python -m pytest tests/test_gpu_index.py tests/test_gpu_search_scale.py tests/test_gpu_parity.py -x -q -m gpu -k "search or index or nns or reader" 2>&1 | tail -15
for nq in 1 8; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python scripts/exp_latency.py $nq 300 > $OUT/lat_$nq.log 2>&1
  tail -2 $OUT/lat_$nq.log
  python scripts/kstats.py $OUT/kt/kt_kernel_stats.csv k_descend k_leaf k_search k_flag k_queries k_visit k_prepare > $OUT/kstats_nq$nq.txt
  head -14 $OUT/kstats_nq$nq.txt
  rm -rf $OUT/kt
done
python scripts/exp_latency.py 1 300 2>&1 | tail -2
python scripts/exp_latency.py 64 200 2>&1 | tail -2
