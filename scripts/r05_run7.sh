#!/bin/bash
# round 5, call 7: fault tests, virtual bench tests, single-query kernel chain
set -x
OUT=gpurun_out/r05g
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_faults.py -x -q -m gpu 2>&1 | tail -30
timeout 1500 python -m pytest tests/test_gpu_multi_device.py -x -q -m gpu 2>&1 | tail -30
for nq in 1 8; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python scripts/exp_latency.py $nq 300 > $OUT/lat_$nq.log 2>&1
  tail -2 $OUT/lat_$nq.log
  python scripts/kstats.py $OUT/kt/kt_kernel_stats.csv k_descend k_leaf k_search k_flag k_queries k_visit k_prepare k_unit k_tile k_ > $OUT/kstats_nq$nq.txt
  grep -v "k_forest\|k_shadow\|k_synth\|k_col\|k_dim\|k_next\|k_build_tiles\|k_stats" $OUT/kstats_nq$nq.txt | head -30
  rm -rf $OUT/kt
done
python scripts/exp_latency.py 1 300 2>&1 | tail -2
