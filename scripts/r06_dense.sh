#!/bin/bash
# Dense levels: the tail launch (AH_DENSE_TAIL) and the exact pairs with whole rows in flight (AH_EXACT_WIDE): parity, then A/B.
OUT=gpurun_out/r06n; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_gpu_dense.py tests/test_gpu_margin_modes.py tests/test_gpu_schedules.py tests/test_gpu_structured.py -x -q -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -4 $OUT/tests.log
for cfg in "AH_DENSE_TAIL=1 AH_EXACT_WIDE=1" "AH_DENSE_TAIL=0 AH_EXACT_WIDE=1" "AH_DENSE_TAIL=1 AH_EXACT_WIDE=0" "AH_DENSE_TAIL=0 AH_EXACT_WIDE=0"; do
  echo "== $cfg"
  env $cfg AH_TIMING=2 timeout 300 python scripts/exp_build.py 10000000 100 2 2>&1 | grep "level  [0-5]:\|batch of\|seconds_total" | tail -8 | cut -c1-150
done
for cfg in "AH_DENSE_TAIL=1 AH_EXACT_WIDE=1" "AH_DENSE_TAIL=0 AH_EXACT_WIDE=0"; do
  env $cfg timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python scripts/exp_build.py 10000000 100 > $OUT/b.log 2>&1
  echo "## $cfg"; python scripts/kstats.py $OUT/kt/kt_kernel_stats.csv k_forest_dense k_forest_exact; rm -rf $OUT/kt
done
AH_SCREEN_VERIFY=1 timeout 300 python scripts/fuzz_gpu.py 100 95 > $OUT/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/fuzz.log | cut -c1-200
