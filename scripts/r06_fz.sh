#!/bin/bash
OUT=gpurun_out/r06nq64; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_small_calls.py tests/test_gpu_search_scale.py tests/test_gpu_index.py tests/test_gpu_structured.py -x -q -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
for q in 9 16 64; do
for cfg in "AH_SEARCH_SCREEN8_MIN_QUERIES=9" "AH_SEARCH_SCREEN8_MIN_QUERIES=65"; do
  echo "== nq=$q $cfg: $(env $cfg timeout 300 python scripts/exp_latency.py $q 300 2>&1 | grep '^nq=' | tail -1)"
done; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python scripts/exp_latency.py 64 300 > $OUT/lat.log 2>&1
echo "## nq=64 default: $(grep '^nq=' $OUT/lat.log | tail -1)"
python scripts/kstats.py $OUT/kt/kt_kernel_stats.csv k_descend k_leaf k_search_select k_units k_queries fillBuffer; rm -rf $OUT/kt
AH_SCREEN_VERIFY=1 timeout 300 python scripts/fuzz_gpu.py 150 104 > $OUT/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/fuzz.log | cut -c1-200
