#!/bin/bash
OUT=gpurun_out/r06t; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_small_calls.py tests/test_gpu_search_scale.py tests/test_gpu_index.py tests/test_gpu_parity.py -x -q -m gpu -k "search or small or index or one_query" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
for q in 1 8; do echo "== nq=$q: $(timeout 300 python scripts/exp_latency.py $q 500 2>&1 | grep '^nq=' | tail -1)"; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python scripts/exp_latency.py 1 300 > $OUT/lat.log 2>&1
python scripts/kstats.py $OUT/kt/kt_kernel_stats.csv k_descend k_leaf k_search_select; rm -rf $OUT/kt
AH_SCREEN_VERIFY=1 timeout 300 python scripts/fuzz_gpu.py 100 97 > $OUT/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/fuzz.log | cut -c1-200
