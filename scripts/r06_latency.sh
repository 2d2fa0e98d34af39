#!/bin/bash
# The one-query call (and a few queries a call): parity of the small submissions' switches, the latency A/B, the kernels, a fuzz sweep.
OUT=gpurun_out/r06lat; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_small_calls.py tests/test_gpu_search_scale.py tests/test_gpu_index.py -x -q -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
for cfg in "AH_SEARCH_MULTI=1" "AH_SEARCH_MULTI_OWN_UNITS=0" "AH_SEARCH_FLAT_TILES=0" "AH_SEARCH_STATUS_WIPE=0" "AH_SEARCH_SPIN_WAIT=0" "AH_SEARCH_MULTI=0"; do
  for q in 1 2 8; do echo "== nq=$q $cfg: $(env $cfg timeout 300 python scripts/exp_latency.py $q 400 2>&1 | grep '^nq=' | tail -1)"; done
done
for q in 1 8; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python scripts/exp_latency.py $q 300 > $OUT/lat.log 2>&1
echo "## nq=$q default: $(grep '^nq=' $OUT/lat.log | tail -1)"
python scripts/kstats.py $OUT/kt/kt_kernel_stats.csv k_descend k_leaf k_search_select k_units fillBuffer; rm -rf $OUT/kt
done
AH_SCREEN_VERIFY=1 timeout 300 python scripts/fuzz_gpu.py 120 98 > $OUT/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/fuzz.log | cut -c1-200
