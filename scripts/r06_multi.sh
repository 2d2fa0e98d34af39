#!/bin/bash
# k_descend_multi (one query on several compute units): parity, then the latency A/B with the kernel times.
OUT=gpurun_out/r06h; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_small_calls.py tests/test_gpu_search_scale.py tests/test_gpu_index.py -x -q -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -12 $OUT/tests.log
for nq in 1 8; do for m in 1 0; do
  echo "== nq=$nq AH_SEARCH_MULTI=$m: $(AH_SEARCH_MULTI=$m timeout 300 python scripts/exp_latency.py $nq 300 2>&1 | grep '^nq=' | tail -1)"
done; done
for m in 1 0; do
  AH_SEARCH_MULTI=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$m -o kt -- python scripts/exp_latency.py 1 300 > $OUT/lat_$m.log 2>&1
  echo "## AH_SEARCH_MULTI=$m: $(grep '^nq=' $OUT/lat_$m.log | tail -1)"
  python scripts/kstats.py $OUT/kt_$m/kt_kernel_stats.csv k_descend k_leaf k_search_select k_units
  rm -rf $OUT/kt_$m
done
AH_SEARCH_MULTI=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_8 -o kt -- python scripts/exp_latency.py 8 300 > $OUT/lat_8.log 2>&1
echo "## nq=8 multi: $(grep '^nq=' $OUT/lat_8.log | tail -1)"; python scripts/kstats.py $OUT/kt_8/kt_kernel_stats.csv k_descend k_leaf k_search_select k_units; rm -rf $OUT/kt_8
