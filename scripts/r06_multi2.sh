#!/bin/bash
# k_descend_multi: where the time goes (AH_SEARCH_MULTI_TRACE) and the trees-per-block sweep.
OUT=gpurun_out/r06i; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_small_calls.py -x -q -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
AH_SEARCH_MULTI_TRACE=1 timeout 300 python scripts/exp_latency.py 1 12 2>&1 | grep "multi\|last block" | tail -24
for tpb in 8 5 4 3 2; do
  echo "== trees per block $tpb: $(AH_SEARCH_MULTI_TREES_PER_BLOCK=$tpb timeout 300 python scripts/exp_latency.py 1 300 2>&1 | grep '^nq=' | tail -1)"
done
for tpb in 4 2; do
AH_SEARCH_MULTI_TREES_PER_BLOCK=$tpb AH_SEARCH_MULTI_TRACE=1 timeout 300 python scripts/exp_latency.py 1 6 2>&1 | grep "multi\|last block" | tail -12
done
