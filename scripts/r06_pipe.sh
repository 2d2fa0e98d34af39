#!/bin/bash
OUT=gpurun_out/r06w; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_search_scale.py tests/test_gpu_parity.py tests/test_gpu_index.py -x -q -m gpu -k "search or index" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
for p in 1 2 3 4; do for c in 64 1000; do
  echo "== AH_SEARCH_PIPELINE=$p, $c base items: $(AH_SEARCH_PIPELINE=$p timeout 300 python scripts/exp_search.py 7 $c 2>&1 | tail -1 | grep -o '"queries_per_s": [0-9.]*')"
done; done
for p in 1 2 4; do
echo "== 10M x 768 x 100 trees, AH_SEARCH_PIPELINE=$p: $(AH_EXP_SHAPE=10000000,768,100,cosine AH_SEARCH_PIPELINE=$p timeout 300 python scripts/exp_search.py 5 1000 2>&1 | tail -1 | grep -o '"queries_per_s": [0-9.]*')"
done
