#!/bin/bash
# The int8 first stage of the search's tile re-rank: parity (every search test, fuzz with the stage on small submissions too)
# and the A/B on the benchmark's search leg.
OUT=gpurun_out/r06f; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_search_scale.py tests/test_gpu_parity.py tests/test_gpu_small_calls.py tests/test_gpu_structured.py tests/test_gpu_index.py -x -q -m gpu -k "search or index or small or clustered or rerank" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -6 $OUT/tests.log
AH_SCREEN_VERIFY=1 timeout 300 python scripts/fuzz_gpu.py 120 92 > $OUT/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/fuzz.log | cut -c1-300
for mv in 2 8; do for c in 64 1000; do echo "== max visits $mv, $c base items: $(AH_SEARCH_SCREEN8_MAX_VISITS=$mv timeout 300 python scripts/exp_search.py 7 $c 2>&1 | tail -1 | cut -c150-260)"; done; done
for v in 1 0; do for c in 64 1000; do AH_SEARCH_SCREEN8=$v timeout 300 python scripts/exp_search.py 7 $c > $OUT/exp_search_${v}_$c.txt 2>&1; echo "== AH_SEARCH_SCREEN8=$v, $c distinct base items"; tail -1 $OUT/exp_search_${v}_$c.txt | cut -c1-400; done; done
AH_EXP_SHAPE=10000000,768,100,cosine AH_SEARCH_SCREEN8=1 timeout 300 python scripts/exp_search.py 5 1000 2>&1 | tail -1 | cut -c1-300
AH_EXP_SHAPE=10000000,768,100,cosine AH_SEARCH_SCREEN8=0 timeout 300 python scripts/exp_search.py 5 1000 2>&1 | tail -1 | cut -c1-300
