#!/bin/bash
OUT=gpurun_out/r06m32; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_small_calls.py tests/test_gpu_search_scale.py tests/test_gpu_index.py -x -q -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
for q in 1 8 16; do echo "== nq=$q: $(timeout 300 python scripts/exp_latency.py $q 400 2>&1 | grep '^nq=' | tail -1)"; done
echo "== 10M x 768 x 100 trees nq=1: $(AH_EXP_SHAPE=10000000,768,100,cosine timeout 300 python scripts/exp_latency.py 1 300 2>&1 | grep '^nq=' | tail -1)"
AH_SCREEN_VERIFY=1 timeout 300 python scripts/fuzz_gpu.py 150 106 > $OUT/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/fuzz.log | cut -c1-200
timeout 600 python scripts/stress_one_query.py 20000 2>&1 | tail -1
