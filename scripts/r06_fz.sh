#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_small_calls.py -x -q -m gpu 2>&1 | tail -2
timeout 600 python scripts/stress_one_query.py 30000 2>&1 | tail -1
for q in 1 8; do echo "== nq=$q: $(timeout 300 python scripts/exp_latency.py $q 500 2>&1 | grep '^nq=' | tail -1)"; done
AH_SEARCH_MULTI_TRACE=1 timeout 300 python scripts/exp_latency.py 1 4 2>&1 | grep "selection" | tail -2
