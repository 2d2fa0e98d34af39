#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_small_calls.py tests/test_gpu_index.py tests/test_gpu_search_scale.py -q -m gpu -x 2>&1 | tail -3
python scripts/exp_latency.py 1 300 2>&1 | tail -2 | cut -c1-100
AH_SEARCH_SINGLE_FUSED=0 python scripts/exp_latency.py 1 300 2>&1 | tail -2 | head -1
