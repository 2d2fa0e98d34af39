#!/bin/bash
# round 5, call 17: select with the keys in registers + rank ordering: search / rerank tests, latency, kernel list
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05q
timeout 1500 python -m pytest tests/test_gpu_index.py tests/test_gpu_search_scale.py tests/test_gpu_screen_edges.py tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -5
python scripts/exp_latency.py 1 300 2>&1 | tail -2 | cut -c1-100
python scripts/exp_latency.py 8 300 2>&1 | tail -2 | cut -c1-100
python scripts/exp_latency.py 64 300 2>&1 | tail -2 | cut -c1-100
AH_SEARCH_SELECT_CLUSTER=1 python scripts/exp_latency.py 1 300 2>&1 | tail -2 | cut -c1-100
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05q/prof -- python scripts/exp_latency.py 1 300 > /dev/null 2>&1
python scripts/kstats.py $(find gpurun_out/r05q/prof -name "*kernel_stats.csv" | head -1) descend select tiles16 units_small prepare h16 flag 2>/dev/null | head -16
