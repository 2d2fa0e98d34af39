#!/bin/bash
# round 5, call 15: the block descent with an arg-max queue + child records prefetched, k_units_small: search tests, latency, kernel list
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05o
timeout 1500 python -m pytest tests/test_gpu_index.py tests/test_gpu_search_scale.py tests/test_gpu_faults.py -q -m gpu -x 2>&1 | tail -5
python scripts/exp_latency.py 1 300 2>&1 | tail -2 | cut -c1-400
python scripts/exp_latency.py 8 300 2>&1 | tail -2 | cut -c1-120
python scripts/exp_latency.py 64 300 2>&1 | tail -2 | cut -c1-120
AH_SEARCH_SMALL_UNITS_MAX_QUERIES=0 python scripts/exp_latency.py 1 300 2>&1 | tail -2 | cut -c1-120
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05o/prof -- python scripts/exp_latency.py 1 300 > /dev/null 2>&1
python scripts/kstats.py $(find gpurun_out/r05o/prof -name "*kernel_stats.csv" | head -1) 2>/dev/null | head -16
