#!/bin/bash
# round 5, call 18: rank loop unrolled; latency; the re-rank call kernel by kernel
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05r
timeout 1500 python -m pytest tests/test_gpu_index.py tests/test_gpu_search_scale.py -q -m gpu -x 2>&1 | tail -3
python scripts/exp_latency.py 1 300 2>&1 | tail -2 | cut -c1-100
python scripts/exp_rerank_latency.py 300 2>&1 | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05r/prof -- python scripts/exp_rerank_latency.py 300 > /dev/null 2>&1
python scripts/kstats.py $(find gpurun_out/r05r/prof -name "*kernel_stats.csv" | head -1) 2>/dev/null | head -12
