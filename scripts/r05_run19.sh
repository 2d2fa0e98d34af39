#!/bin/bash
# round 5, call 19: one-launch selection of the single-query re-rank: parity tests, latency, kernel list
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05s
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_index.py tests/test_gpu_screen_edges.py tests/test_gpu_faults.py tests/test_gpu_reference_snapshots.py -q -m gpu -x 2>&1 | tail -3
python scripts/exp_rerank_latency.py 300 2>&1 | tail -1
AH_RERANK_SMALL=0 python scripts/exp_rerank_latency.py 300 2>&1 | tail -1
python scripts/exp_latency.py 1 300 2>&1 | tail -2 | cut -c1-100
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05s/prof -- python scripts/exp_rerank_latency.py 300 > /dev/null 2>&1
python scripts/kstats.py $(find gpurun_out/r05s/prof -name "*kernel_stats.csv" | head -1) 2>/dev/null | head -8
