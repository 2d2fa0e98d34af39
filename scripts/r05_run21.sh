#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05t
python scripts/exp_latency.py 1000 50 2>&1 | tail -2 | cut -c1-100
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05t/prof -- python scripts/exp_latency.py 1000 50 > /dev/null 2>&1
python scripts/kstats.py $(find gpurun_out/r05t/prof -name "*kernel_stats.csv" | head -1) descend select tiles flag prepare h16 scan scatter 2>/dev/null | head -16
