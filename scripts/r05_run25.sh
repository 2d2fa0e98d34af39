#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python scripts/exp_latency.py 1 400 2>&1 | tail -2 | cut -c1-300
AH_EXPERIMENT_SKIP_STATUS_MEMSET=1 python scripts/exp_latency.py 1 400 2>&1 | tail -2 | cut -c1-300
python scripts/exp_latency.py 1 400 2>&1 | tail -2 | head -1
AH_EXPERIMENT_SKIP_STATUS_MEMSET=1 python scripts/exp_latency.py 1 400 2>&1 | tail -2 | head -1
