#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
AH_DEBUG=1 python scripts/exp_latency.py 1 30 2>&1 | grep -E "DBGTICK" | tail -5
