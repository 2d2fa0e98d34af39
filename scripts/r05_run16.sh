#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
AH_SEARCH_SELECT_CLUSTER=1 python scripts/exp_latency.py 1 30 2>&1 | grep -E "DBGSEL" | tail -5
