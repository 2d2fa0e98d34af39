#!/bin/bash
# round 5, call 9: the whole GPU suite + single-query latency after the barrier cut
set -x
OUT=gpurun_out/r05i
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python scripts/exp_latency.py 1 300 2>&1 | tail -2
python scripts/exp_latency.py 8 300 2>&1 | tail -2
timeout 3400 python -m pytest tests -q -m gpu --maxfail=10 > $OUT/pytest_all.txt 2>&1
tail -15 $OUT/pytest_all.txt
