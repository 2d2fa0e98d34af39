#!/bin/bash
# round 5, call 27: the selection of a screened re-rank submission group by group under the next group's screen (AH_RERANK_SELECT_OVERLAP of
# that build; the switch and the second stream were removed after scripts/r05_run29.sh)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_screen_edges.py tests/test_gpu_faults.py -q -m gpu -x -k "rerank or screen or batch" 2>&1 | tail -3
python scripts/exp_rerank_batch.py 2>&1 | tail -4
