#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for pipe in 1 2 3 0 1 2; do
  echo "== AH_DENSE_PIPE=$pipe, 100 trees"
  AH_DENSE_PIPE=$pipe AH_TIMING=2 python scripts/exp_build.py 10000000 100 2 2>&1 | grep -E "level  [3-5]:|batch of" | tail -4 | cut -c1-120
done
