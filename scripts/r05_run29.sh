#!/bin/bash
# round 5, call 29: AH_RERANK_SELECT_OVERLAP inside bench.py's own rerank leg (alternating, same box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for ov in 1 0 1 0; do
  AH_RERANK_SELECT_OVERLAP=$ov python bench.py --steps 5 --warmup 1 --no-cpu --no-build --no-search --no-live-pmc 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['rerank']
print('overlap=$ov', {k: round(v['queries_per_s']) for k, v in r.items() if isinstance(v, dict) and 'queries_per_s' in v})"
done
