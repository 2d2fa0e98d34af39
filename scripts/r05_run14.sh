#!/bin/bash
# round 5, call 14: the read-back worker's pinned double buffer — size and copy threads against the tail of a 10M x 100-tree build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05n
for mb in 64 128 256 512; do for th in 8 14; do
  echo "== AH_READBACK_MB=$mb AH_HOST_THREADS=$th"
  AH_READBACK_MB=$mb AH_HOST_THREADS=$th python scripts/exp_build.py 10000000 100 4 | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(round(d['wall'],4), round(d.get('device_seconds',0),4), round(d['wall']-d.get('device_seconds',0),4))
"
done; done 2>&1 | tee gpurun_out/r05n/readback.txt
