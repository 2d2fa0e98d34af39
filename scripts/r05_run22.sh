#!/bin/bash
# round 5, call 22: k-steps in flight of the big submissions' binary16 leaf tiles (AH_TILES16_KF = 1 / 2 / 4), nq = 1000 and 250
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in libarroy_hip.so libarroy_hip_kf2.so libarroy_hip_kf4.so; do
  echo "== $lib"
  AH_LIB_PATH=$GRAFT_REPO_ROOT/arroy_amd/$lib python scripts/exp_latency.py 1000 40 2>&1 | tail -2 | head -1
  AH_LIB_PATH=$GRAFT_REPO_ROOT/arroy_amd/$lib python scripts/exp_latency.py 250 60 2>&1 | tail -2 | head -1
done
