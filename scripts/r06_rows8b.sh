#!/bin/bash
for cfg in "AH_ROWS8=1 AH_DEBUG=77"; do
  echo "== $cfg (no fallback: timing only)"
  env $cfg AH_TIMING=2 timeout 300 python scripts/exp_build.py 10000000 100 1 2>&1 | grep "level  [6-9]:" | tail -4 | cut -c1-160
done
