#!/bin/bash
# GPU box: wall time + per-level trace of the 10M x 768 x 100-tree build for a list of environment settings.
#   SWEEP="NAME=VAL,NAME2=VAL2;NAME=VAL3;..."   (one build configuration per ';'-separated entry; '-' = defaults)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/sweep
mkdir -p $OUT
TREES=${TREES:-100}
IFS=";" read -ra CF <<< "${SWEEP:--}"
i=0
for cfg in "${CF[@]}"; do
  i=$((i+1))
  ( if [ "$cfg" != "-" ]; then IFS="," read -ra KV <<< "$cfg"; for kv in "${KV[@]}"; do export "$kv"; done; fi
    timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_$i -o kt -- python scripts/exp_build.py 10000000 $TREES 2 > $OUT/run_$i.log 2>&1
    python scripts/level_trace.py $OUT/kt_$i/kt_kernel_trace.csv > $OUT/levels_$i.txt 2>&1
    rm -rf $OUT/kt_$i )
  echo "== $i: $cfg"
  grep -o '"wall": [0-9.]*' $OUT/run_$i.log | tr '\n' ' '; echo
  grep -v "tc=None" $OUT/levels_$i.txt | awk 'NR>15' | cut -c1-75,100-200 | head -16
done
