#!/bin/bash
# round 5, call 5: ring depth x preload variants of the narrow kernel (13-tree share, kernel trace) + the fault-injection tests
set -x
OUT=gpurun_out/r05e
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "" _v1 _v2 _v3; do
  for dbg in 0 1; do
    AH_LIB_PATH=$GRAFT_REPO_ROOT/arroy_amd/libarroy_hip$v.so AH_DENSE_NARROW_MAX_COLS=128 AH_DENSE_NARROW_STREAM=0 AH_DENSE_DEBUG=$dbg rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python scripts/exp_build.py 10000000 13 2 > $OUT/v$v.log 2>&1
    python scripts/level_trace.py $OUT/kt/kt_kernel_trace.csv > $OUT/levels${v}_$dbg.txt 2>&1
    echo "== variant '$v' debug $dbg"; grep "tc=mfma" $OUT/levels${v}_$dbg.txt | tail -6 | cut -c1-60
    rm -rf $OUT/kt
  done
done
timeout 1200 python -m pytest tests/test_gpu_faults.py -x -q -m gpu 2>&1 | tail -25
