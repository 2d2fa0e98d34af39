#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02_variants
mkdir -p $OUT
run_kt () { # name, env...
  local name=$1; shift
  env "$@" timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/$name -o $name -- python scripts/exp_build.py 10000000 32 > $OUT/$name.log 2>&1
  python scripts/level_trace.py $OUT/$name/*kernel_trace.csv > $OUT/${name}_levels.txt 2>&1
  rm -rf $OUT/$name
}
for v in "" _c8w3 _c8w4 _c4w4 _c4w5; do
  for m in 16 8 4; do
    run_kt kt${v}_tc$m AH_LIB_PATH=$GRAFT_REPO_ROOT/arroy_amd/libarroy_hip$v.so AH_MARGIN_MODE=$m
  done
done
ls $OUT | head -50
