#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02l
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_margin_modes.py -x -q > $OUT/pytest_modes.log 2>&1
run_kt () { # name, env...
  local name=$1; shift
  env "$@" timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/$name -o $name -- python scripts/exp_build.py 10000000 100 > $OUT/$name.log 2>&1
  python scripts/level_trace.py $OUT/$name/*kernel_trace.csv > $OUT/${name}_levels.txt 2>&1
  rm -rf $OUT/$name
}
run_kt kt_auto AH_X=0
run_kt kt_tc16 AH_MARGIN_MODE=16
run_kt kt_tc8 AH_MARGIN_MODE=8
run_kt kt_lds8 AH_MARGIN_MODE=0x108
AH_TIMING=1 timeout 300 python scripts/exp_build.py 10000000 100 2 > $OUT/timing.log 2>&1
tail -n 2 $OUT/pytest_modes.log; grep "\[ah\]" $OUT/timing.log
