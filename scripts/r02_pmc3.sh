#!/bin/bash
# Vector-memory pipeline counters of the row-major margin kernels (TC = 16): screened vs f32, 10M x 768 x 32 trees.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02s
mkdir -p $OUT
run_pmc () { # name, env, counters...
  local name=$1; shift
  local envs=$1; shift
  env ${envs//,/ } timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python scripts/exp_build.py 10000000 32 > $OUT/$name.log 2>&1
  python scripts/pmc_summary.py $OUT/$name/*counter_collection.csv | grep -v "^[0-9]" | grep "rows" > $OUT/${name}_summary.txt 2>&1
  rm -rf $OUT/$name
}
for mode in "f32 AH_MARGIN_MODE=16,AH_SCREEN=0"; do
  set -- $mode
  run_pmc ta_$1 "$2" TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
  run_pmc ta2_$1 "$2" TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
  run_pmc tcp_$1 "$2" TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum
  run_pmc tcp2_$1 "$2" TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum
  run_pmc grbm_$1 "$2" GRBM_GUI_ACTIVE GRBM_COUNT
done
cat $OUT/*_summary.txt
