#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02e
mkdir -p $OUT
run_pmc () { # name, mode, counters...
  local name=$1; shift
  local mode=$1; shift
  AH_MARGIN_MODE=$mode timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python scripts/exp_build.py 10000000 32 > $OUT/$name.log 2>&1
  python scripts/pmc_summary.py $OUT/$name/*counter_collection.csv | grep -v "^[0-9]" | grep "screen\|margin" > $OUT/${name}_summary.txt 2>&1
  rm -rf $OUT/$name
}
for m in 16 8 0x108; do
run_pmc sq_$m $m SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD
run_pmc sq2_$m $m SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM
run_pmc tcc_$m $m TCC_HIT_sum TCC_MISS_sum
run_pmc tcp_$m $m TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
done
cat $OUT/*_summary.txt
