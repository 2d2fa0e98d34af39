#!/bin/bash
# Round 2, run 1: counters for the forest margin kernels (10M x 768 x 100 trees) + forced row-major group sizes on the deep levels.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02a
mkdir -p $OUT
rocprofv3 -L > $OUT/counters_all.txt 2>&1
grep -o -E "\b(TCC|TCP|SQ|TA|TD|GRBM)_[A-Z0-9_]+(_sum)?\b" $OUT/counters_all.txt | sort -u > $OUT/counters.txt
run_pmc () { # name, counters...
  local name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python scripts/exp_build.py 10000000 100 > $OUT/$name.log 2>&1
  python scripts/pmc_summary.py $OUT/$name/*counter_collection.csv > $OUT/${name}_summary.txt 2>&1
  rm -rf $OUT/$name
}
run_pmc tcc TCC_HIT_sum TCC_MISS_sum
run_pmc tcc2 TCC_EA0_RDREQ_sum TCC_REQ_sum
run_pmc tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
run_pmc sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD
run_pmc sq2 SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS
run_kt () { # name, env...
  local name=$1; shift
  env "$@" timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/$name -o $name -- python scripts/exp_build.py 10000000 100 > $OUT/$name.log 2>&1
  python scripts/level_trace.py $OUT/$name/*kernel_trace.csv > $OUT/${name}_levels.txt 2>&1
  rm -rf $OUT/$name
}
run_kt kt_default AH_X=0
run_kt kt_tc2 AH_ROWMAJOR=1 AH_ROWMAJOR_CACHE_MB=100000 AH_ROWMAJOR_MAX_TC=2
run_kt kt_tc4 AH_ROWMAJOR=1 AH_ROWMAJOR_CACHE_MB=100000 AH_ROWMAJOR_MAX_TC=4
run_kt kt_tc8 AH_ROWMAJOR=1 AH_ROWMAJOR_CACHE_MB=100000 AH_ROWMAJOR_MAX_TC=8
run_kt kt_tc16 AH_ROWMAJOR=1 AH_ROWMAJOR_CACHE_MB=100000 AH_ROWMAJOR_MAX_TC=16
ls -la $OUT
