#!/bin/bash
# round 4, GPU call 3: (a) the default bench line again with the tail timers (where the second of run 1 went);
# (b) the counters of the row-major / dense levels of the 10M x 100-tree build (review item 3: counters first)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04c
mkdir -p $OUT
AH_TIMING=1 timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"
grep "batch of 100" $OUT/bench.err
AH_TIMING=2 python scripts/exp_build.py 10000000 100 2 2>&1 | tail -19 > $OUT/levels_timing_100trees.txt
for set in "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  name=$(echo $set | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$name -o pmc -- python scripts/exp_build.py 10000000 100 > $OUT/pmc_$name.log 2>&1
  python scripts/pmc_rows.py $OUT/pmc_$name/pmc_counter_collection.csv > $OUT/rows_${name}.txt 2>&1
  rm -rf $OUT/pmc_$name
done
ls -la $OUT
