#!/bin/bash
# fuzz sweeps after the tail in groups of trees went in: its knobs drawn at random with every other switch
OUT=gpurun_out/r06z2; mkdir -p $OUT
for cfg in "240 211" "240 212 AH_SCREEN_VERIFY=1" "240 213"; do
  set -- $cfg
  env $3 timeout 500 python scripts/fuzz_gpu.py $1 $2 > $OUT/fuzz_$2.log 2>&1; echo "fuzz seed $2 ($3) rc=$?"; tail -1 $OUT/fuzz_$2.log | cut -c1-300
  grep -c "groups_run=[1-9]" $OUT/fuzz_$2.log
done
