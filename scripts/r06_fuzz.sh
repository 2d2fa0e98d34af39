#!/bin/bash
# Round 6's last fuzz sweeps: every small-submission switch drawn at random, the screens' self-check on in one of them.
OUT=gpurun_out/r06z; mkdir -p $OUT
for cfg in "240 101" "240 102 AH_SCREEN_VERIFY=1" "240 103"; do
  set -- $cfg
  env $3 timeout 400 python scripts/fuzz_gpu.py $1 $2 > $OUT/fuzz_$2.log 2>&1; echo "fuzz seed $2 ($3) rc=$?"; tail -1 $OUT/fuzz_$2.log | cut -c1-200
done
