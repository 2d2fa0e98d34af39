#!/bin/bash
cd /root/repo
OUT=gpurun_out/r03u
mkdir -p $OUT
for rep in 1 2; do
for t in 1024 256 512; do
  lib=/root/repo/arroy_amd/libarroy_hip.so
  [ $t != 1024 ] && lib=/root/repo/arroy_amd/libarroy_hip_t$t.so
  AH_LIB_PATH=$lib timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu --no-build --no-search 2>/dev/null | tail -1 > $OUT/bench_${t}_$rep.json
done
done
