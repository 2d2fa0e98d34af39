#!/bin/bash
# Where the time of the CLUSTERED 10M x 768 x 100-tree build goes (AH_TIMING=2: one line per level), next to the uniform
# build; then the N > 1 path at the REAL size on one GPU: 2 virtual devices x 10M rows, union digest vs the one-device build.
OUT=gpurun_out/r06d; mkdir -p $OUT
AH_TIMING=2 python scripts/exp_build.py 10000000 100 2 768 4 > $OUT/levels_clustered.txt 2>&1; echo "clustered rc=$?"
grep -c . $OUT/levels_clustered.txt
AH_TIMING=2 python scripts/exp_build.py 10000000 100 2 768 1 > $OUT/levels_uniform.txt 2>&1; echo "uniform rc=$?"
timeout 900 python bench.py --gpus 2 --virtual --build-items 10000000 --no-cpu --no-extra --steps 5 --warmup 1 > $OUT/bench_virtual_2x10m.json 2> $OUT/bench_virtual_2x10m.err; echo "virtual rc=$?"
tail -c 1500 $OUT/bench_virtual_2x10m.err
