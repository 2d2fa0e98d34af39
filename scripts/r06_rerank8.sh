#!/bin/bash
# The int8 first stage of the re-rank's certified top-k screen: the A/B against binary16 / f32 on uniform (1), N(0,1) (2) and
# clustered (4) rows, answers compared bit for bit; kernel times of the uniform run.
OUT=gpurun_out/r06e; mkdir -p $OUT
for d in 1 2 4; do timeout 300 python scripts/exp_rerank_batch.py $d > $OUT/exp_dist$d.txt 2>&1; echo "== distribution $d"; cut -c1-420 $OUT/exp_dist$d.txt; done
