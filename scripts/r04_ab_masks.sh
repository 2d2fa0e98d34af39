#!/bin/bash
# kernel stats of one 10M x 768 x 100-tree build with the side bits for the tile gather (AH_MASK_BITS, default) and the
# two-pairs-per-octet rounds of k_forest_exact_pairs
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04ab
mkdir -p $OUT
for bits in 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bits$bits -o b -- env AH_MASK_BITS=$bits python scripts/exp_build.py 10000000 100 1 > $OUT/prof_bits$bits.log 2>&1
  f=$(find $OUT/prof_bits$bits -name "*kernel_stats.csv" | head -1)
  python scripts/kstats.py $f | head -14
  cp $f $OUT/kernel_stats_bits$bits.csv
  rm -rf $OUT/prof_bits$bits
done
