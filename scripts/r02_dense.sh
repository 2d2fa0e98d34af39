#!/bin/bash
# GPU box: parity of the dense MFMA screen (every shape of test_gpu_margin_modes + the bound test), then per-level timing
# of the 10M x 768 build with the dense pass forced for every level up to AH_DENSE_MAX_COLS columns, for 100 and 13 trees.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/dense
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_margin_modes.py -x -q -k "dense_mfma or validated or bound_holds" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
IFS=";" read -ra CF <<< "${CFGS:-100 1 16384;100 0 16384;13 1 16384;13 0 16384}"
for cfg in "${CF[@]}"; do
  IFS=" " read -r a1 a2 a3 <<< "$cfg"; set -- $a1 $a2 $a3
  export AH_DENSE=$2 AH_DENSE_MAX_COLS=$3
  tag=t$1_dense$2
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$tag -o kt -- python scripts/exp_build.py 10000000 $1 2 > $OUT/$tag.log 2>&1
  python scripts/level_trace.py $OUT/kt_$tag/kt_kernel_trace.csv > $OUT/levels_$tag.txt 2>&1
  cp $OUT/kt_$tag/kt_kernel_stats.csv $OUT/stats_$tag.csv
  rm -rf $OUT/kt_$tag
  tail -2 $OUT/$tag.log | cut -c1-600
  grep -v "tc=None" $OUT/levels_$tag.txt | tail -14
done
