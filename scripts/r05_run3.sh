#!/bin/bash
# round 5, call 3: where does the narrow dense kernel's time go?  kernel trace + k-loop / epilogue split (13-tree share)
set -x
OUT=gpurun_out/r05c
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cfg in "old AH_DENSE_NARROW=0" "narrow AH_DENSE_NARROW_STREAM=0" "narrow_nt AH_DENSE_NARROW_STREAM=1" "noepi AH_DENSE_NARROW_STREAM=0 AH_DENSE_DEBUG=1" "nokloop AH_DENSE_NARROW_STREAM=0 AH_DENSE_DEBUG=2"; do
  set -- $cfg
  name=$1; shift
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$name -o kt -- python scripts/exp_build.py 10000000 13 2 > $OUT/$name.log 2>&1
  python scripts/level_trace.py $OUT/kt_$name/kt_kernel_trace.csv > $OUT/levels_$name.txt 2>&1
  echo "== $name"; tail -22 $OUT/levels_$name.txt | head -8
  python scripts/kstats.py $OUT/kt_$name/kt_kernel_stats.csv k_forest_dense k_forest_exact k_forest_advance k_forest_masks k_forest_pack > $OUT/kstats_$name.txt
  cat $OUT/kstats_$name.txt
  rm -rf $OUT/kt_$name
done
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "smoke or forest" 2>&1 | tail -3
