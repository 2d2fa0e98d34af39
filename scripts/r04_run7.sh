#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04g
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_index.py tests/test_gpu_search_scale.py -q -m gpu -k "search or index" > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -5 $OUT/tests.log
for cfg in "on 64 1" "on 1000 1"; do
  set -- $cfg
  AH_SEARCH_SCREEN=$3 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python scripts/exp_search.py 10 $2 > $OUT/search_$1_$2.log 2>&1
  echo "== screen $1, $2 base items: $(grep queries_per_s $OUT/search_$1_$2.log | cut -c1-120)" | tee $OUT/stats_$1_$2.txt
  python scripts/kstats.py $OUT/kt/kt_kernel_stats.csv k_descend_wave k_leaf_tiles k_search_select k_flag k_queries | tee -a $OUT/stats_$1_$2.txt
  rm -rf $OUT/kt
done
AH_EXP_SHAPE=10000000,768,100,cosine python scripts/exp_search.py 5 64 2>&1 | tail -1 | cut -c1-200
AH_EXP_SHAPE=10000000,768,100,cosine python scripts/exp_search.py 5 1000 2>&1 | tail -1 | cut -c1-200
AH_SEARCH_SCREEN=0 AH_EXP_SHAPE=10000000,768,100,cosine python scripts/exp_search.py 5 1000 2>&1 | tail -1 | cut -c1-200
