#!/bin/bash
# kernel-level profile of the search leg: clustered / distinct queries, screen on / off
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04f
mkdir -p $OUT
for cfg in "on 64 1" "on 1000 1" "off 64 0" "off 1000 0"; do
  set -- $cfg
  AH_SEARCH_SCREEN=$3 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python scripts/exp_search.py 10 $2 > $OUT/search_$1_$2.log 2>&1
  echo "== screen $1, $2 base items: $(grep queries_per_s $OUT/search_$1_$2.log | cut -c1-200)" | tee $OUT/stats_$1_$2.txt
  python scripts/kstats.py $OUT/kt/kt_kernel_stats.csv k_descend k_leaf k_search_select k_flag k_queries k_visit k_prepare | tee -a $OUT/stats_$1_$2.txt
  rm -rf $OUT/kt
done
