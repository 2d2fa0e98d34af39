#!/bin/bash
# round 3, step n: the search leg on query sets that share leaves less and less (1000 queries from 64 / 250 / 1000 items)
cd /root/repo
OUT=gpurun_out/r03n
mkdir -p $OUT
rm -f $OUT/sets.txt
for c in 64 250 1000; do
  for cfg in "1 1" "0 1" "0 0"; do
    set -- $cfg
    echo "bases $c tiles $1 wave $2: $(AH_SEARCH_TILES=$1 AH_SEARCH_WAVE=$2 timeout 300 python scripts/exp_search.py 6 $c 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["queries_per_s"]), d["checksum"])')" >> $OUT/sets.txt
  done
  AH_DEBUG=1 timeout 300 python scripts/exp_search.py 1 $c 2>&1 | grep "search tiles" | head -1 >> $OUT/sets.txt
done
