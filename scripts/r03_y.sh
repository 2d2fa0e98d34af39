#!/bin/bash
# round 3: the leaf tiles with the hash-set dedup (id spaces beyond the LDS bitmap): parity, then 10M x 768 cosine x 100 trees
cd /root/repo
OUT=gpurun_out/r03y
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "search" > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
for t in 1 0; do
  echo "10M tiles $t: $(AH_EXP_SHAPE=10000000,768,100,cosine AH_SEARCH_TILES=$t timeout 300 python scripts/exp_search.py 5 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["queries_per_s"]), d["checksum"], [round(x*1e3,2) for x in d["seconds"]])')" >> $OUT/big.txt
done
