#!/bin/bash
cd /root/repo
OUT=gpurun_out/r03p
mkdir -p $OUT
rm -f $OUT/variants.txt
AH_X_VARIANT=2 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "search" > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
for v in 1 2 1 2; do
  echo "variant $v: $(AH_X_VARIANT=$v timeout 300 python scripts/exp_search.py 8 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["queries_per_s"]), d["checksum"])')" >> $OUT/variants.txt
done
for v in 0 1 2; do
  echo "bases 1000 variant $v: $(AH_X_VARIANT=$v timeout 300 python scripts/exp_search.py 8 1000 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["queries_per_s"]), d["checksum"])')" >> $OUT/variants.txt
done
