#!/bin/bash
cd /root/repo
OUT=gpurun_out/r03v
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "search or rerank or batch" > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
timeout 300 python scripts/exp_search.py 8 2>/dev/null | tail -1 | cut -c1-300 > $OUT/search.json
