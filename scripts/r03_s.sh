#!/bin/bash
cd /root/repo
OUT=gpurun_out/r03s
mkdir -p $OUT
rm -f gpurun_out/fuzz_fail.npz
timeout 300 python scripts/fuzz_gpu.py 100 5 > $OUT/fuzz_5.log 2>&1; echo "rc $?" >> $OUT/fuzz_5.log
timeout 300 python scripts/fuzz_gpu.py 100 99 > $OUT/fuzz_99.log 2>&1; echo "rc $?" >> $OUT/fuzz_99.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_index.py -m gpu -x -q -k "search or index or filter" > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
for keep in 0 0.5 0.1; do
    echo "keep $keep: $(timeout 300 python scripts/exp_search.py 6 64 $keep 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["queries_per_s"]), d["checksum"])')" >> $OUT/filter.txt
done
