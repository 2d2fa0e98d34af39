#!/bin/bash
# round 3, step i: the LDS-bitmap sort + dedup of the search candidates, A/B on one box
cd /root/repo
mkdir -p gpurun_out/r03i
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "search" > gpurun_out/r03i/pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03i/pytest.log
for b in 1 0 1 0; do
  AH_SEARCH_BITMAP=$b timeout 600 python bench.py --no-build-10m --no-cpu --no-e2e > gpurun_out/r03i/bench_bitmap$b.$RANDOM.json 2> gpurun_out/r03i/err.log
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_i -o s -- python /root/repo/bench.py --no-build-10m --no-cpu --no-e2e > /dev/null 2>&1
f=$(find /tmp/prof_i -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -30 "$f" > /root/repo/gpurun_out/r03i/kernel_stats.csv
