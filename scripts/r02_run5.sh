#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02h
mkdir -p $OUT
nproc > $OUT/host.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/host.txt 2>&1; uptime >> $OUT/host.txt; lscpu | head -20 >> $OUT/host.txt
timeout 1500 python -m pytest tests/test_gpu_staging.py -x -q > $OUT/pytest_staging.log 2>&1
( time timeout 1500 python bench.py --steps 20 --warmup 5 --no-build-10m ) > $OUT/bench.json 2> $OUT/bench.err
for t in 4 8 12 24; do AH_STAGE_THREADS=$t timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu --no-build --no-extra --extra staging 2>/dev/null | python -c "import sys,json; print('threads $t', json.loads(sys.stdin.read())['extra'])" >> $OUT/staging.txt; done
tail -n 5 $OUT/pytest_staging.log; tail -n 5 $OUT/bench.err; cat $OUT/staging.txt $OUT/host.txt
