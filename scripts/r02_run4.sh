#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02g
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_staging.py -x -q > $OUT/pytest_staging.log 2>&1
( time timeout 1500 python bench.py --steps 20 --warmup 5 ) > $OUT/bench.json 2> $OUT/bench.err
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu --no-build --no-extra --extra staging > $OUT/bench_staging.json 2> $OUT/bench_staging.err
AH_STAGE_REGISTER=1 timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu --no-build --no-extra --extra staging > $OUT/bench_staging_reg.json 2> $OUT/bench_staging_reg.err
tail -n 5 $OUT/pytest_staging.log; tail -n 5 $OUT/bench.err
