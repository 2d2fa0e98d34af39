#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02b
mkdir -p $OUT
timeout 300 python scripts/exp_build.py 1000000 50 3 > $OUT/b1m.log 2>&1
AH_SCREEN=0 timeout 300 python scripts/exp_build.py 1000000 50 3 > $OUT/b1m_exact.log 2>&1
timeout 300 python scripts/exp_build.py 10000000 100 2 > $OUT/b10m.log 2>&1
AH_SCREEN=0 timeout 300 python scripts/exp_build.py 10000000 100 1 > $OUT/b10m_exact.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_margin_modes.py -x -q > $OUT/pytest_modes.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_margin_modes.py > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/*.log
