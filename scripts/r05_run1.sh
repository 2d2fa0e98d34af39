#!/bin/bash
# round 5, call 1: parity of the narrow dense kernel on small shapes, then the A/B at 10M x 768
set -x
mkdir -p gpurun_out/r05a
python -m pytest tests/test_gpu_margin_modes.py tests/test_gpu_schedules.py -x -q -m gpu -k "dense or coverage or launch_map or screen_bound" > gpurun_out/r05a/pytest.txt 2>&1
tail -5 gpurun_out/r05a/pytest.txt
timeout 900 python scripts/r05_dense.py 10000000 768 > gpurun_out/r05a/ab.jsonl 2> gpurun_out/r05a/ab_levels.txt
cat gpurun_out/r05a/ab.jsonl
grep -E "===|level  [0-6]:" gpurun_out/r05a/ab_levels.txt | head -150
