#!/bin/bash
OUT=gpurun_out/r06c; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_multi_device.py tests/test_gpu_tail_groups.py -x -q -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
# eight concurrent 13-tree shares of 10M x 768 on one GPU (what an 8-GPU node's host sees), tails in groups
timeout 600 python scripts/exp_concurrent_share.py 2>&1 | tail -6
