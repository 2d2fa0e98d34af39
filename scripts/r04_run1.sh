#!/bin/bash
# round 4, GPU call 1: the new parity tests, the whole GPU suite, the default bench line
mkdir -p gpurun_out/r04a
cat /sys/kernel/mm/transparent_hugepage/enabled > gpurun_out/r04a/thp.txt 2>&1
nproc >> gpurun_out/r04a/thp.txt; free -g >> gpurun_out/r04a/thp.txt
timeout 900 python -m pytest tests/test_gpu_search_scale.py tests/test_gpu_screen_edges.py -x -q -m gpu > gpurun_out/r04a/new_tests.log 2>&1
echo "new tests rc=$?" | tee -a gpurun_out/r04a/new_tests.log
timeout 1200 python -m pytest tests -q -m gpu --deselect tests/test_gpu_search_scale.py --deselect tests/test_gpu_screen_edges.py > gpurun_out/r04a/suite.log 2>&1
echo "suite rc=$?" | tee -a gpurun_out/r04a/suite.log
AH_TIMING=1 timeout 900 python bench.py > gpurun_out/r04a/bench.json 2> gpurun_out/r04a/bench.err
echo "bench rc=$?"
tail -3 gpurun_out/r04a/new_tests.log; tail -3 gpurun_out/r04a/suite.log
