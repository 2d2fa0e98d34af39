#!/bin/bash
# round 4, GPU call 2: fixed scale tests; where the warm 10M builds of the staged dataset lose a second (AH_TIMING=1 setup lines)
mkdir -p gpurun_out/r04b
timeout 900 python -m pytest tests/test_gpu_search_scale.py tests/test_gpu_screen_edges.py -q -m gpu > gpurun_out/r04b/new_tests.log 2>&1
echo "new tests rc=$?" | tee -a gpurun_out/r04b/new_tests.log
AH_TIMING=1 timeout 900 python bench.py --no-extra > gpurun_out/r04b/bench.json 2> gpurun_out/r04b/bench.err
echo "bench rc=$?"
AH_TIMING=1 timeout 600 python bench.py --no-extra --no-cpu > gpurun_out/r04b/bench_nocpu.json 2> gpurun_out/r04b/bench_nocpu.err
echo "bench nocpu rc=$?"
tail -5 gpurun_out/r04b/new_tests.log
grep "batch of 100\|batch setup" gpurun_out/r04b/bench.err | tail -30
