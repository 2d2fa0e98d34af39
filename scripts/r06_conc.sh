#!/bin/bash
OUT=gpurun_out/r06v; mkdir -p $OUT
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_small_calls.py -x -q -m gpu -k "many_threads or several_compute" > $OUT/tests_$i.log 2>&1; echo "run $i rc=$?"; tail -2 $OUT/tests_$i.log; done
