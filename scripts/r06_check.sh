#!/bin/bash
# Round 6, first device run: the new parity tests (structured data, whole trees vs the oracle, the no-peer-access replica, the
# small-submission gate), then the whole bench line.  Usage: gpurun --timeout 2400 -- bash scripts/r06_check.sh
OUT=gpurun_out/r06a; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_structured.py tests/test_gpu_small_calls.py tests/test_gpu_multi_device.py -x -q -m gpu --durations=15 > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -25 $OUT/tests.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
tail -c 2500 $OUT/bench.err
