#!/bin/bash
OUT=gpurun_out/r06k; mkdir -p $OUT
AH_SEARCH_MULTI_TRACE=1 timeout 300 python scripts/exp_latency.py 1 6 2>&1 | grep "selection\|multi block" | tail -16
