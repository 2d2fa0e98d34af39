#!/bin/bash
AH_SEARCH_MULTI_TRACE=1 timeout 300 python scripts/exp_latency.py 1 6 2>&1 | grep "tiles (latest" | tail -6
