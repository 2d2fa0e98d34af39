#!/bin/bash
AH_TIMING=2 timeout 300 python scripts/exp_build.py 10000000 100 2 768 $1 2>&1 | grep "level\|tail:\|batch of\|^{" | tail -52 | cut -c1-190
