#!/bin/bash
# the cold child alone, with the library's own timing: where the first build's set-up goes
AH_TIMING=1 timeout 600 python bench.py --cold-child --build-items 10000000 2>&1 | grep -v "^\[ah\] level" | cut -c1-400 | tail -14
