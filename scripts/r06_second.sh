#!/bin/bash
# a second 10M x 768 dataset built while the first one is alive (bench.py's `normal` leg): the read-back worker's stream priority
for P in 0 1; do AH_READBACK_PRIORITY=$P timeout 300 python scripts/exp_second_dataset.py keep 2>&1 | tail -1 | sed "s/^/AH_READBACK_PRIORITY=$P /"; done
