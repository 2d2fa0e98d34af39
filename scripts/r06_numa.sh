#!/bin/bash
# NUMA placement of the build's output path (AH_NUMA).  (1) the calling process confined to the CPUs of either host node; (2) the first build
# confined to one node — its blobs are first-touched there — and the later builds, on recycled blobs, free to run anywhere (scripts/exp_numa.py)
lscpu | grep "NUMA node[01] CPU"
for FAR in 0 1; do for NU in 0 1; do
echo "== first build on node $FAR, AH_NUMA=$NU"
AH_NUMA=$NU timeout 300 python scripts/exp_numa.py $FAR 2>&1 | grep "^{" | tr '\n' ' '; echo
done; done
