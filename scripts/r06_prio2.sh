#!/bin/bash
# the pipelined id upload of ah_rerank_batch on a stream of the highest priority (AH_READBACK_PRIORITY=1) or the default one: scripts/exp_rerank_batch.py, A/B/A/B/A/B
for P in 1 0 1 0 1 0; do echo "== AH_READBACK_PRIORITY=$P"; AH_READBACK_PRIORITY=$P timeout 300 python scripts/exp_rerank_batch.py 1 2>&1 | grep -i "q/s\|queries/s\|per s" | head -4 | cut -c1-300; done
