#!/bin/bash
for m in 1 0; do for q in 1 8; do
echo "== 10M x 768 x 100 trees cosine, nq=$q AH_SEARCH_MULTI=$m: $(AH_EXP_SHAPE=10000000,768,100,cosine AH_SEARCH_MULTI=$m timeout 300 python scripts/exp_latency.py $q 300 2>&1 | grep '^nq=' | tail -1)"
done; done
echo "== 1M x 768 x 50 trees cosine nq=1: $(AH_EXP_SHAPE=1000000,768,50,cosine timeout 300 python scripts/exp_latency.py 1 300 2>&1 | grep '^nq=' | tail -1)"
echo "== same, AH_SEARCH_MULTI=0: $(AH_EXP_SHAPE=1000000,768,50,cosine AH_SEARCH_MULTI=0 timeout 300 python scripts/exp_latency.py 1 300 2>&1 | grep '^nq=' | tail -1)"
