#!/bin/bash
# round 3, step l: full GPU suite + the search leg after the leaf-tile re-rank
cd /root/repo
OUT=gpurun_out/r03l
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
for t in 1 0; do
  AH_SEARCH_TILES=$t timeout 300 python scripts/exp_search.py 10 2>/dev/null | tail -1 > $OUT/search_tiles$t.json
done
timeout 600 python bench.py --no-build-10m --no-cpu --no-e2e > $OUT/bench.json 2> $OUT/bench.err
