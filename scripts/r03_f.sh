#!/bin/bash
# GPU call: suite + A/B of the rows' second int8 digit (stage 1 of the node-major screen)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03f
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
for lo in 1 0; do
  for d in 1 2; do
    AH_SCREEN8_LO=$lo AH_TIMING=2 timeout 300 python scripts/exp_build.py 10000000 100 2 768 $d > $OUT/levels_t100_lo${lo}_d$d.txt 2>&1
    echo "== lo=$lo dist=$d"; grep -h "level 1[0-4]\|batch" $OUT/levels_t100_lo${lo}_d$d.txt | tail -6
  done
done
AH_TIMING=2 timeout 300 python scripts/exp_build.py 10000000 13 2 > $OUT/levels_t13.txt 2>&1; grep -h batch $OUT/levels_t13.txt | tail -1
python - <<'PY'
import json
for lo in (1,0):
    for d in (1,2):
        for l in open(f'gpurun_out/r03f/levels_t100_lo{lo}_d{d}.txt'):
            if l.startswith('{'): st=json.loads(l)
        print(lo,d,{k:st[k] for k in ('wall','seconds_device','screen8_pairs','screen8_decided','screen8b_decided','screen_fallbacks')})
PY
