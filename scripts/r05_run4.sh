#!/bin/bash
# round 5, call 4: narrow dense kernel with preloaded epilogue operands and a deeper A ring
set -x
OUT=gpurun_out/r05d
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_margin_modes.py tests/test_gpu_schedules.py -x -q -m gpu -k "dense or coverage or launch_map" > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
R05_TREES=13 timeout 900 python scripts/r05_dense.py 10000000 768 old,narrow64,narrow64_nt,narrow128,narrow128_nt,narrow256_nostream > $OUT/ab.jsonl 2> $OUT/ab_levels.txt
cat $OUT/ab.jsonl
grep -E "===|level  [0-5]:" $OUT/ab_levels.txt | head -80
for cfg in "narrow128 AH_DENSE_NARROW_MAX_COLS=128 AH_DENSE_NARROW_STREAM=0" "noepi AH_DENSE_NARROW_MAX_COLS=128 AH_DENSE_NARROW_STREAM=0 AH_DENSE_DEBUG=1"; do
  set -- $cfg
  name=$1; shift
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$name -o kt -- python scripts/exp_build.py 10000000 13 2 > $OUT/$name.log 2>&1
  python scripts/level_trace.py $OUT/kt_$name/kt_kernel_trace.csv > $OUT/levels_$name.txt 2>&1
  echo "== $name"; grep "tc=mfma" $OUT/levels_$name.txt | tail -6
  rm -rf $OUT/kt_$name
done
