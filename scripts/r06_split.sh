#!/bin/bash
# k_forest_create_split under rocprofv3 --kernel-trace --stats (one 10M x 768 x 100-tree build after a warm one), after the parity tests of the splits
OUT=gpurun_out/r06p; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "split or forest or two_means" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -2 $OUT/tests.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python scripts/exp_build.py 10000000 100 2 > $OUT/build.log 2>&1
python scripts/kstats.py $OUT/kt/kt_kernel_stats.csv k_forest_create_split k_forest_screen_node k_forest_scatter
grep "^{" $OUT/build.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('wall %.4f device %.4f' % (d['wall'], d['seconds_device']))
"
rm -rf $OUT/kt
