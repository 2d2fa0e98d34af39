#!/bin/bash
# Run on the GPU box (gpurun -- bash scripts/collect_profiles.sh r03): the judged bench line, its rocprofv3 kernel summary,
# the PMC passes of the scan / re-rank / 1-bit scan kernels (FETCH_SIZE / WRITE_SIZE, each in its own run, no trace domain
# besides the kernel trace), the per-level breakdown of the 10M builds and the fabric counters of the node-major levels.
# Everything lands in gpurun_out/profiles/ and is copied into profiles/ by hand afterwards (scripts/pmc_kernels_json.py
# writes the traffic file bench.py reads, every entry stamped with its kernel-source hash).
set -u
R=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles
mkdir -p $OUT
python bench.py --steps 50 --warmup 5 2>$OUT/${R}_bench.err | tail -1 > $OUT/${R}_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python bench.py --steps 50 --warmup 5 --no-cpu --no-build-10m > $OUT/kt.log 2>&1
cp $OUT/kt/kt_kernel_stats.csv $OUT/${R}_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python bench.py --steps 5 --warmup 1 --no-cpu --no-build --no-search > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python bench.py --steps 5 --warmup 1 --no-cpu --no-build --no-search > $OUT/write.log 2>&1
python scripts/pmc_summary.py $OUT/fetch/fetch_counter_collection.csv | grep -v "^[0-9]" > $OUT/${R}_pmc_fetch_size.csv
python scripts/pmc_summary.py $OUT/write/write_counter_collection.csv | grep -v "^[0-9]" > $OUT/${R}_pmc_write_size.csv
python scripts/pmc_kernels_json.py $OUT/${R}_pmc_fetch_size.csv $OUT/${R}_pmc_write_size.csv > $OUT/${R}_pmc_kernels.json
# one line per level (second build of each run): 100 trees and the 13-tree share, uniform and ~N(0,1) rows
AH_TIMING=2 python scripts/exp_build.py 10000000 100 2 2>&1 | tail -18 > $OUT/${R}_levels_timing_100trees.txt
AH_TIMING=2 python scripts/exp_build.py 10000000 13 2 2>&1 | tail -18 > $OUT/${R}_levels_timing_13trees.txt
AH_TIMING=2 python scripts/exp_build.py 10000000 100 2 768 2 2>&1 | tail -18 > $OUT/${R}_levels_timing_100trees_normal.txt
# per-level kernel breakdown of the 10M x 768 x 100-tree build: default (screened) and f32 only
for mode in screened f32; do
  if [ $mode = f32 ]; then export AH_SCREEN=0; else unset AH_SCREEN; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$mode -o kt -- python scripts/exp_build.py 10000000 100 > $OUT/${R}_build10m_${mode}.log 2>&1
  python scripts/level_trace.py $OUT/kt_$mode/kt_kernel_trace.csv > $OUT/${R}_forest_levels_${mode}.txt 2>&1
  cp $OUT/kt_$mode/kt_kernel_stats.csv $OUT/${R}_build10m_${mode}_kernel_stats.csv
done
unset AH_SCREEN
# fabric / L2 counters of the node-major levels (two-digit int8 stage): bytes per pair actually moved
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  name=$(echo $set | tr ' ' '_')
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$name -o pmc -- python scripts/exp_build.py 10000000 100 > $OUT/pmc_$name.log 2>&1
  python scripts/pmc_node.py $OUT/pmc_$name/pmc_counter_collection.csv k_forest_screen_node > $OUT/${R}_node_major_${name}.txt 2>&1
  rm -rf $OUT/pmc_$name
done
rm -rf $OUT/kt $OUT/fetch $OUT/write $OUT/kt_screened $OUT/kt_f32
ls -la $OUT
