#!/bin/bash
# after a kernel change late in the round: the hash-stamped PMC traffic of the scan / re-rank / 1-bit kernels, the scan's
# kernel summary, and the per-level evidence of the 10M builds (parts 1-3 of scripts/collect_profiles_r04.sh)
set -u
R=r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles
mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python bench.py --steps 5 --warmup 1 --no-cpu --no-build --no-search --no-live-pmc > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python bench.py --steps 5 --warmup 1 --no-cpu --no-build --no-search --no-live-pmc > $OUT/write.log 2>&1
python scripts/pmc_summary.py $OUT/fetch/fetch_counter_collection.csv | grep -v "^[0-9]" > $OUT/${R}_pmc_fetch_size.csv
python scripts/pmc_summary.py $OUT/write/write_counter_collection.csv | grep -v "^[0-9]" > $OUT/${R}_pmc_write_size.csv
python scripts/pmc_kernels_json.py $OUT/${R}_pmc_fetch_size.csv $OUT/${R}_pmc_write_size.csv > $OUT/${R}_pmc_kernels.json
rm -rf $OUT/fetch $OUT/write
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python bench.py --steps 50 --warmup 5 --no-cpu --no-build-10m --no-live-pmc > $OUT/kt.log 2>&1
cp $OUT/kt/kt_kernel_stats.csv $OUT/${R}_kernel_stats.csv
rm -rf $OUT/kt
AH_TIMING=2 python scripts/exp_build.py 10000000 100 2 2>&1 | tail -19 > $OUT/${R}_levels_timing_100trees.txt
AH_TIMING=2 python scripts/exp_build.py 10000000 13 2 2>&1 | tail -19 > $OUT/${R}_levels_timing_13trees.txt
AH_TIMING=2 python scripts/exp_build.py 10000000 100 2 768 2 2>&1 | tail -19 > $OUT/${R}_levels_timing_100trees_normal.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_b -o kt -- python scripts/exp_build.py 10000000 100 > $OUT/${R}_build10m_screened.log 2>&1
python scripts/level_trace.py $OUT/kt_b/kt_kernel_trace.csv > $OUT/${R}_forest_levels_screened.txt 2>&1
cp $OUT/kt_b/kt_kernel_stats.csv $OUT/${R}_build10m_screened_kernel_stats.csv
rm -rf $OUT/kt_b
grep "k_distances_f32<2, false>" $OUT/${R}_kernel_stats.csv | cut -c140-220
tail -3 $OUT/${R}_forest_levels_screened.txt
grep -h "batch of" $OUT/${R}_levels_timing_100trees.txt $OUT/${R}_levels_timing_13trees.txt | cut -c1-200
cat $OUT/${R}_pmc_kernels.json | head -30
