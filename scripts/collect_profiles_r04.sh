#!/bin/bash
# Run on the GPU box (gpurun -- bash scripts/collect_profiles_r04.sh): the evidence of round 4.
#   1. PMC passes (FETCH_SIZE / WRITE_SIZE, each its own run, no other trace domain) of the scan / re-rank / 1-bit kernels
#      -> profiles/r04_pmc_kernels.json (hash-stamped; what bench.py quotes as `traffic`), put in place BEFORE the bench run
#   2. the judged bench line -> r04_bench.json; its rocprofv3 kernel summary -> r04_kernel_stats.csv
#   3. per-level timing of the 10M builds (100 trees, 13-tree share, ~N(0,1) rows), kernel totals of one build
#   4. the on-device search leg kernel by kernel (clustered / distinct queries), HBM reads of the screened tiles
#   5. K concurrent 13-tree builds with a host-thread budget (8-GPU readiness on one box)
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
cp $OUT/${R}_pmc_kernels.json profiles/${R}_pmc_kernels.json
rm -rf $OUT/fetch $OUT/write
python bench.py --steps 50 --warmup 5 2>$OUT/${R}_bench.err | tail -1 > $OUT/${R}_bench.json
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
for cfg in "clustered 64" "distinct 1000"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_s -o kt -- python scripts/exp_search.py 10 $2 > $OUT/search_$1.log 2>&1
  { echo "# python scripts/exp_search.py 10 $2 ($1 queries): $(grep queries_per_s $OUT/search_$1.log | cut -c1-80)"; python scripts/kstats.py $OUT/kt_s/kt_kernel_stats.csv k_descend k_leaf k_search_select k_flag k_queries k_visit k_prepare; } > $OUT/${R}_search_kernels_$1.txt
  rm -rf $OUT/kt_s
done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fs -o fs -- python scripts/exp_search.py 3 1000 > $OUT/search_fetch.log 2>&1
python scripts/pmc_summary.py $OUT/fs/fs_counter_collection.csv | grep -E "k_leaf_tiles|k_search_select|k_descend_wave" > $OUT/${R}_search_pmc_fetch_size.csv
rm -rf $OUT/fs
{ AH_EXP_SHAPE=10000000,768,100,cosine python scripts/exp_search.py 5 64; AH_EXP_SHAPE=10000000,768,100,cosine python scripts/exp_search.py 5 1000; AH_SEARCH_SCREEN=0 AH_EXP_SHAPE=10000000,768,100,cosine python scripts/exp_search.py 5 1000; } 2>&1 | grep queries_per_s | cut -c1-200 > $OUT/${R}_search_10m.txt
timeout 300 python scripts/exp_concurrent_share.py > $OUT/${R}_concurrent_shares.txt 2>&1
ls -la $OUT
