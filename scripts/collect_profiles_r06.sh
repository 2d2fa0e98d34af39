#!/bin/bash
# Run on the GPU box (gpurun -- bash scripts/collect_profiles_r06.sh): the evidence of round 6.
#   1. PMC passes (FETCH_SIZE / WRITE_SIZE, each its own run, no other trace domain) of the scan / re-rank / 1-bit kernels
#      -> profiles/r06_pmc_kernels.json (hash-stamped; what bench.py quotes as `traffic_stored`), in place BEFORE the bench run
#   2. the judged bench line -> r06_bench.json; its rocprofv3 kernel summary -> r06_kernel_stats.csv
#   3. per-level timing of the 10M builds (100 trees, 13-tree share), kernel totals + per-level kernel split of both,
#      the dense levels of the share with the narrow kernel off / on
#   4. the search call kernel by kernel: one query per call (block descent / wave descent), 1000 distinct queries
#   5. bench.py --gpus 4 --virtual (threads) and --gpus 2 --virtual-devices under torch.distributed.run: the N > 1 path on one GPU
set -u
R=r06
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
# (the second build of each: 15 levels, the last two group of trees by group of trees — AH_BUILD_TAIL_GROUPS — i.e. 13 + 2 x 5 level lines)
AH_TIMING=2 python scripts/exp_build.py 10000000 100 2 2>&1 | tail -28 > $OUT/${R}_levels_timing_100trees.txt
AH_TIMING=2 python scripts/exp_build.py 10000000 13 2 2>&1 | tail -28 > $OUT/${R}_levels_timing_13trees.txt
AH_BUILD_TAIL_GROUPS=0 AH_TIMING=2 python scripts/exp_build.py 10000000 100 2 2>&1 | tail -19 > $OUT/${R}_levels_timing_100trees_level_by_level.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_b -o kt -- python scripts/exp_build.py 10000000 100 > $OUT/${R}_build10m_screened.log 2>&1
{
  echo "# kernel time per level of the 10M x 768 x 100-tree build (rocprofv3 --kernel-trace, scripts/level_trace.py).  The script numbers the"
  echo "# levels by counting k_next_scan launches: L13 and L14 run group of trees by group of trees (AH_BUILD_TAIL_GROUPS = 5), so the lines"
  echo "# L13 .. L22 are level 13 and level 14 of groups 0 .. 4 (30 / 23 / 19 / 15 / 13 trees) in turn."
  python scripts/level_trace.py $OUT/kt_b/kt_kernel_trace.csv
} > $OUT/${R}_forest_levels_screened.txt 2>&1
cp $OUT/kt_b/kt_kernel_stats.csv $OUT/${R}_build10m_screened_kernel_stats.csv
rm -rf $OUT/kt_b
{
  echo "# the dense levels of the 13-tree share of 10M x 768 (python scripts/exp_build.py 10000000 13 2, second build), kernel time per"
  echo "# level from rocprofv3 --kernel-trace (scripts/level_trace.py): dense = k_forest_dense_screen / k_forest_dense_narrow"
  for cfg in "k_forest_dense_screen_only AH_DENSE_NARROW=0" "default_narrow_up_to_64_columns AH_DENSE_NARROW=-1" "narrow_up_to_128_columns AH_DENSE_NARROW_MAX_COLS=128"; do
    set -- $cfg
    name=$1; shift
    env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_d -o kt -- python scripts/exp_build.py 10000000 13 2 > $OUT/dense_$name.log 2>&1
    echo "## $name ($*)"
    python scripts/level_trace.py $OUT/kt_d/kt_kernel_trace.csv | grep "tc=mfma" | tail -6
    python scripts/kstats.py $OUT/kt_d/kt_kernel_stats.csv k_forest_dense k_forest_exact k_forest_masks k_forest_advance
    rm -rf $OUT/kt_d
  done
} > $OUT/${R}_dense_narrow_levels.txt 2>&1
{
  echo "# ah_search_batch kernel by kernel, 1M x 1536 dot product, 20 trees, search_k = 10 000, count = 100 (scripts/exp_latency.py NQ CALLS"
  echo "# under rocprofv3 --kernel-trace --stats): average microseconds per launch"
  for cfg in "1 300 AH_SEARCH_BLOCK_MAX_QUERIES=64" "1 300 AH_SEARCH_BLOCK_MAX_QUERIES=0" "8 300 AH_SEARCH_BLOCK_MAX_QUERIES=64" "1000 100 AH_SEARCH_BLOCK_MAX_QUERIES=64"; do
    set -- $cfg
    nq=$1; calls=$2; shift 2
    env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_s -o kt -- python scripts/exp_latency.py $nq $calls > $OUT/lat.log 2>&1
    echo "## $nq queries per call, $* : $(grep "^nq=" $OUT/lat.log | tail -1)"
    python scripts/kstats.py $OUT/kt_s/kt_kernel_stats.csv k_descend k_leaf k_search_select k_flag k_queries k_visit k_prepare k_units
    rm -rf $OUT/kt_s
  done
  env rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_s -o kt -- python scripts/exp_rerank_latency.py 300 > $OUT/lat.log 2>&1
  echo "## ah_rerank_by_vector, one list of 10 000 - 11 535 sorted ids per call (scripts/exp_rerank_latency.py): $(grep "^rerank_by_vector" $OUT/lat.log | tail -1)"
  python scripts/kstats.py $OUT/kt_s/kt_kernel_stats.csv k_prepare_query k_distances k_topk
  rm -rf $OUT/kt_s
  echo "## the same with AH_RERANK_SMALL=0 (the general selection: two tournament rounds, emit, three copies back): $(AH_RERANK_SMALL=0 python scripts/exp_rerank_latency.py 300 2>&1 | tail -1)"
  echo "## ah_search_batch nq = 1 with the small submissions' switches off one at a time (wall time per call, python wrapper included)"
  for knob in AH_SEARCH_BLOCK_MAX_QUERIES AH_SEARCH_SMALL_UNITS_MAX_QUERIES AH_SEARCH_SMALL_TILES_MAX_QUERIES AH_SEARCH_FUSED_FLAG AH_SEARCH_FUSED_PREPARE; do
    echo "$knob=0: $(env $knob=0 python scripts/exp_latency.py 1 300 2>&1 | grep "^nq=" | tail -1)"
  done
} > $OUT/${R}_search_call_kernels.txt 2>&1
python bench.py --gpus 4 --virtual --steps 10 --warmup 2 --no-cpu --no-extra --no-live-pmc 2>$OUT/virtual4.err | tail -1 > $OUT/${R}_bench_virtual_4_threads.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29581 bench.py --gpus 2 --virtual-devices --steps 10 --warmup 2 --no-cpu --no-extra --no-live-pmc 2>$OUT/virtual2.err | grep "^{" | tail -1 > $OUT/${R}_bench_virtual_2_ranks.json
python - <<'PY'
import json
j = json.load(open("gpurun_out/profiles/r06_bench.json"))
b = j["build_10m"]
print("value", j["value"], "frac", j["roofline"]["frac"], "traffic", j["roofline"]["traffic"], j["roofline"]["kernel_ms"])
print("build_10m", b["seconds"], b["seconds_samples"], "dev", b["seconds_device"], b["identical"], "cold", b["cold"]["first_build_s"], b["cold"]["total_s"])
print("stream", b["stream"]["seconds"], "share", b["share_13"]["seconds"], b["share_13"]["seconds_device"], b["share_13"]["speedup_100_trees_over_share"], "normal", b["normal"]["seconds"], "f32", b["f32_only"]["seconds"])
print("union", j.get("build_10m_union"))
s = j["search"]; print("search", {k: round(v["queries_per_s"]) for k, v in s.items() if isinstance(v, dict) and "queries_per_s" in v}, s["verified"])
l = s["latency"]; print("latency", {k: (round(v["p50_us"]), round(v["p99_us"])) for k, v in l.items() if isinstance(v, dict) and "p50_us" in v}, l["cpu_one_core"])
r = j["rerank"]; print("rerank", {k: round(v["queries_per_s"]) for k, v in r.items() if isinstance(v, dict) and "queries_per_s" in v}, r["roofline"]["frac"], r["roofline"]["traffic"], r["roofline_screened"]["frac"])
print("bq", j["bq_scan"]["roofline"]["frac"], j["bq_scan"]["roofline"]["traffic"], "read", j["roofline"]["measured_read_only_gb_per_s"])
c = j["cpu_baseline"]; print("cpu", c["value"], c.get("build_seconds_config_1"), c["build_10m"]["build_seconds_config_2"])
for f in ("r06_bench_virtual_4_threads.json", "r06_bench_virtual_2_ranks.json"):
    v = json.load(open("gpurun_out/profiles/" + f))
    print(f, v["n_gpus"], v["build_10m_union"], {k: (d["trees"], round(d["seconds"], 4), round(d["seconds_device"], 4)) for k, d in v["build_10m_per_device"].items()})
PY
grep "k_distances_f32<2, false>" $OUT/${R}_kernel_stats.csv | cut -c140-220
