#!/bin/bash
# the PMC passes of scripts/collect_profiles_r04.sh alone + the judged bench line with their traffic in place
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
python - <<'PY'
import json
j = json.load(open("gpurun_out/profiles/r04_bench.json"))
b = j["build_10m"]
print("value", j["value"], "frac", j["roofline"]["frac"], "traffic", j["roofline"]["traffic"], j["roofline"]["kernel_ms"])
print("build_10m", b["seconds"], b["seconds_samples"], "dev", b["seconds_device"], b["identical"], "cold", b["cold"]["first_build_s"], b["cold"]["total_s"])
print("stream", b["stream"]["seconds"], "share", b["share_13"]["seconds"], b["share_13"]["speedup_100_trees_over_share"], "normal", b["normal"]["seconds"], "f32", b["f32_only"]["seconds"])
s = j["search"]; print("search", {k: round(v["queries_per_s"]) for k, v in s.items() if isinstance(v, dict) and "queries_per_s" in v}, s["verified"])
r = j["rerank"]; print("rerank", {k: round(v["queries_per_s"]) for k, v in r.items() if isinstance(v, dict) and "queries_per_s" in v}, r["roofline"]["frac"], r["roofline"]["traffic"])
print("bq", j["bq_scan"]["roofline"]["frac"], j["bq_scan"]["roofline"]["traffic"], "read", j["roofline"]["measured_read_only_gb_per_s"])
c = j["cpu_baseline"]; print("cpu", c["value"], c.get("build_seconds_config_1"), c["build_10m"]["build_seconds_config_2"])
PY
grep "k_distances_f32<2, false>" $OUT/${R}_kernel_stats.csv | cut -c140-220
