#!/bin/bash
# last GPU call of round 3: PMC traffic of the three roofline kernels (sources as committed), the search leg's kernel
# summary, then the whole suite, the smoke and the judged bench line with the fresh traffic file in place
set -u
R=r03
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles
mkdir -p $OUT gpurun_out/final
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python bench.py --steps 5 --warmup 1 --no-cpu --no-build --no-search > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python bench.py --steps 5 --warmup 1 --no-cpu --no-build --no-search > $OUT/write.log 2>&1
python scripts/pmc_summary.py $OUT/fetch/fetch_counter_collection.csv | grep -v "^[0-9]" > $OUT/${R}_pmc_fetch_size.csv
python scripts/pmc_summary.py $OUT/write/write_counter_collection.csv | grep -v "^[0-9]" > $OUT/${R}_pmc_write_size.csv
python scripts/pmc_kernels_json.py $OUT/${R}_pmc_fetch_size.csv $OUT/${R}_pmc_write_size.csv > $OUT/${R}_pmc_kernels.json
rm -rf $OUT/fetch $OUT/write
cp $OUT/${R}_pmc_kernels.json $OUT/${R}_pmc_fetch_size.csv $OUT/${R}_pmc_write_size.csv profiles/
# the search leg: kernel summary + HBM reads of its kernels
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_search -o kt -- python scripts/exp_search.py 5 > $OUT/search.log 2>&1
f=$(find $OUT/kt_search -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && grep -v "k_forest\|k_dense\|k_rows\|k_shadow\|k_col\|k_dim\|k_next\|k_build_tiles" "$f" > $OUT/${R}_search_kernel_stats.csv
rm -rf $OUT/kt_search
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_search -o pmc -- python scripts/exp_search.py 2 > $OUT/search_pmc.log 2>&1
f=$(find $OUT/fetch_search -name '*counter_collection.csv' | head -1)
[ -n "$f" ] && python scripts/pmc_summary.py "$f" | grep "kernel,counter\|k_leaf_tiles\|k_descend\|k_search_select\|k_flag" > $OUT/${R}_search_pmc_fetch_size.csv
rm -rf $OUT/fetch_search
# suite, smoke, bench
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/final/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/final/pytest.log
tail -3 gpurun_out/final/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; tail -1 gpurun_out/final/smoke.log
python bench.py --steps 50 --warmup 5 2>gpurun_out/final/bench.err | tail -1 > gpurun_out/final/r03_bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/final/r03_bench.json'))
print(d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['build_10m']['seconds'], d['build_10m']['identical'], d['build_10m']['share_13']['seconds'], d['build_10m']['normal']['seconds'], d['rerank']['roofline']['traffic'], d['bq_scan']['roofline']['traffic'], d['search']['callers_1']['queries_per_s'])
PY
