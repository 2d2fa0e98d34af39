#!/bin/bash
# Run on the GPU box (gpurun -- bash scripts/collect_profiles.sh): the judged bench line, its rocprofv3 kernel summary,
# the two PMC passes (FETCH_SIZE / WRITE_SIZE, each in its own run, no trace domains besides the kernel trace), and the
# extra configurations.  Everything lands in gpurun_out/profiles/ and is copied into profiles/ by hand afterwards.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles
mkdir -p $OUT
python bench.py --steps 50 --warmup 5 2>$OUT/bench.err | tail -1 > $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python bench.py --steps 50 --warmup 5 --no-cpu --no-build-10m > $OUT/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python bench.py --steps 5 --warmup 1 --no-cpu --no-build-10m > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python bench.py --steps 5 --warmup 1 --no-cpu --no-build-10m > $OUT/write.log 2>&1
python bench.py --steps 20 --warmup 3 --no-cpu --extra c3,c4,c5,search,staging 2>$OUT/extra.err | tail -1 > $OUT/bench_extra.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktx -o ktx -- python bench.py --steps 5 --warmup 1 --no-cpu --no-build --extra c3,c4,c5,search > $OUT/ktx.log 2>&1
python scripts/level_trace.py $OUT/ktx/ktx_kernel_trace.csv > $OUT/c3_levels.txt 2>&1
rm -f $OUT/kt/kt_kernel_trace.csv $OUT/ktx/ktx_kernel_trace.csv   # large; the stats files are the summaries
ls -la $OUT $OUT/*
