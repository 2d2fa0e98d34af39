# kernel-level profile of the extra configs (C4 re-rank, C5 1-bit scan, end-to-end search)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_extra -o extra -- python bench.py --steps 5 --warmup 1 --no-cpu --no-build --extra ${1:-c4,c5,search} > gpurun_out/extra.log 2>&1
grep -v "^[EWI]2026" gpurun_out/extra.log | tail -3
