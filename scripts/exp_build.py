"""Experiment driver (no torch): fill N x 768 cosine rows in HBM, build T trees, print the build statistics.
Used under rocprofv3 (--kernel-trace / --pmc) to study the forest kernels; env switches of forest.hip apply."""
import json
import sys
import time
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arroy_amd import Dataset, distances, shard  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
trees = int(sys.argv[2]) if len(sys.argv) > 2 else 100
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dims = int(sys.argv[4]) if len(sys.argv) > 4 else 768
dist = int(sys.argv[5]) if len(sys.argv) > 5 else 1  # ah_synth_distribution: 1 uniform[-1,1), 2 ~N(0,1), 3 with outlier dimensions
ds = Dataset(distances.Cosine, dims, n)
ds.fill_synthetic(42, dist, n)
ds.finalize()
# 13 trees = the share GPU 0 of an 8-GPU node builds of a 100-tree forest (t = 0 mod 8); otherwise trees 0..T-1
seeds = shard.tree_seeds(42, shard.trees_for_rank(100, 0, 8)) if trees == 13 else shard.tree_seeds(42, range(trees))
for r in range(reps):
    t0 = time.perf_counter()
    f = ds.build_forest(seeds)
    el = time.perf_counter() - t0
    st = dict(f.stats)
    st["wall"] = el
    st["env"] = {k: v for k, v in os.environ.items() if k.startswith("AH_")}
    print(json.dumps(st), flush=True)
    f.close()
