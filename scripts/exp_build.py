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
ds = Dataset(distances.Cosine, dims, n)
ds.fill_synthetic(42, 1, n)
ds.finalize()
seeds = shard.tree_seeds(42, range(trees))
for r in range(reps):
    t0 = time.perf_counter()
    f = ds.build_forest(seeds)
    el = time.perf_counter() - t0
    st = dict(f.stats)
    st["wall"] = el
    st["env"] = {k: v for k, v in os.environ.items() if k.startswith("AH_")}
    print(json.dumps(st), flush=True)
    f.close()
