"""Latency experiment: the 1M x 1536 dot-product index of bench.py's search leg (20 trees), then `calls` ah_search_batch calls of
`nq` queries each (distinct queries every call).  Under `rocprofv3 --kernel-trace --stats` the per-kernel averages say where a
single-query call spends its time; without it the script prints the wall-time percentiles."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arroy_amd import Dataset, distances, shard  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 300
# AH_EXP_SHAPE=items,dims,trees,metric (default: the bench leg 1000000,1536,20,dot; e.g. 10000000,768,100,cosine)
shape = os.environ.get("AH_EXP_SHAPE", "1000000,1536,20,dot").split(",")
n, dims, k, n_trees = int(shape[0]), int(shape[1]), 100, int(shape[2])
cls = {"dot": distances.DotProduct, "cosine": distances.Cosine, "euclidean": distances.Euclidean}[shape[3]]
ds = Dataset(cls, dims, n)
ds.fill_synthetic(42, 1, n)
if cls is distances.DotProduct:
    ds.preprocess_dot()
ds.finalize()
forest = ds.build_forest(shard.tree_seeds(42, range(n_trees)))
index = ds.create_index(forest)
rng = np.random.default_rng(42)
far = np.stack([ds.item_vector(int(i)) for i in rng.choice(n, 2000, replace=False)])
far = (far + rng.standard_normal(far.shape).astype(np.float32) * 0.05).astype(np.float32)
index.search(k, queries=far[:64], search_k=10_000, raw=True)
samples = []
for i in range(calls + 20):
    q = far[(i * nq) % (2000 - nq):][:nq]
    t0 = time.perf_counter()
    index.search(k, queries=q, search_k=10_000, raw=True)
    if i >= 20:
        samples.append(time.perf_counter() - t0)
a = np.sort(samples) * 1e6
print(f"nq={nq}: p50 {a[len(a) // 2]:.1f} us  p99 {a[int(len(a) * 0.99)]:.1f} us  mean {a.mean():.1f} us (python wrapper included)")
print(index.stats())
