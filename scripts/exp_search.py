"""The on-device search leg of bench.py alone (1M x 1536 dot product, 20 trees, 1000 queries; AH_EXP_SHAPE: others), one caller, for
rocprofv3 runs: python scripts/exp_search.py [repeats [distinct base items [filter share]]]."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from arroy_amd import Dataset, distances, shard  # noqa: E402

repeats = int(sys.argv[1]) if len(sys.argv) > 1 else 5
clusters = int(sys.argv[2]) if len(sys.argv) > 2 else 64  # distinct base items of the 1000 queries (bench.py: 64)
keep = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0   # candidate filter: this share of the items (0 = no filter)
# AH_EXP_SHAPE=items,dims,trees,metric (default: the bench leg 1000000,1536,20,dot; e.g. 10000000,768,100,cosine)
shape = os.environ.get("AH_EXP_SHAPE", "1000000,1536,20,dot").split(",")
n, dims, n_trees, nq, k = int(shape[0]), int(shape[1]), int(shape[2]), 1000, 100
cls = {"dot": distances.DotProduct, "cosine": distances.Cosine, "euclidean": distances.Euclidean}[shape[3]]
ds = Dataset(cls, dims, n, device=0)
ds.fill_synthetic(bench.SEED, 1, n)
if cls is distances.DotProduct:
    ds.preprocess_dot()
ds.finalize()
forest = ds.build_forest(shard.tree_seeds(bench.SEED, range(n_trees)))
index = ds.create_index(forest)
rng = np.random.default_rng(bench.SEED)
queries = np.stack([ds.item_vector(int(i)) for i in rng.choice(n, clusters, replace=False)])
queries = (np.tile(queries, (nq // clusters + 1, 1))[:nq] + rng.standard_normal((nq, dims)).astype(np.float32) * 0.05).astype(np.float32)
cand = np.sort(rng.choice(n, int(n * keep), replace=False)).astype(np.uint32) if keep > 0 else None
times = []
for _ in range(repeats + 1):
    t0 = time.perf_counter()
    ids, d, counts = index.search(k, queries=queries, search_k=10_000, raw=True, candidates=cand, candidates_sorted=True)
    times.append(time.perf_counter() - t0)
print(json.dumps({"seconds": times[1:], "queries_per_s": nq / min(times[1:]), "checksum": int(ids.astype(np.uint64).sum())}))
