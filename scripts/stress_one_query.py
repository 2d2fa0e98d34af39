"""Stress of the one-query call (round 6: results and status written to pinned memory by the last kernel, the host polling the
status word): N calls over a fixed set of queries, every answer compared with the answer of the big submissions' path.  A result
read before it landed, or a stale status word, shows up as a mismatch."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arroy_amd import Dataset, _lib, distances, shard  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n, dims, trees, k, sk = 200_000, 768, 20, 50, 5000
ds = Dataset(distances.Cosine, dims, n)
ds.fill_synthetic(7, 2, n)
ds.finalize()
forest = ds.build_forest(shard.tree_seeds(7, range(trees)))
index = ds.create_index(forest)
rng = np.random.default_rng(1)
qs = np.stack([ds.item_vector(int(i)) for i in rng.choice(n, 256, replace=False)])
qs = (qs + rng.standard_normal(qs.shape).astype(np.float32) * 0.05).astype(np.float32)
with _lib.tuning(AH_SEARCH_BLOCK_MAX_QUERIES=0, AH_SEARCH_SMALL_UNITS_MAX_QUERIES=0, AH_SEARCH_SMALL_TILES_MAX_QUERIES=0, AH_SEARCH_MULTI=0):
    want = index.search(k, queries=qs, search_k=sk, raw=True)
bad = 0
for i in range(calls):
    qi = (i * 37) % len(qs)
    nq = 1 if i % 5 else 2
    got = index.search(k, queries=qs[qi:qi + nq], search_k=sk, raw=True)
    hi = min(len(qs), qi + nq)
    if not (np.array_equal(got[0][:hi - qi], want[0][qi:hi]) and got[1][:hi - qi].tobytes() == want[1][qi:hi].tobytes()
            and np.array_equal(got[2][:hi - qi], want[2][qi:hi])):
        bad += 1
        if bad < 4:
            print("mismatch at call", i, "query", qi, "nq", nq)
st = index.stats()
print(f"{calls} calls, {bad} mismatches; descent_multi {st['descent_multi']}, fallback_chunks {st['fallback_chunks']}")
sys.exit(1 if bad else 0)
