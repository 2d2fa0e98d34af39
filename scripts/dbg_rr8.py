import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from arroy_amd import Dataset, distances, _lib
n, dims, k = 1_000_000, 1536, 100
for dist in (2, 4):
    ds = Dataset(distances.DotProduct, dims, n); ds.fill_synthetic(42, dist, n); ds.preprocess_dot(); ds.finalize()
    rng = np.random.default_rng(1)
    q = np.stack([ds.item_vector(int(i)) for i in rng.choice(n, 16, replace=False)])
    for m in (11000, 3000):
        lists = [np.unique(rng.integers(0, n, size=m + 400, dtype=np.uint32))[:m] for _ in range(16)]
        for kk in (100, 10):
            ds.rerank_stats(reset=True)
            ds.rerank_batch(q, lists, kk)
            st = ds.rerank_stats(reset=True)
            print(dist, m, kk, {a: st[a] for a in ("queries_screened", "survivors", "chunks_int8", "chunks_int8_retried")}, flush=True)
    # per query alone
    for i in range(16):
        ds.rerank_stats(reset=True)
        ds.rerank_batch(q[i:i+1], [lists[i]], 10)
        st = ds.rerank_stats(reset=True)
        print("  query", i, st["survivors"], st["chunks_int8"], st["chunks_int8_retried"])
    ds.close()
