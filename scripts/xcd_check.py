"""A/B check of a forest build under two environment settings (run twice, compare the printed digests)."""
import hashlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arroy_amd import Dataset, distances, shard
n, trees, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3], 0)
ds = Dataset(distances.Cosine, 768, n)
ds.fill_synthetic(42, 1, n)
ds.finalize()
f = ds.build_forest(shard.tree_seeds(42, range(trees)), margin_mode=mode)
h = hashlib.sha256()
h.update(f.nodes.tobytes()); h.update(f.descendants.tobytes()); h.update(bytes(f.normals))
print(n, trees, hex(mode), "retries", f.stats["retries"], "fallbacks", f.stats["screen_fallbacks"], "digest", h.hexdigest()[:16])
