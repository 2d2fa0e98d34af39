"""Round-5 A/B of the dense MFMA screen: the narrow kernel (rows straight to registers, a loader wave for the normals)
against k_forest_dense_screen, on the 13-tree share and the 100-tree build of 10M x 768.  Forest digests must agree."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arroy_amd import Dataset, _lib, distances, shard  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dims = int(sys.argv[2]) if len(sys.argv) > 2 else 768
configs = [
    ("old", dict(AH_DENSE_NARROW=0)),
    ("narrow256", dict(AH_DENSE_NARROW_MAX_COLS=256)),
    ("narrow256_nostream", dict(AH_DENSE_NARROW_MAX_COLS=256, AH_DENSE_NARROW_STREAM=0)),
    ("narrow512", dict(AH_DENSE_NARROW_MAX_COLS=512)),
    ("narrow1024", dict(AH_DENSE_NARROW_MAX_COLS=1024)),
    ("narrow4096", dict(AH_DENSE_NARROW_MAX_COLS=4096)),
    ("narrow64", dict(AH_DENSE_NARROW_MAX_COLS=64, AH_DENSE_NARROW_STREAM=0)),
    ("narrow64_nt", dict(AH_DENSE_NARROW_MAX_COLS=64, AH_DENSE_NARROW_STREAM=1)),
    ("narrow128", dict(AH_DENSE_NARROW_MAX_COLS=128, AH_DENSE_NARROW_STREAM=0)),
    ("narrow128_nt", dict(AH_DENSE_NARROW_MAX_COLS=128, AH_DENSE_NARROW_STREAM=1)),
]
if len(sys.argv) > 3:
    configs = [c for c in configs if c[0] in sys.argv[3].split(",")]
ds = Dataset(distances.Cosine, dims, n)
ds.fill_synthetic(42, 1, n)
ds.finalize()
for trees in [int(t) for t in os.environ.get('R05_TREES', '13,100').split(',')]:
    seeds = shard.tree_seeds(42, shard.trees_for_rank(100, 0, 8)) if trees == 13 else shard.tree_seeds(42, range(trees))
    ref = None
    for name, knobs in configs:
        for rep in range(2):
            with _lib.tuning(AH_TIMING=2 if rep == 1 else 0, **knobs):
                print(f"=== {trees} trees, {name}, rep {rep}", file=sys.stderr, flush=True)
                t0 = time.perf_counter()
                f = ds.build_forest(seeds)
                el = time.perf_counter() - t0
            st = f.stats
            total, _ = f.digest()
            if ref is None:
                ref = total
            print(json.dumps({"trees": trees, "config": name, "rep": rep, "wall": round(el, 4), "device": round(st["seconds_device"], 4),
                              "margin": round(st["seconds_margin"], 4), "dense_launches": st["dense_launches"],
                              "violations": st["screen_violations"], "digest_ok": total == ref}), flush=True)
            f.close()
