"""What a second 10M x 768 dataset's builds look like right after the first one's (bench.py's `normal` leg): the first dataset kept / closed,
the device cache trimmed or not, seconds of six builds in a row."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arroy_amd import Dataset, distances, shard, _lib  # noqa: E402

mode = sys.argv[1]  # keep | close | close_trim
n = 10_000_000
seeds = shard.tree_seeds(42, range(100))
ds = Dataset(distances.Cosine, 768, n)
ds.fill_synthetic(42, 1, n)
ds.finalize()
for _ in range(2):
    ds.build_forest(seeds).close()
if mode != "keep":
    ds.close()
if mode == "close_trim":
    _lib.device_cache_trim()
dn = Dataset(distances.Cosine, 768, n)
dn.fill_synthetic(42, 2, n)
dn.finalize()
dn.build_forest(seeds[:1]).close()
out = []
for rep in range(6):
    t0 = time.perf_counter()
    f = dn.build_forest(seeds)
    el = time.perf_counter() - t0
    out.append((round(el, 4), round(f.stats["seconds_device"], 4), round(f.stats["seconds_after_device"], 4)))
    f.close()
print(mode, json.dumps(out), flush=True)
