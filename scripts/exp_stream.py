"""Experiment driver: the 10M x 768 x 100-tree build through ah_build_forest_stream with a sink that only counts (what bench.py's
`stream` leg times), AH_TIMING output included."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arroy_amd import Dataset, distances, shard  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
trees = int(sys.argv[2]) if len(sys.argv) > 2 else 100
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ds = Dataset(distances.Cosine, 768, n)
ds.fill_synthetic(42, 1, n)
ds.finalize()
seeds = shard.tree_seeds(42, range(trees))
f = ds.build_forest(seeds)
f.close()
cnt = {"calls": 0, "bytes": 0}


def sink(b):
    cnt["calls"] += 1
    cnt["bytes"] += int(b.payload_len)
    return 0


for r in range(reps):
    cnt.update(calls=0, bytes=0)
    t0 = time.perf_counter()
    _roots, st, _c = ds.build_forest_stream(seeds, sink=sink)
    el = time.perf_counter() - t0
    print(json.dumps({"wall": el, "seconds_total": st["seconds_total"], "seconds_device": st["seconds_device"],
                      "seconds_after_device": st["seconds_after_device"], "sink_calls": cnt["calls"], "gb": cnt["bytes"] / 1e9}), flush=True)
