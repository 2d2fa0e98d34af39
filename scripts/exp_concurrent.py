"""Experiment: the 100-tree 10M build as ONE call vs. as K concurrent calls of 100/K trees each (host threads, one HIP
stream per call).  If the latency-bound bookkeeping of one call hides under the bandwidth-bound margin passes of another,
the wall time drops: that would be worth doing inside ah_build_forest."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arroy_amd import Dataset, distances, shard  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
trees = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ds = Dataset(distances.Cosine, 768, n)
ds.fill_synthetic(42, 1, n)
ds.finalize()
seeds = shard.tree_seeds(42, range(trees))
ds.build_forest(seeds[:1]).close()
for k in (1, 2, 4, 1, 2):
    parts = [seeds[i::k] for i in range(k)]
    out = [None] * k

    def run(i):
        out[i] = ds.build_forest(parts[i])
    t0 = time.perf_counter()
    ths = [threading.Thread(target=run, args=(i,)) for i in range(k)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    el = time.perf_counter() - t0
    dev = [round(f.stats["seconds_device"], 3) for f in out]
    print(f"{k} concurrent calls of {len(parts[0])} trees: wall {el:.3f} s, device seconds per call {dev}", flush=True)
    for f in out:
        f.close()
