"""8-GPU readiness on one box (round-3 review, item 5b): K concurrent 13-tree builds of the 10M x 768 dataset from K host
threads of ONE process — the host side of what eight GPUs would ask of a 16-CPU container — with a host-thread budget per
build (ah_build_options.max_host_threads).  The GPU is shared here, so the device seconds of a call grow with K; what the
experiment shows is the HOST part of every call (wall - device) and how it moves with the budget."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arroy_amd import Dataset, distances, shard  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ds = Dataset(distances.Cosine, 768, n)
ds.fill_synthetic(42, 1, n)
ds.finalize()
ds.build_forest(shard.tree_seeds(42, [0])).close()
for k, budget in ((1, 8), (1, 2), (2, 8), (2, 2), (4, 8), (4, 2), (8, 2), (8, 1)):
    parts = [shard.tree_seeds(42, shard.trees_for_rank(100, r, 8)) for r in range(k)]
    out = [None] * k

    def run(i):
        t0 = time.perf_counter()
        f = ds.build_forest(parts[i], max_host_threads=budget)
        out[i] = (time.perf_counter() - t0, f.stats["seconds_device"], f.stats["seconds_setup"], f.stats["seconds_after_device"])
        f.close()
    for rep in range(2):
        t0 = time.perf_counter()
        ths = [threading.Thread(target=run, args=(i,)) for i in range(k)]
        [t.start() for t in ths]
        [t.join() for t in ths]
        el = time.perf_counter() - t0
    host = [round(w - d, 3) for w, d, _s, _a in out]
    print(f"{k} concurrent 13-tree builds, {budget} host threads each: wall {el:.3f} s; per call wall - device {host}, "
          f"after the last launch {[round(a, 3) for _w, _d, _s, a in out]}", flush=True)
