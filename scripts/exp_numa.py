"""AH_NUMA experiment: the first build of the process runs with the calling thread confined to the host node the GPU does NOT hang off
(its blobs are first-touched there), the later ones with the whole machine again (the blobs are recycled): seconds of every build."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arroy_amd import Dataset, distances, shard  # noqa: E402


def cpus_of(node):
    out = set()
    for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out


far = int(sys.argv[1])  # node to confine the first build to
n = 10_000_000
ds = Dataset(distances.Cosine, 768, n)
ds.fill_synthetic(42, 1, n)
ds.finalize()
seeds = shard.tree_seeds(42, range(100))
everything = os.sched_getaffinity(0)
os.sched_setaffinity(0, cpus_of(far) & everything)
for rep in range(5):
    if rep == 1:
        os.sched_setaffinity(0, everything)
    t0 = time.perf_counter()
    f = ds.build_forest(seeds)
    el = time.perf_counter() - t0
    print(json.dumps({"rep": rep, "wall": round(el, 4), "device": round(f.stats["seconds_device"], 4), "after": round(f.stats["seconds_after_device"], 4),
                      "recycled": f.stats["host_blob_recycled"]}), flush=True)
    f.close()
