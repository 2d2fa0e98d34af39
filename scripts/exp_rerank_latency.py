"""Latency experiment of the single-query re-rank (ah_rerank_by_vector: what integration/arroy-hip/src/hip.rs wires for
`Reader::nns_by_vector`'s last loop, src/reader.rs:381-399): 1M x 1536 dot-product rows, one list of 10 000 - 11 535 sorted ids
per call.  Under `rocprofv3 --kernel-trace --stats` the per-kernel averages say where the call's time goes."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arroy_amd import Dataset, distances, _lib  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n, dims, k = 1_000_000, 1536, 100
ds = Dataset(distances.DotProduct, dims, n)
ds.fill_synthetic(42, 1, n)
ds.preprocess_dot()
ds.finalize()
rng = np.random.default_rng(42)
far = np.stack([ds.item_vector(int(i)) for i in rng.choice(n, 64, replace=False)])
far = (far + rng.standard_normal(far.shape).astype(np.float32) * 0.05).astype(np.float32)
lists = [np.sort(rng.choice(n, int(rng.integers(10_000, 11_536)), replace=False)).astype(np.uint32) for _ in range(32)]
L = _lib.lib()
oi, od, on = np.zeros(k, np.uint32), np.zeros(k, np.float32), C.c_size_t(0)
samples = []
for i in range(calls + 20):
    ids, q = lists[i % 32], far[i % 64]
    t0 = time.perf_counter()
    st = L.ah_rerank_by_vector(ds._h, q.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p), ids.size, k,
                               oi.ctypes.data_as(C.c_void_p), od.ctypes.data_as(C.c_void_p), C.byref(on))
    el = time.perf_counter() - t0
    if st:
        _lib.check(st)
    if i >= 20:
        samples.append(el)
a = np.sort(samples) * 1e6
print(f"rerank_by_vector: p50 {a[len(a) // 2]:.1f} us  p99 {a[int(len(a) * 0.99)]:.1f} us  mean {a.mean():.1f} us")
