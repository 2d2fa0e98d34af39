"""Debug helper (GPU): build one forest config, time it, compare with the oracle node by node."""
import faulthandler; faulthandler.dump_traceback_later(12, exit=True)
import sys, time
import numpy as np
sys.path.insert(0, ".")
from arroy_amd import Dataset, distances as D
from oracle import oracle as O

def first_diff(a, b, path="root"):
    if a[0] != b[0]:
        return f"{path}: kind {a[0]} vs {b[0]}"
    if a[0] == "D":
        if a[1] != b[1]:
            return f"{path}: descendants differ: {len(a[1])} vs {len(b[1])} items; {a[1][:8]} vs {b[1][:8]}"
        return None
    if a[1] != b[1]:
        na = None if a[1] is None else np.frombuffer(a[1], dtype=np.uint8)
        nb = None if b[1] is None else np.frombuffer(b[1], dtype=np.uint8)
        return f"{path}: normal differs: gpu={'None' if na is None else na[:16]} oracle={'None' if nb is None else nb[:16]}"
    return first_diff(a[2], b[2], path + "L") or first_diff(a[3], b[3], path + "R")

def count(t):
    return 1 if t[0] == "D" else 1 + count(t[2]) + count(t[3])

n, dims, k, m = [int(x) for x in sys.argv[1:5]]
rng = np.random.default_rng(n + dims + m)
vecs = rng.standard_normal((n, dims)).astype(np.float32)
vecs[3] = vecs[1]; vecs[n // 2] = vecs[1]; vecs[5] = 0
cls = D.BY_METRIC[m]
ds = Dataset(cls, dims, n)
ds.upload_vectors(np.arange(n, dtype=np.uint32), vecs)
od = O.Data(m, vecs)
if m == 3:
    ds.preprocess_dot(); od.preprocess_dot()
ds.finalize()
seeds = [int(x) for x in sys.argv[5:]] or [42, 7]
for seed in seeds:
    t = time.time()
    f = ds.build_forest([seed], split_after=k, progress=lambda lvl, nodes, items: print('   level', lvl, 'records', nodes, 'items', items, flush=True))
    el = time.time() - t
    ref = od.build_tree(k, seed)
    a, b = f.canonical(0), ref.canonical()
    print(f"cfg={n},{dims},{k},{m} seed={seed} gpu_s={el:.3f} levels={f.stats['levels']} nodes={len(f.nodes)} "
          f"retries={f.stats['retries']} dummy={f.stats['dummy_normals']} | oracle nodes={len(ref.nodes)} "
          f"retries={ref.retries} dummy={ref.dummy_normals} | equal={a == b}", flush=True)
    if a != b:
        print("   first diff:", first_diff(a, b), flush=True)
