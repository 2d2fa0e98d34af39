#!/usr/bin/env python3
"""arroy's Writer / Reader surface over the GPU library (arroy_amd/index.py), the way arroy's README uses it:
    python examples/quickstart.py        (needs an MI355X; `python -c "import __graft_entry__ as g; g.build()"` first)
"""
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arroy_amd import distances, index  # noqa: E402

dims, n = 256, 20_000
rng = np.random.default_rng(0)
vectors = rng.standard_normal((n, dims)).astype(np.float32)

db = index.Database(distances.Cosine)
writer = index.Writer(db, 0, dims)
for i in range(n):
    writer.add_item(i, vectors[i])
writer.builder(random.Random(42)).n_trees(8).build()

reader = index.Reader.open(db, 0)
print("items:", reader.n_items(), "trees:", reader.n_trees())
for item, dist in reader.nns(5).by_item(7):
    print(f"  neighbour of 7: {item:6d}  cosine distance {dist:.4f}")

# incremental update: new items are routed through the existing trees, overgrown leaves are re-split
for i in range(n, n + 1000):
    writer.add_item(i, rng.standard_normal(dims).astype(np.float32))
writer.del_item(7)
writer.builder(random.Random(43)).n_trees(8).build()
reader = index.Reader.open(db, 0)
print("after the update:", reader.n_items(), "items;", "7 present:", reader.contains_item(7))
print("top-3 for a fresh vector:", [i for i, _ in reader.nns(3).search_k(5000).by_vector(vectors[11])])
