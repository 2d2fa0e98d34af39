"""Throughput experiment of ah_rerank_batch as bench.py's `rerank` leg runs it: 1M x 1536 dot-product rows, 1000 queries x 10 000 - 11 535
sorted candidates in submissions of 125 queries from one caller.  Round 5 used it for the A/B of a selection that ran group by group on a
second stream (switch AH_RERANK_SELECT_OVERLAP of that build; measured + 3 % here, - 8 % with four callers, removed — DESIGN_HISTORY.md "Measured and
rejected"); what is left compares the certified top-k screen — int8 rows first (round 6), binary16 rows first (round 4) — with the f32-only
path, answers compared bit for bit, and prints where the wall time of a pass goes (ah_dataset_rerank_stats)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arroy_amd import Dataset, distances, _lib  # noqa: E402

n, dims, nq, k = 1_000_000, 1536, 1000, 100
ds = Dataset(distances.DotProduct, dims, n)
ds.fill_synthetic(42, 1, n)
ds.preprocess_dot()
ds.finalize()
rng = np.random.default_rng(42)
queries = np.stack([ds.item_vector(int(i)) for i in rng.choice(n, 64, replace=False)])
queries = np.tile(queries, (nq // 64 + 1, 1))[:nq]
lists = []
for _ in range(nq):
    m = int(rng.integers(10_000, 11_536))
    lists.append(np.unique(rng.integers(0, n, size=m + 400, dtype=np.uint32))[:m])


def flat(b, e):
    off = np.zeros(e - b + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(l) for l in lists[b:e]])
    return queries[b:e], (np.concatenate(lists[b:e]), off)


batches = [flat(b, min(nq, b + 125)) for b in range(0, nq, 125)]
ref = None
dist = int(sys.argv[1]) if len(sys.argv) > 1 else 1
if dist != 1:  # another distribution of include/arroy_hip_policy.h (4: clustered rows) for the same experiment
    ds.close()
    ds = Dataset(distances.DotProduct, dims, n)
    ds.fill_synthetic(42, dist, n)
    ds.preprocess_dot()
    ds.finalize()
for label, knobs in (("int8 first", {}), ("binary16", dict(AH_RERANK_SCREEN8=0)), ("f32 only", dict(AH_RERANK_SCREEN=0)), ("int8 first", {}),
                     ("binary16", dict(AH_RERANK_SCREEN8=0)), ("f32 only", dict(AH_RERANK_SCREEN=0))):
    with _lib.tuning(AH_RERANK_TIMING=1, **knobs):
        ds.rerank_stats(reset=True)
        for q, l in batches[:2]:
            ds.rerank_batch(q, l, k)
        best = []
        for rep in range(5):
            t0 = time.perf_counter()
            outs = [ds.rerank_batch(q, l, k) for q, l in batches]
            best.append(time.perf_counter() - t0)
        el = sorted(best)[len(best) // 2]
        st = ds.rerank_stats(reset=True)
    got = np.concatenate([o[0] for o in outs]), np.concatenate([o[1] for o in outs])
    if ref is None:
        ref = got
    same = np.array_equal(ref[0], got[0]) and ref[1].tobytes() == got[1].tobytes()
    print(f"{label:12s} {nq / el:9.0f} queries/s  ({el * 1e3:.2f} ms per 1000 queries, median of 5)  identical to the first: {same}  "
          f"per pass: sync_wait {st['seconds_sync_wait'] / 5 * 1e3:.2f} ms, ids {st['seconds_ids'] / 5 * 1e3:.2f}, prep {st['seconds_prep'] / 5 * 1e3:.2f}, "
          f"enqueue {st['seconds_enqueue'] / 5 * 1e3:.2f}; survivors per screened query {st['survivors'] / max(1, st['queries_screened']):.0f}, "
          f"int8 sub-batches {st['chunks_int8']} (+ {st['chunks_int8_retried']} redone on binary16)")
