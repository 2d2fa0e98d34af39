#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity sweep (not part of the test suite; run it on a GPU box for as long as you like):
    python scripts/fuzz_gpu.py [seconds] [seed]
Every iteration draws a metric, a shape, an id pattern and a set of operations and asserts bit-exact agreement with
the CPU oracle, like tests/test_gpu_parity.py does for its fixed cases."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OMP_NUM_THREADS", "16")

import test_gpu_parity as T  # noqa: E402  (helpers: make_data, assert_bit_equal, check_forest_valid)
from arroy_amd import distances as D  # noqa: E402
from oracle import oracle as O  # noqa: E402

T.D, T.O = D, O  # the test module imports these lazily through a fixture


def one(rng, it):
    metric = int(rng.integers(0, 7))
    cls = D.BY_METRIC[metric]
    dims = int(rng.choice([1, 2, 3, 15, 16, 17, 31, 32, 33, 63, 64, 65, 96, 100, 127, 128, 200, 257, 520, 768]))
    n = int(rng.choice([1, 2, 7, 64, 65, 300, 1000, 2049, 5000, 12000]))
    ids = None
    if rng.random() < 0.4:
        span = int(n * rng.choice([2, 50, 100000]))
        ids = np.sort(rng.choice(max(span, n + 1), n, replace=False)).astype(np.uint32)
    scale = float(rng.choice([1.0, 1.0, 1e-3, 1e3, 1e-6, 3e4, 7e-24]))  # the last three leave the binary16 range of the screen; at 7e-24 f32 squares underflow
    # half of the draws: numpy's i.i.d. N(0,1); the rest the policy header's outlier / clustered (with exact duplicates) /
    # low-rank rows — margins crowding the planes, imbalance retries, the random fallback
    dist = [None, None, None, 3, 4, 5][int(rng.integers(0, 6))]
    ds, oracle, vecs, ids = T.make_data(cls, n, dims, seed=int(rng.integers(1 << 30)), ids=ids, scale=scale, dist=dist)
    desc = f"it={it} metric={metric} n={n} dims={dims} sparse={ids[-1] != n - 1} scale={scale} dist={dist}"
    q = (rng.standard_normal(dims) * scale).astype(np.float32)
    qv, qh = oracle.query_leaf(q)
    # distances (scan + gather)
    T.assert_bit_equal(ds.distances(query=q), oracle.distances(qv, qh), desc + " scan")
    m = int(rng.integers(1, n + 1))
    rows = np.sort(rng.choice(n, m, replace=False)).astype(np.uint32)
    T.assert_bit_equal(ds.distances(query=q, ids=ids[rows]), oracle.distances(qv, qh, rows), desc + " gather")
    # re-rank
    k = int(rng.choice([1, 3, 10, 100, 3000]))
    oi, od = ds.rerank(k, query=q, sorted_ids=ids[rows])
    ei, ed = oracle.rerank(qv, qh, rows, k)
    assert list(oi) == [int(x) for x in ei], desc + " rerank ids"
    T.assert_bit_equal(od, ed, desc + " rerank dists")
    from arroy_amd._lib import tuning as _tuning0
    with _tuning0(AH_RERANK_SMALL=0):  # the general selection behind the one-launch one (k_topk_small) of short lists
        gi, gd = ds.rerank(k, query=q, sorted_ids=ids[rows])
    assert list(gi) == list(oi), desc + " rerank small / general ids"
    T.assert_bit_equal(gd, od, desc + " rerank small / general dists")
    # batched re-rank (both the query-major and the row-major path occur)
    nq = int(rng.choice([1, 3, 40]))
    qs = (rng.standard_normal((nq, dims)) * scale).astype(np.float32)
    lists = [np.sort(rng.choice(n, int(rng.integers(0, n + 1)), replace=False)).astype(np.uint32) for _ in range(nq)]
    kb = int(rng.choice([1, 5, 50]))
    bi, bd, bc = ds.rerank_batch(qs, [ids[l] for l in lists], kb)
    from arroy_amd._lib import tuning as _tuning
    with _tuning(AH_RERANK_SCREEN=0):  # the certified top-k screen of the lists (default) and the f32-only path: the same bits
        xi, xd, xc = ds.rerank_batch(qs, [ids[l] for l in lists], kb)
    assert np.array_equal(bc, xc) and np.array_equal(bi, xi) and np.array_equal(bd.view(np.uint32), xd.view(np.uint32)), desc + " batch screen on/off"
    for i in range(nq):
        v, h = oracle.query_leaf(qs[i])
        ei, ed = oracle.rerank(v, h, lists[i], kb) if len(lists[i]) else (np.zeros(0, np.uint32), np.zeros(0, np.float32))
        assert bc[i] == len(ei) and list(bi[i, : bc[i]]) == [int(x) for x in ei], desc + f" batch q={i}"
        T.assert_bit_equal(bd[i, : bc[i]], ed, desc + f" batch q={i}")
    # split + sides
    if n >= 2:
        sample = rng.choice(n, 12, replace=True).astype(np.uint32)
        if sample[0] == sample[1]:
            sample[1] = (sample[0] + 1) % n
        nv, nh = ds.create_split(ids[sample])
        ev, eh = oracle.create_split(sample)
        assert nv.tobytes() == ev.tobytes(), desc + " normal"
        T.assert_bit_equal(nh, eh, desc + " normal header")
        sides, n_left, margins = ds.split_sides(nv, nh)
        es, el, em = oracle.split_sides(ev, eh)
        T.assert_bit_equal(margins, em, desc + " margins")
        assert np.array_equal(sides, es) and n_left == el, desc + " sides"
    # forest + search + routing
    split_after = int(rng.choice([0, 1, 2, 8, 50, 300]))
    seeds = [int(x) for x in rng.integers(0, 2**63, int(rng.choice([1, 2, 3, 9, 17])))]
    # every margin-kernel family, with and without the certified binary16 screen (ah_margin_mode)
    mode = int(rng.choice([0, 0, 1, 2, 4, 8, 16, 0x108, 0x110, 0x200, 0x200])) | (0x1000 if rng.random() < 0.3 else 0)
    mask_bits = int(rng.integers(0, 2))  # the sides of a row-major level gathered as bits / as bytes
    narrow = int(rng.choice([64, 128, 256, 0]))  # widest level the narrow dense kernel takes (64 / 128-column shapes; 0: never)
    # the tail of the build in groups of trees (0: every level for all trees), from levels of various fullness on; AH_ROWMAJOR=0
    # makes every level node-major, i.e. one that may be cut
    tail = dict(AH_BUILD_TAIL_GROUPS=int(rng.choice([0, 2, 3, 4, 7, 32])), AH_BUILD_TAIL_MIN_MB=0,
                AH_BUILD_TAIL_NODE_ITEMS=int(rng.choice([1, 2, 2, 4, 64])), AH_ROWMAJOR=int(rng.choice([-1, -1, 0])))
    with _tuning(AH_MASK_BITS=mask_bits, AH_DENSE_NARROW_MAX_COLS=narrow, AH_DENSE_NARROW_STREAM=int(rng.integers(0, 2)), **tail):
        forest = ds.build_forest(seeds, split_after=split_after, margin_mode=mode)
    desc += f" mode={mode:#x} trees={len(seeds)} mask_bits={mask_bits} narrow={narrow} tail={tail} groups_run={forest.stats['tail_groups']}"
    assert forest.stats["screen_violations"] == 0
    T.check_forest_valid(forest, n, ids=ids)
    for t, seed in enumerate(seeds):
        assert forest.canonical(t) == oracle.build_tree(split_after, seed).canonical(), desc + f" tree {t} sa={split_after}"
    # the same forest through the streaming sink (ah_build_forest_stream), sometimes in several batches of trees
    if rng.random() < 0.5:
        in_flight = int(rng.choice([0, 0, 1, 2]))
        try:
            with _tuning(**tail):
                _roots, sstats, streamed = ds.build_forest_stream(seeds, split_after=split_after, margin_mode=mode, max_trees_in_flight=in_flight)
        except BaseException as e:
            raise AssertionError(desc + f" stream in_flight={in_flight} sa={split_after}: {e!r}")
        assert [streamed.canonical(t) for t in range(len(seeds))] == [forest.canonical(t) for t in range(len(seeds))], desc + " stream"
        assert len(streamed.splits) + len(streamed.leaves) == len(forest.nodes), desc + " stream node count"
    index = ds.create_index(forest)
    count, search_k = int(rng.choice([1, 10, 200, 2500])), int(rng.choice([0, 1, 100, 2**62]))
    got = index.search(count, queries=qs[: min(nq, 4)], search_k=search_k)
    for i in range(min(nq, 4)):
        v, h = oracle.query_leaf(qs[i])
        want, _ = O.search(oracle, forest, v, h, count, search_k, 0, None)
        assert [a for a, _ in got[i]] == [a for a, _ in want], desc + f" search q={i} count={count} sk={search_k}"
        T.assert_bit_equal([b for _, b in got[i]], [b for _, b in want], desc + " search dists")
    # the same submission by every path of ah_search_batch: wave / octet descent x leaf tiles / sorted re-rank
    from arroy_amd._lib import tuning
    qs2 = np.concatenate([qs, vecs[rng.integers(n, size=6)], vecs[rng.integers(n, size=3)] + np.float32(1e-4 * scale)])
    count2, sk2 = int(rng.choice([1, 7, 64, 900])), int(rng.choice([0, 30, 700, 9000, 2**62]))
    cand = None
    if rng.random() < 0.5:  # `candidates: &RoaringBitmap`, item ids (some of them not in the database)
        cand = [int(x) for x in rng.choice(int(ids[-1]) + 3, int(rng.integers(0, min(n, 4000) + 1)), replace=False)]
    ref = None
    for wave in (2, 1, 0):  # 2: one block of 32 octets per query (the default for submissions this small); 1: one wave; 0: one octet
        for tiles in (1, 0, 2):  # 2: the leaf tiles without the certified top-k screen (f32 rows for every candidate)
            if tiles == 2 and wave == 0:
                continue
            # (the small submissions' own kernels — unit builder, in-flight leaf tiles, fused flags / query preparation — on or off at
            # random: every combination returns the same bits)
            small = dict(AH_SEARCH_SMALL_UNITS_MAX_QUERIES=int(rng.choice([0, 64])), AH_SEARCH_SMALL_TILES_MAX_QUERIES=int(rng.choice([0, 8, 64])),
                         AH_SEARCH_FUSED_FLAG=int(rng.integers(0, 2)), AH_SEARCH_FUSED_PREPARE=int(rng.integers(0, 2)),
                         AH_SEARCH_SINGLE_FUSED=int(rng.integers(0, 2)),
                         # (round 6: the trees of a query dealt over several blocks, 1 - 8 trees each; the single query's ids copied by
                         # the tile launch or by the descent)
                         AH_SEARCH_MULTI=int(rng.integers(0, 2)), AH_SEARCH_MULTI_TREES_PER_BLOCK=int(rng.integers(1, 9)),
                         AH_SEARCH_MULTI_IDS_BY_TILES=int(rng.integers(0, 2)), AH_SEARCH_FLAT_TILES=int(rng.integers(0, 2)),
                         AH_SEARCH_ITEM_LIST=int(rng.integers(0, 2)), AH_SEARCH_STATUS_WIPE=int(rng.integers(0, 2)),
                         AH_SEARCH_SPIN_WAIT=int(rng.integers(0, 2)), AH_SEARCH_MULTI_OWN_UNITS=int(rng.integers(0, 2))) if ref is not None else {}
            with tuning(AH_SEARCH_WAVE=min(wave, 1), AH_SEARCH_BLOCK_MAX_QUERIES=64 if wave == 2 else 0, AH_SEARCH_TILES=min(tiles, 1),
                        AH_SEARCH_SCREEN=0 if tiles == 2 else 1, **small):
                oi, od, oc = index.search(count2, queries=qs2, search_k=sk2, candidates=cand, raw=True)
                if wave == 2:  # ... and a call of ONE query (arroy's own shape) is the same row
                    pick = int(rng.integers(0, len(qs2)))
                    o1 = index.search(count2, queries=qs2[pick:pick + 1], search_k=sk2, candidates=cand, raw=True)
                    assert o1[2][0] == oc[pick] and np.array_equal(o1[0][0], oi[pick]) and np.array_equal(o1[1][0].view(np.uint32), od[pick].view(np.uint32)), \
                        desc + f" one-query call differs from its row in the batch: tiles={tiles} q={pick} count={count2} sk={sk2} {small}"
                    # ... and a call of five (up to 8 queries a call take k_descend_multi when the index has more than 8 trees)
                    o5 = index.search(count2, queries=qs2[:5], search_k=sk2, candidates=cand, raw=True)
                    assert np.array_equal(o5[2], oc[:5]) and np.array_equal(o5[0], oi[:5]) and np.array_equal(o5[1].view(np.uint32), od[:5].view(np.uint32)), \
                        desc + f" five-query call differs from its rows in the batch: tiles={tiles} count={count2} sk={sk2} {small}"
            if ref is None:
                ref = (oi, od, oc)
                for i in (0, len(qs2) - 1):
                    v, h = oracle.query_leaf(qs2[i])
                    want, _ = O.search(oracle, forest, v, h, count2, sk2, 0, cand)
                    assert list(oi[i, : oc[i]]) == [a for a, _ in want], desc + f" search2 q={i} count={count2} sk={sk2}"
                    T.assert_bit_equal(list(od[i, : oc[i]]), [b for _, b in want], desc + " search2 dists")
            else:
                same = np.array_equal(oc, ref[2]) and np.array_equal(oi, ref[0]) and np.array_equal(od.view(np.uint32), ref[1].view(np.uint32))
                if not same:  # which side left the oracle, and for which query
                    bad = [i for i in range(len(qs2)) if oc[i] != ref[2][i] or not np.array_equal(oi[i], ref[0][i])
                           or not np.array_equal(od[i].view(np.uint32), ref[1][i].view(np.uint32))]
                    i = bad[0]
                    v, h = oracle.query_leaf(qs2[i])
                    want, _ = O.search(oracle, forest, v, h, count2, sk2, 0, cand)
                    w_ids = [a for a, _ in want]
                    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                    np.savez(os.path.join(ROOT, "gpurun_out", "fuzz_fail.npz"), metric=metric, vecs=vecs, ids=ids, roots=forest.roots,
                             nodes=forest.nodes, normals=forest.normals, normal_stride=forest.normal_stride, vec_off=forest._vec_off,
                             hdr_off=forest._hdr_off, descendants=forest.descendants, query=qs2[i], count=count2, sk=sk2,
                             first_mode=ref[0][i, :ref[2][i]], this_mode=oi[i, :oc[i]])
                    raise AssertionError(desc + f" wave={wave} tiles={tiles} count={count2} sk={sk2} filter={None if cand is None else len(cand)} "
                                         f"sa={split_after} queries {bad}: q={i} oracle {w_ids[:8]} this {list(oi[i, :oc[i]])[:8]} "
                                         f"first mode {list(ref[0][i, :ref[2][i]])[:8]}")
    return desc


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    t0, it = time.time(), 0
    while time.time() - t0 < seconds:
        desc = one(rng, it)
        it += 1
        if it % 20 == 0:
            print(f"[{time.time() - t0:6.1f}s] {it} iterations ok; last: {desc}", flush=True)
    print(f"fuzz ok: {it} random configurations, seed {seed}")


if __name__ == "__main__":
    main()
