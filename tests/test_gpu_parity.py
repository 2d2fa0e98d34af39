"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bit-exact for everything: integer / index work by nature, f32 because the kernels reproduce the
reference's AVX+FMA (dims >= 32), SSE (16..31) and scalar (< 16) summation order.  The tolerance the
north star allows for f32 distances (1e-5 relative) is therefore asserted as *zero* ulps here.
Run with:  gpurun -- python -m pytest tests -m gpu -x -q
"""
import os

import numpy as np
import pytest

from conftest import hex_f32

pytestmark = pytest.mark.gpu

D = None  # arroy_amd.distances, imported lazily so collection works without the .so
O = None


@pytest.fixture(scope="module", autouse=True)
def _imports():
    global D, O
    import arroy_amd
    from arroy_amd import distances
    from oracle import oracle
    assert arroy_amd.device_count() >= 1, "no GPU visible: these tests must run on an MI355X"
    D, O = distances, oracle


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(a, b, what=""):
    a, b = np.asarray(a, dtype=np.float32), np.asarray(b, dtype=np.float32)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    bad = np.nonzero(bits(a) != bits(b))[0]
    assert bad.size == 0, f"{what}: {bad.size} of {a.size} differ, first at {bad[:5]}: {a[bad[:5]]} vs {b[bad[:5]]}"


def make_data(metric_cls, n, dims, seed, ids=None, scale=1.0, dist=None):
    """dist: None = numpy's N(0,1) * scale, or a distribution of include/arroy_hip_policy.h (the structured ones: clustered
    rows with exact duplicates, rows near a 32-dimensional subspace), scaled the same way."""
    from arroy_amd import Dataset
    rng = np.random.default_rng(seed)
    if dist is None:
        vecs = (rng.standard_normal((n, dims)) * scale).astype(np.float32)
    else:
        vecs = (O.synth(seed, dist, n, dims).astype(np.float64) * scale).astype(np.float32)
    # a few exact duplicates and zeros: ties and degenerate norms
    if n > 10:
        vecs[3] = vecs[1]
        vecs[n // 2] = vecs[1]
        vecs[5] = 0.0
    ids = np.arange(n, dtype=np.uint32) if ids is None else np.asarray(ids, dtype=np.uint32)
    ds = Dataset(metric_cls, dims, n)
    half = n // 2
    ds.upload_vectors(ids[:half], vecs[:half])  # two calls: chunked append
    ds.upload_vectors(ids[half:], vecs[half:])
    oracle = O.Data(metric_cls.metric, vecs, ids=None if np.array_equal(ids, np.arange(n)) else ids)
    if metric_cls.metric == 3:  # DotProduct: preprocess on both sides
        m_gpu = ds.preprocess_dot()
        m_cpu = oracle.preprocess_dot()
        assert bits(m_gpu) == bits(m_cpu)
    ds.finalize()
    return ds, oracle, vecs, ids


ALL_METRICS = [0, 1, 2, 3, 4, 5, 6]
DIMS = [3, 17, 30, 32, 70, 96, 128, 768]


# ---- golden vectors of the reference, through the GPU (src/tests/upgrade.rs:58-67,116-128) ------------

@pytest.mark.parametrize("name", ["large_v0_6", "smol_v0_6"])
def test_reference_golden_nns_on_gpu(golden, name):
    from arroy_amd import Dataset
    g = golden[name]
    vecs = np.stack([hex_f32(h) for h in g["vectors_hex"]])
    hdrs = [bytes.fromhex(h) for h in g["headers_hex"]]
    # the stored LMDB record layout [tag][header][vector] (src/node.rs:224-228), pointers misaligned
    records = [b"\x00" + hdrs[i] + vecs[i].tobytes() for i in range(len(hdrs))]
    ds = Dataset(D.Euclidean, g["dims"], len(records))
    ds.upload_records(g["ids"], records)
    ds.finalize()
    ids, dists = ds.rerank(g["count"], query=np.array(g["query"], dtype=np.float32))
    from test_oracle_golden import rust_display_f32
    assert [[int(i), rust_display_f32(d)] for i, d in zip(ids, dists)] == g["expected"]
    assert_bit_equal(ds.item_vector(g["ids"][0]), vecs[0])


# ---- batched distances (src/reader.rs:381-391) ----------------------------------------------------------

@pytest.mark.parametrize("metric", ALL_METRICS)
@pytest.mark.parametrize("dims", DIMS)
def test_distances_bit_exact(metric, dims):
    cls = D.BY_METRIC[metric]
    n = 700
    ds, oracle, vecs, ids = make_data(cls, n, dims, seed=100 + metric * 31 + dims)
    rng = np.random.default_rng(dims)
    q = rng.standard_normal(dims).astype(np.float32)
    qv, qh = oracle.query_leaf(q)
    assert_bit_equal(ds.distances(query=q), oracle.distances(qv, qh), f"scan by_vector m={metric} d={dims}")
    qv, qh = oracle.item_leaf(7)
    assert_bit_equal(ds.distances(item=7), oracle.distances(qv, qh), "scan by_item")
    sub = np.sort(rng.choice(n, 123, replace=False)).astype(np.uint32)
    assert_bit_equal(ds.distances(item=7, ids=sub), oracle.distances(qv, qh, rows=sub), "gather")
    d_self = ds.distances(item=9, ids=[9])
    if metric in (0, 1, 4, 5):
        assert d_self[0] == 0.0


@pytest.mark.parametrize("metric", ALL_METRICS)
def test_sparse_item_ids_and_lut(metric):
    cls = D.BY_METRIC[metric]
    n, dims = 300, 40
    ids = np.sort(np.random.default_rng(1).choice(5000, n, replace=False)).astype(np.uint32)
    ds, oracle, vecs, _ = make_data(cls, n, dims, seed=5, ids=ids)
    qv, qh = oracle.item_leaf(11)
    rows = np.arange(0, n, 3, dtype=np.uint32)
    got = ds.distances(item=int(ids[11]), ids=ids[rows])
    assert_bit_equal(got, oracle.distances(qv, qh, rows=rows))
    oi, od = ds.rerank(10, item=int(ids[11]))
    ei, ed = oracle.rerank(qv, qh, None, 10)
    assert list(oi) == list(ei)
    assert_bit_equal(od, ed)
    from arroy_amd import MissingKey
    with pytest.raises(MissingKey):
        missing = int(np.setdiff1d(np.arange(5000), ids)[0])
        ds.distances(item=int(ids[0]), ids=[missing])


def test_largest_item_ids_are_ordinary_ids():
    """Item ids u32::MAX - 1 and u32::MAX (src/tests/writer.rs:141-179) through the gather, the re-rank, the forest,
    the device search and its candidate filter."""
    n, dims = 300, 40
    ids = np.concatenate([np.arange(0, 2 * (n - 2), 2), [2**32 - 2, 2**32 - 1]]).astype(np.uint32)
    ds, oracle, vecs, ids = make_data(D.Euclidean, n, dims, seed=77, ids=ids)
    q = vecs[-1] + np.float32(0.01)
    qv, qh = oracle.query_leaf(q)
    assert_bit_equal(ds.distances(query=q, ids=ids[-5:]), oracle.distances(qv, qh, np.arange(n - 5, n, dtype=np.uint32)))
    oi, od = ds.rerank(3, query=q)
    ei, ed = oracle.rerank(qv, qh, None, 3)
    assert list(oi) == [int(x) for x in ei] and int(oi[0]) == 2**32 - 1
    assert_bit_equal(od, ed)
    forest = ds.build_forest([5, 6], split_after=16)
    check_forest_valid(forest, n, ids=ids)
    index = ds.create_index(forest)
    for cand in (None, [2**32 - 1, 2**32 - 2, 4], [2**32 - 2]):
        got = index.search(3, queries=q[None, :], search_k=2**62, candidates=cand)[0]
        want, _ = O.search(oracle, forest, qv, qh, 3, 2**62, 0, cand)
        assert [i for i, _ in got] == [i for i, _ in want]
        assert_bit_equal([d for _, d in got], [d for _, d in want])


def test_very_sparse_ids_use_binary_search():
    n, dims = 64, 32
    ids = (np.arange(n, dtype=np.uint64) * 50_000_000 + 7).astype(np.uint32)
    ds, oracle, vecs, _ = make_data(D.Euclidean, n, dims, seed=9, ids=ids)
    qv, qh = oracle.item_leaf(3)
    assert_bit_equal(ds.distances(item=int(ids[3]), ids=ids[::2]), oracle.distances(qv, qh, rows=np.arange(0, n, 2)))


# ---- re-rank + top-k (src/reader.rs:376-400, 607-640) ---------------------------------------------------

@pytest.mark.parametrize("metric", ALL_METRICS)
@pytest.mark.parametrize("n,k", [(50, 10), (5000, 1), (5000, 100), (9000, 2048), (9000, 3000), (300, 1000)])
def test_rerank_matches_reference_top_k(metric, n, k):
    cls = D.BY_METRIC[metric]
    dims = 64
    ds, oracle, vecs, ids = make_data(cls, n, dims, seed=n + k + metric, scale=1.0 if metric < 4 else 1.0)
    rng = np.random.default_rng(k)
    q = rng.standard_normal(dims).astype(np.float32)
    qv, qh = oracle.query_leaf(q)
    cand = np.sort(rng.choice(n, max(1, (2 * n) // 3), replace=False)).astype(np.uint32)
    for rows in (None, cand):
        oi, od = ds.rerank(k, query=q, sorted_ids=rows)
        ei, ed = oracle.rerank(qv, qh, rows, k)
        assert list(oi) == list(ei), f"ids differ m={metric} n={n} k={k}"
        assert_bit_equal(od, ed)


def test_rerank_ties_break_by_item_id():
    """BQ distances are small integers: massive ties, the order must be (distance, id)."""
    n, dims = 4000, 64
    ds, oracle, vecs, ids = make_data(D.BinaryQuantizedEuclidean, n, dims, seed=77)
    q = np.random.default_rng(2).standard_normal(dims).astype(np.float32)
    qv, qh = oracle.query_leaf(q)
    oi, od = ds.rerank(500, query=q)
    ei, ed = oracle.rerank(qv, qh, None, 500)
    assert list(oi) == list(ei)
    assert_bit_equal(od, ed)
    assert len(set(od.tolist())) < 40  # really tie-heavy


def test_rerank_rejects_unsorted_candidates_and_bad_dimensions():
    from arroy_amd import ArroyHipError, InvalidVecDimension
    ds, oracle, vecs, ids = make_data(D.Euclidean, 100, 32, seed=1)
    with pytest.raises(ArroyHipError):
        ds.rerank(5, item=0, sorted_ids=[5, 3, 9])
    with pytest.raises(InvalidVecDimension):
        ds.rerank(5, query=np.zeros(31, dtype=np.float32))
    with pytest.raises(InvalidVecDimension):
        ds2 = __import__("arroy_amd").Dataset(D.Euclidean, 32, 4)
        ds2.upload_records([0], [b"\x00" + b"\x00" * 4 + b"\x00" * 4 * 31])  # 31-dim record in a 32-dim index


def test_rerank_batch_equals_single_queries():
    n, dims, k = 3000, 96, 20
    ds, oracle, vecs, ids = make_data(D.Cosine, n, dims, seed=4)
    rng = np.random.default_rng(8)
    qs = rng.standard_normal((5, dims)).astype(np.float32)
    lists = [np.sort(rng.choice(n, m, replace=False)).astype(np.uint32) for m in (900, 15, 2500, 1, 0)]
    oi, od, oc = ds.rerank_batch(qs, lists, k)
    for i in range(5):
        if len(lists[i]) == 0:
            assert oc[i] == 0
            continue
        ei, ed = ds.rerank(k, query=qs[i], sorted_ids=lists[i])
        assert oc[i] == len(ei)
        assert list(oi[i, : oc[i]]) == list(ei)
        assert_bit_equal(od[i, : oc[i]], ed)


@pytest.mark.parametrize("nq", [48, 6])
@pytest.mark.parametrize("metric,dims,sparse_ids", [(0, 70, False), (2, 96, True), (3, 128, False), (2, 33, False)])
def test_rerank_batch_row_major_path(metric, dims, sparse_ids, nq):
    """A submission with >= 2 candidates per stored row is re-ranked row-major (pairs counting-sorted by row,
    batch.hip); same distances bit for bit as the query-major kernel, the single-query path and the oracle.
    nq=48: ~24 pairs per row (row-run kernel); nq=6: between 2 and 3 pairs per row (pair-per-slot kernel)."""
    cls = D.BY_METRIC[metric]
    n, k = 1500, 25
    ids = np.sort(np.random.default_rng(3).choice(40_000, n, replace=False)).astype(np.uint32) if sparse_ids else None
    ds, oracle, vecs, ids = make_data(cls, n, dims, seed=17 + metric, ids=ids)
    rng = np.random.default_rng(9)
    qs = rng.standard_normal((nq, dims)).astype(np.float32)
    sizes = [int(x) for x in rng.integers(1, n, nq)] if nq > 6 else [n // 3] * nq
    sizes[0], sizes[1] = n, 0  # every row, and an empty list
    lists = [np.sort(rng.choice(ids, m, replace=False)).astype(np.uint32) for m in sizes]
    assert sum(sizes) >= 2 * n  # the library's policy threshold for the row-major path
    assert (sum(sizes) >= 3 * n) == (nq > 6)  # ... and for the row-run kernel
    oi, od, oc = ds.rerank_batch(qs, lists, k)
    for i in range(nq):
        if sizes[i] == 0:
            assert oc[i] == 0
            continue
        ei, ed = ds.rerank(k, query=qs[i], sorted_ids=lists[i])
        assert oc[i] == len(ei) and list(oi[i, : oc[i]]) == list(ei)
        assert_bit_equal(od[i, : oc[i]], ed)
    for i in (0, 2, nq - 1):  # and against the CPU restatement
        q, qh = oracle.query_leaf(qs[i])
        rows = np.searchsorted(ids, lists[i]).astype(np.uint32)  # the oracle addresses rows, the C ABI item ids
        ci, cd = oracle.rerank(q, qh, rows, k)
        assert list(oi[i, : oc[i]]) == list(ci)
        assert_bit_equal(od[i, : oc[i]], cd)
    # a missing id is still reported through the row-major path
    bad = [l.copy() for l in lists]
    bad[5] = np.sort(np.append(bad[5][:-1], np.uint32(0xFFFFFFF0))).astype(np.uint32)
    with pytest.raises(Exception):
        ds.rerank_batch(qs, bad, k)


# ---- build side: margins / sides / create_split ------------------------------------------------------------

@pytest.mark.parametrize("metric", ALL_METRICS)
@pytest.mark.parametrize("dims", [3, 30, 64, 100, 768])
def test_create_split_and_sides_bit_exact(metric, dims):
    cls = D.BY_METRIC[metric]
    n = 400
    ds, oracle, vecs, ids = make_data(cls, n, dims, seed=dims * 7 + metric)
    rng = np.random.default_rng(metric)
    for trial in range(3):
        sample = rng.choice(n, 12, replace=trial == 2).astype(np.uint32)
        if sample[0] == sample[1]:
            sample[1] = (sample[0] + 1) % n
        nv, nh = ds.create_split(sample)
        env, enh = oracle.create_split(sample)
        assert nv.tobytes() == env.tobytes(), f"normal differs m={metric} d={dims} trial={trial}"
        assert_bit_equal(nh, enh, "normal header")
        sides, n_left, margins = ds.split_sides(nv, nh)
        es, enl, em = oracle.split_sides(env, enh)
        assert_bit_equal(margins, em, "margins")
        assert np.array_equal(sides, es) and n_left == enl
        sub = np.sort(rng.choice(n, 77, replace=False)).astype(np.uint32)
        sides, n_left, margins = ds.split_sides(nv, nh, sorted_ids=sub)
        es, enl, em = oracle.split_sides(env, enh, rows=sub)
        assert_bit_equal(margins, em)
        assert np.array_equal(sides, es) and n_left == enl


def test_side_of_signed_zero():
    """`is_sign_positive` (src/distance/mod.rs:103-110): +0.0 -> Right, -0.0 -> Left."""
    from arroy_amd import Dataset
    vecs = np.zeros((4, 32), dtype=np.float32)
    vecs[1, 0] = 1.0
    vecs[2, 0] = -1.0
    ds = Dataset(D.Euclidean, 32, 4)
    ds.upload_vectors(np.arange(4), vecs)
    ds.finalize()
    normal = np.zeros(32, dtype=np.float32)
    normal[0] = 1.0
    sides, n_left, margins = ds.split_sides(normal, [0.0])
    assert list(sides) == [1, 1, 0, 1] and n_left == 1
    sides, n_left, margins = ds.split_sides(normal, [-0.0])  # bias -0.0: -0.0 + +0.0 = +0.0
    assert list(sides) == [1, 1, 0, 1]
    normal[:] = 0.0
    normal[0] = -0.0
    sides, n_left, margins = ds.split_sides(normal, [-0.0])
    o = O.Data(0, vecs)
    es, enl, em = o.split_sides(normal, np.array([-0.0], dtype=np.float32))
    assert np.array_equal(sides, es)
    assert_bit_equal(margins, em)


# ---- whole forest: GPU level-synchronous build == CPU depth-first oracle ------------------------------------

def check_forest_valid(forest, n_items, ids=None):
    """`Reader::assert_validity` (src/reader.rs:509-589): every tree reaches every item exactly once."""
    expect = np.arange(n_items, dtype=np.uint32) if ids is None else np.sort(ids)
    for t in range(forest.n_trees):
        got = []
        stack = [int(forest.roots[t])]
        seen = set()
        while stack:
            i = stack.pop()
            assert i not in seen
            seen.add(i)
            nd = forest.nodes[i]
            if nd["kind"] == 1:
                d = forest.descendants_of(i)
                assert np.all(np.diff(d.astype(np.int64)) > 0)  # ascending ids inside a Descendants node
                got.append(d)
            else:
                stack += [int(nd["left"]), int(nd["right"])]
        got = np.sort(np.concatenate(got)) if got else np.zeros(0, np.uint32)
        assert np.array_equal(got, expect), f"tree {t} does not cover every item exactly once"


@pytest.mark.parametrize("metric", ALL_METRICS)
@pytest.mark.parametrize("n,dims,split_after", [(3000, 32, 32), (2500, 40, 100), (700, 8, 8), (6000, 96, 0)])
def test_forest_equals_oracle(metric, n, dims, split_after):
    cls = D.BY_METRIC[metric]
    ds, oracle, vecs, ids = make_data(cls, n, dims, seed=n + dims + metric)
    seeds = [42, 7, 2**63 + 5]
    forest = ds.build_forest(seeds, split_after=split_after)
    assert forest.n_trees == 3
    check_forest_valid(forest, n)
    for t, seed in enumerate(seeds):
        ref = oracle.build_tree(split_after, seed)
        assert forest.canonical(t) == ref.canonical(), f"tree {t} differs from the oracle (m={metric})"
    total = sum(oracle.build_tree(split_after, s).margin_evals for s in seeds)
    assert forest.stats["margin_evaluations"] == total


def test_baseline_config_1_10k_x_128_euclidean_10_trees():
    """BASELINE configs[0] (the reference's own CPU-runnable case, examples/build-tree-no-commit.rs shape): 10k x 128
    Euclidean, n_trees=10, split_after = dimensions.  The whole forest equals the oracle's node for node, and searches
    through it (default search_k, and exhaustive) equal the oracle's `nns_by_leaf`."""
    from arroy_amd import Dataset
    n, dims, trees = 10_000, 128, 10
    vecs = O.synth(42, 0, n, dims)  # i.i.d. uniform [0,1): what the reference's tests and examples use
    ds = Dataset(D.Euclidean, dims, n)
    ds.upload_vectors(np.arange(n, dtype=np.uint32), vecs)
    ds.finalize()
    oracle = O.Data(0, vecs)
    seeds = [int(x) for x in np.random.default_rng(42).integers(0, 2**63, trees)]
    forest = ds.build_forest(seeds)
    check_forest_valid(forest, n)
    for t, seed in enumerate(seeds):
        assert forest.canonical(t) == oracle.build_tree(0, seed).canonical(), f"tree {t} differs from the oracle"
    index = ds.create_index(forest)
    queries = O.synth(7, 0, 16, dims)
    for count, search_k in [(10, 0), (100, 0), (20, 2**62)]:
        got = index.search(count, queries=queries, search_k=search_k)
        for qi in range(len(queries)):
            qv, qh = oracle.query_leaf(queries[qi])
            want, _ = O.search(oracle, forest, qv, qh, count, search_k, 0, None)
            assert [i for i, _ in got[qi]] == [i for i, _ in want]
            assert_bit_equal([d for _, d in got[qi]], [d for _, d in want])


def test_forest_degenerate_inputs():
    from arroy_amd import Dataset
    # all items identical: every split fails 4 times, then the random fallback halves the node
    n, dims = 600, 32
    vecs = np.ones((n, dims), dtype=np.float32)
    ds = Dataset(D.Euclidean, dims, n)
    ds.upload_vectors(np.arange(n), vecs)
    ds.finalize()
    forest = ds.build_forest([1, 2], split_after=16)
    check_forest_valid(forest, n)
    assert forest.stats["dummy_normals"] > 0
    o = O.Data(0, vecs)
    for t, seed in enumerate([1, 2]):
        assert forest.canonical(t) == o.build_tree(16, seed).canonical()
    # fewer items than split_after: one Descendants root per tree (src/writer.rs:499-501)
    forest = ds.build_forest([5], split_after=1000)
    assert forest.nodes.size == 1 and forest.nodes[0]["kind"] == 1
    assert list(forest.descendants_of(0)) == list(range(n))
    # zero trees
    assert ds.build_forest([], split_after=16).n_trees == 0


def test_forest_with_sparse_ids_and_batches():
    n, dims = 2000, 32
    ids = np.sort(np.random.default_rng(3).choice(100000, n, replace=False)).astype(np.uint32)
    ds, oracle, vecs, _ = make_data(D.Cosine, n, dims, seed=12, ids=ids)
    seeds = list(range(10, 15))
    a = ds.build_forest(seeds, split_after=50)
    b = ds.build_forest(seeds, split_after=50, max_trees_in_flight=2)  # batching must not change anything
    check_forest_valid(a, n, ids)
    for t, seed in enumerate(seeds):
        assert a.canonical(t) == b.canonical(t) == oracle.build_tree(50, seed).canonical()


def test_build_can_be_cancelled():
    from arroy_amd import BuildCancelled
    ds, oracle, vecs, ids = make_data(D.Euclidean, 5000, 32, seed=1)
    with pytest.raises(BuildCancelled):
        ds.build_forest([1, 2, 3], split_after=8, cancel=lambda: True)


# ---- full-size properties (BASELINE config 2 shape: 1M x 768 cosine) ----------------------------------------

def test_full_size_properties_1m_x_768_cosine():
    from arroy_amd import Dataset
    n, dims = 1_000_000, 768
    ds = Dataset(D.Cosine, dims, n)
    ds.fill_synthetic(42, 1, n)
    ds.finalize()
    # (1) sampled rows are exactly the policy generator's rows; their headers the exact norms
    rows = np.array([0, 1, 12345, 999_999], dtype=np.uint32)
    for r in rows:
        assert_bit_equal(ds.item_vector(int(r)), O.synth(42, 1, 1, dims, first_item=int(r))[0])
    sample = np.sort(np.random.default_rng(0).choice(n, 1500, replace=False)).astype(np.uint32)
    vec_s = np.concatenate([O.synth(42, 1, 1, dims, first_item=int(r)) for r in sample])
    od = O.Data(2, vec_s)
    # (2) distances of the sampled rows are bit-exact against the oracle on the same rows
    q = O.synth(7, 1, 1, dims)[0]
    qv, qh = od.query_leaf(q)
    assert_bit_equal(ds.distances(query=q, ids=sample), od.distances(qv, qh))
    # (3) the top-k of the full scan is the sorted head of the full distance array under (distance, id)
    full = ds.distances(query=q)
    assert_bit_equal(full[sample], od.distances(qv, qh))
    oi, odist = ds.rerank(100, query=q)
    order = np.lexsort((np.arange(n), full))[:100]
    assert list(oi) == list(order)
    assert_bit_equal(odist, full[order])
    # (4) a split partitions: sides follow the margins' sign bit, counts add up
    nv, nh = ds.create_split(sample[:12])
    sides, n_left, margins = ds.split_sides(nv, nh)
    assert n_left == int((sides == 0).sum()) and np.array_equal(sides, (~np.signbit(margins)).astype(np.uint8))
    assert 0.05 * n < n_left < 0.95 * n
    # (5) a 2-tree forest over all items is structurally valid and deterministic
    f1 = ds.build_forest([1, 2])
    check_forest_valid(f1, n)
    f2 = ds.build_forest([1, 2])
    assert f1.normals.tobytes() == f2.normals.tobytes() and np.array_equal(f1.descendants, f2.descendants)
    assert all(f1.tree_stats(t)["descendants"] > n // 768 for t in range(2))


def subtree_items(forest, node):
    """Sorted item ids under `node` (vectorised over the Descendants blob: fine for millions of items)."""
    parts, stack = [], [int(node)]
    while stack:
        nd = forest.nodes[stack.pop()]
        if nd["kind"] == 1:
            parts.append(forest.descendants[int(nd["offset"]): int(nd["offset"]) + int(nd["count"])])
        else:
            stack += [int(nd["left"]), int(nd["right"])]
    return np.sort(np.concatenate(parts)) if parts else np.zeros(0, np.uint32)


def test_full_size_properties_10m_x_768_forest():
    """BASELINE configs[2] shape (10M x 768 cosine; 6 trees instead of 100 to keep the suite short): size-independent
    properties of the level-synchronous build at the size where the LDS / row-major / node-major margin kernels,
    the row-order node_of advance and the overlapped read-back are all in play."""
    from arroy_amd import Dataset
    n, dims, trees = 10_000_000, 768, 6
    ds = Dataset(D.Cosine, dims, n)
    ds.fill_synthetic(42, 1, n)
    ds.finalize()
    f = ds.build_forest(list(range(100, 100 + trees)))
    assert f.n_trees == trees
    st = f.stats
    assert st["margin_row_passes"] > 0 and st["levels"] >= 13
    nodes = f.nodes
    leaves = nodes[nodes["kind"] == 1]
    # (1) leaves hold at most split_after items, every tree partitions the item set (each id exactly once)
    assert int(leaves["count"].max()) <= dims and int(leaves["count"].sum()) == trees * n
    for t in range(trees):
        lt = leaves[leaves["tree"] == t]
        assert int(lt["count"].sum()) == n
    d = f.descendants
    assert d.size == trees * n
    for t in range(trees):  # trees own consecutive n-sized slices of the blob (final permutations)
        counts = np.bincount(d[t * n:(t + 1) * n], minlength=n)
        assert counts.min() == 1 and counts.max() == 1
    # (2) split nodes at every depth: the sides recomputed by the single-node API (`ah_split_sides`, the margin
    #     loop of src/writer.rs:1201-1207) reproduce exactly the children the forest recorded
    rng = np.random.default_rng(1)
    splits = np.flatnonzero(nodes["kind"] == 2)
    by_depth = {}
    for i in splits[rng.permutation(splits.size)[:4000]]:
        by_depth.setdefault(int(nodes[i]["depth"]), []).append(int(i))
    checked = 0
    for depth, cand in sorted(by_depth.items()):
        for i in cand[:2]:
            nd = nodes[i]
            if not nd["has_normal"]:
                continue
            left, right = subtree_items(f, nd["left"]), subtree_items(f, nd["right"])
            if left.size + right.size > 3_000_000:
                continue  # the top levels are covered by the smaller shapes
            ids = np.sort(np.concatenate([left, right]))
            hdr, vec = f.normal_of(i)
            sides, n_left, _ = ds.split_sides(vec, hdr, sorted_ids=ids, want_margins=False)
            assert n_left == left.size
            assert np.array_equal(ids[sides == 0], left) and np.array_equal(ids[sides == 1], right)
            checked += 1
    assert checked >= 12
    # (3) deterministic: the same seeds give the same bytes
    g = ds.build_forest(list(range(100, 100 + trees)))
    assert g.normals.tobytes() == f.normals.tobytes() and np.array_equal(g.descendants, d)
    assert np.array_equal(g.nodes, nodes)


def test_full_size_properties_1m_x_1536_dot_rerank():
    """BASELINE configs[3] shape: 1M x 1536 dot product after `preprocess`, search_k = 10 000 candidate re-ranks."""
    from arroy_amd import Dataset
    n, dims, k = 1_000_000, 1536, 100
    ds = Dataset(D.DotProduct, dims, n)
    ds.fill_synthetic(42, 1, n)
    max_norm = ds.preprocess_dot()
    assert 0.0 < float(max_norm) < float(np.sqrt(np.float32(dims)))  # uniform [-1,1) components
    ds.finalize()
    rng = np.random.default_rng(2)
    nq = 260  # > 2 candidates per stored row in total: the submission takes the row-major path
    lists = [np.sort(rng.choice(n, int(rng.integers(10_000, 11_536)), replace=False)).astype(np.uint32) for _ in range(nq)]
    qs = rng.standard_normal((nq, dims)).astype(np.float32)
    assert sum(len(l) for l in lists) >= 2 * n
    oi, od, oc = ds.rerank_batch(qs, lists, k)
    assert np.all(oc == k)
    for i in (0, 57, nq - 1):
        # the oracle on exactly these rows (vectors regenerated by the policy generator; `built_distance` of the dot
        # product does not read the headers, src/distance/dot_product.rs:52-56)
        vec = np.concatenate([O.synth(42, 1, 1, dims, first_item=int(r)) for r in lists[i]])
        sub = O.Data(3, vec)
        q, qh = sub.query_leaf(qs[i])
        ci, cd = sub.rerank(q, qh, None, k)
        assert list(oi[i]) == [int(lists[i][r]) for r in ci]
        assert_bit_equal(od[i], cd)
        # and the single-query path agrees with the batched one
        ei, ed = ds.rerank(k, query=qs[i], sorted_ids=lists[i])
        assert list(ei) == list(oi[i])
        assert_bit_equal(ed, od[i])


@pytest.mark.parametrize("metric", [4, 5, 6])
def test_full_size_properties_5m_x_768_bq_scan(metric):
    """BASELINE configs[4] shape: 5M x 768 1-bit vectors, Q=1 scan for the three binary-quantized metrics."""
    from arroy_amd import Dataset
    cls = D.BY_METRIC[metric]
    n, dims = 5_000_000, 768
    ds = Dataset(cls, dims, n)
    ds.fill_synthetic(42, 1, n)
    ds.finalize()
    q = O.synth(9, 1, 1, dims)[0]
    full = ds.distances(query=q)
    assert full.shape == (n,) and not np.isnan(full).any()
    sample = np.sort(np.random.default_rng(3).choice(n, 4000, replace=False)).astype(np.uint32)
    sample[:2] = (0, 1)
    sample[-1] = n - 1
    sample = np.unique(sample)
    vec = np.concatenate([O.synth(42, 1, 1, dims, first_item=int(r)) for r in sample])
    od = O.Data(metric, vec)
    qv, qh = od.query_leaf(q)
    assert_bit_equal(full[sample], od.distances(qv, qh))
    assert_bit_equal(ds.distances(query=q, ids=sample), full[sample])
    oi, odist = ds.rerank(100, query=q)
    order = np.lexsort((np.arange(n), full))[:100]  # (distance, id): 1-bit distances tie a lot
    assert list(oi) == list(order)
    # normalized_distance: bq_euclidean.rs:56-58, bq_manhattan.rs:55-57, bq_cosine: identity
    norm = {4: lambda x: x / np.float32(dims), 5: lambda x: np.maximum(x, np.float32(0)) / np.float32(dims), 6: lambda x: x}[metric]
    assert_bit_equal(odist, norm(full[order]).astype(np.float32))


@pytest.mark.parametrize("metric", [4, 5, 6])
@pytest.mark.parametrize("dims", [64, 1000, 4096, 5000])
def test_bq_scan_wide_and_narrow_rows(metric, dims):
    """1-bit rows from 8 bytes to > 512 bytes (cooperative kernel and the wide-row fallback)."""
    cls = D.BY_METRIC[metric]
    n = 333
    ds, oracle, vecs, ids = make_data(cls, n, dims, seed=dims + metric)
    q = np.random.default_rng(dims).standard_normal(dims).astype(np.float32)
    qv, qh = oracle.query_leaf(q)
    assert_bit_equal(ds.distances(query=q), oracle.distances(qv, qh))
    sub = np.arange(5, n, 7, dtype=np.uint32)
    assert_bit_equal(ds.distances(query=q, ids=sub), oracle.distances(qv, qh, rows=sub))


# ---- the reference's own insta snapshot, through the GPU ---------------------------------------------------

def _reference_order_build_on_gpu(ds, n_items, split_after, n_trees, seed, skip_u32):
    """`Writer::build` in the reference's order (depth-first, rand 0.8 StdRng, one rayon thread) where every
    create_split / margin loop is a C-ABI call into the HIP library: the "host RNG" mode of the integration
    (ah_create_split takes the sampled ids from the caller, exactly like `R: Rng` stays on the Rust side)."""
    rng0 = O.ChaCha12(seed)
    for _ in range(skip_u32):
        rng0.next_u32()
    rng1 = O.ChaCha12(rng0.gen_seed())
    task_seeds = [rng1.gen_seed() for _ in range(n_trees)]
    nodes, next_id = {}, [n_trees]

    def alloc():
        next_id[0] += 1
        return next_id[0] - 1

    def rec(ids, rng, node_id):
        if len(ids) <= split_after:
            node_id = alloc() if node_id is None else node_id
            nodes[node_id] = ("D", [int(i) for i in ids])
            return node_id
        remaining = 3
        while True:
            a, b = rng.index_sample2(len(ids))
            sample = [ids[a], ids[b]] + [ids[rng.gen_range_inclusive(0, len(ids) - 1)] for _ in range(10)]
            nv, nh = ds.create_split(sample)
            sides, n_left, _ = ds.split_sides(nv, nh, sorted_ids=ids, want_margins=False)
            imb = O.lib().ao_split_imbalance(n_left, len(ids) - n_left)
            if imb < 0.95 or remaining == 0:
                break
            remaining -= 1
        if imb > 0.99:
            sides = np.array([0 if rng.gen_bool() else 1 for _ in ids], dtype=np.uint8)
            nv = None
        left = rec(ids[sides == 0], rng, None)
        right = rec(ids[sides == 1], rng, None)
        node_id = alloc() if node_id is None else node_id
        nodes[node_id] = ("S", left, right, "%.4f" % nh[0],
                          None if nv is None else ["%.4f" % x for x in nv.view(np.float32)])
        return node_id

    for root in reversed(range(n_trees)):  # tasks run last-in-first-out on the single rayon worker
        rec(np.arange(n_items, dtype=np.uint32), O.ChaCha12(task_seeds[root]), root)
    return nodes


def test_reference_insta_snapshot_through_the_gpu(golden):
    """src/tests/writer.rs:296-308: the 10-tree, 92-node snapshot of arroy's own test-suite, with every split
    computed by the HIP kernels (ids, children, bias, normal prefix and descendants of every node)."""
    from arroy_amd import Dataset
    g = golden["random_points_10_trees"]
    seed = bytes([42] * 32)
    rng = O.ChaCha12(seed)
    vecs = np.array([[rng.gen_f32() for _ in range(g["dims"])] for _ in range(g["n_items"])], dtype=np.float32)
    ds = Dataset(D.Euclidean, g["dims"], g["n_items"])
    ds.upload_vectors(np.arange(g["n_items"], dtype=np.uint32), vecs)
    ds.finalize()
    nodes = _reference_order_build_on_gpu(ds, g["n_items"], g["dims"], g["n_trees"], seed,
                                          skip_u32=g["n_items"] * g["dims"])
    assert sorted(nodes) == sorted(int(k) for k in g["trees"])
    for k, want in g["trees"].items():
        got = nodes[int(k)]
        if want["kind"] == "D":
            assert got == ("D", want["descendants"]), f"tree node {k}"
        else:
            assert (got[0], got[1], got[2], got[3]) == ("S", want["left"], want["right"], want["bias"]), f"node {k}"
            assert got[4][:10] == want["vector10"], f"normal of tree node {k}"


# ---- the whole search on device (src/reader.rs:317-401) vs the oracle's restatement --------------------------

@pytest.mark.parametrize("metric", ALL_METRICS)
def test_device_search_equals_oracle(metric):
    cls = D.BY_METRIC[metric]
    n, dims = 4000, 48
    ds, oracle, vecs, ids = make_data(cls, n, dims, seed=900 + metric)
    forest = ds.build_forest([11, 12, 13, 14, 15], split_after=40)
    index = ds.create_index(forest)
    rng = np.random.default_rng(metric)
    queries = rng.standard_normal((24, dims)).astype(np.float32)
    items = rng.choice(n, 24, replace=False).astype(np.uint32)
    for count, search_k, over, cand in [(10, 0, 0, None), (5, 300, 0, None), (20, 2500, 2, None), (7, 1, 0, None),
                                        (10, 2**62, 0, None), (10, 800, 0, range(0, n, 3)), (3, 50, 0, [])]:
        got_v = index.search(count, queries=queries, search_k=search_k, oversampling=over, candidates=cand)
        got_i = index.search(count, items=items, search_k=search_k, oversampling=over, candidates=cand)
        for qi in range(len(queries)):
            qv, qh = oracle.query_leaf(queries[qi])
            want, _ = O.search(oracle, forest, qv, qh, count, search_k, over, cand)
            assert [i for i, _ in got_v[qi]] == [i for i, _ in want], f"by_vector q={qi} {count},{search_k},{over}"
            assert_bit_equal([d for _, d in got_v[qi]], [d for _, d in want])
            qv, qh = oracle.item_leaf(int(items[qi]))
            want, _ = O.search(oracle, forest, qv, qh, count, search_k, over, cand)
            assert [i for i, _ in got_i[qi]] == [i for i, _ in want], f"by_item q={qi} {count},{search_k},{over}"
            assert_bit_equal([d for _, d in got_i[qi]], [d for _, d in want])


@pytest.mark.parametrize("bitmap", [1, 0])
def test_device_search_big_queue_and_big_candidate_sets(bitmap):
    """Tiny leaves (split_after=2) and a huge search_k: the queue overflows its LDS slot (re-run with the queue in
    global memory) and the candidate set exceeds the LDS sort (global bitonic + dedup path, or the LDS bitmap)."""
    from arroy_amd._lib import tuning
    n, dims = 30000, 32
    ds, oracle, vecs, ids = make_data(D.Euclidean, n, dims, seed=31)
    forest = ds.build_forest([1, 2, 3], split_after=2)
    index = ds.create_index(forest)
    queries = np.random.default_rng(1).standard_normal((3, dims)).astype(np.float32)
    for search_k in (4000, 40000, 2**62):
        with tuning(AH_SEARCH_BITMAP=bitmap):
            got = index.search(15, queries=queries, search_k=search_k)
        for qi in range(3):
            qv, qh = oracle.query_leaf(queries[qi])
            want, cand = O.search(oracle, forest, qv, qh, 15, search_k)
            assert [i for i, _ in got[qi]] == [i for i, _ in want]
            assert_bit_equal([d for _, d in got[qi]], [d for _, d in want])
    # exhaustive search_k == exact top-k
    exact_ids, exact_d = ds.rerank(15, query=queries[0])
    assert [i for i, _ in index.search(15, queries=queries[:1], search_k=2**62)[0]] == list(exact_ids)


@pytest.mark.parametrize("last_id", [39 * 1024 * 32 - 1, 39 * 1024 * 32, 2**32 - 1])
def test_device_search_sort_dedup_paths_agree(last_id):
    """nns.sort_unstable(); nns.dedup() (src/reader.rs:378-379) by the LDS bitmap (id space up to 39*1024*32 ids) and
    by the bitonic network: the same answers, on sparse ids whose largest value sits on either side of the limit."""
    from arroy_amd._lib import tuning
    n, dims = 20000, 32
    rng = np.random.default_rng(77)
    ids = np.sort(rng.choice(min(last_id, 3_000_000), n - 1, replace=False)).astype(np.uint32)
    ids = np.concatenate([ids, np.array([last_id], dtype=np.uint32)])
    ds, oracle, vecs, ids = make_data(D.Cosine, n, dims, seed=78, ids=ids)
    forest = ds.build_forest(list(range(20)), split_after=60)
    index = ds.create_index(forest)
    queries = rng.standard_normal((64, dims)).astype(np.float32)
    res = {}
    # 1: the LDS bitmap where the id space fits it (else the hash set of the candidates); 0: never the bitmap (the hash set,
    # or the bitonic sort when 0xFFFFFFFF is a stored id); 2: sort + dedup + row-major re-rank
    for mode in (1, 0, 2):
        with tuning(AH_SEARCH_BITMAP=1 if mode else 0, AH_SEARCH_TILES=0 if mode == 2 else 1):
            res[mode] = [index.search(25, queries=queries, search_k=sk, raw=True) for sk in (0, 3000, 2**62)]
    for mode in (0, 2):
        for a, b in zip(res[1], res[mode]):
            assert np.array_equal(a[2], b[2]) and np.array_equal(a[0], b[0]), mode
            assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)), mode
    oi, od, oc = res[1][1]
    for qi in (0, 63):
        qv, qh = oracle.query_leaf(queries[qi])
        want, _ = O.search(oracle, forest, qv, qh, 25, 3000)
        assert list(oi[qi, :oc[qi]]) == [i for i, _ in want]
        assert_bit_equal(list(od[qi, :oc[qi]]), [d for _, d in want])


@pytest.mark.parametrize("metric,dims", [(0, 100), (2, 64), (3, 96)])
def test_device_search_leaf_tiles_equal_the_sorted_path(metric, dims):
    """ah_search_batch by leaf tiles (rows of a leaf x the queries that reached it, duplicates flagged, order by
    (distance, id)) and by sort + dedup + row-major re-rank: the same bits.  Queries in clusters of 1..7 near copies so
    that leaves are shared by 1, 2, 3, 4 and more queries; equal vectors under different ids for the ties."""
    from arroy_amd._lib import tuning
    cls = D.BY_METRIC[metric]
    n = 30000
    ds, oracle, vecs, ids = make_data(cls, n, dims, seed=400 + metric)
    forest = ds.build_forest(list(range(100, 112)), split_after=150)
    index = ds.create_index(forest)
    rng = np.random.default_rng(5)
    queries = []
    for c in range(40):
        base = vecs[rng.integers(n)]
        for _ in range(1 + c % 7):
            queries.append(base + rng.standard_normal(dims).astype(np.float32) * 1e-3)
    queries.append(vecs[1])  # three items hold this vector: a tie on the smallest distance
    queries = np.asarray(queries, dtype=np.float32)
    keep_third = rng.choice(n, n // 3, replace=False)
    for count, sk, cand in [(10, 0, None), (100, 3000, None), (1000, 6000, None), (1500, 20000, None),
                            (100, 3000, keep_third), (100, 3000, range(0, n, 2)), (40, 900, range(0, n, 50)), (10, 500, [7])]:
        res = {}
        for t in (1, 0, 2):  # 2: the tiles after the descent of one octet per query
            with tuning(AH_SEARCH_TILES=min(t, 1), AH_SEARCH_WAVE=1 if t < 2 else 0):
                res[t] = index.search(count, queries=queries, search_k=sk, candidates=cand, raw=True)
        for t in (0, 2):
            assert np.array_equal(res[1][2], res[t][2]) and np.array_equal(res[1][0], res[t][0]), (count, sk, t)
            assert np.array_equal(res[1][1].view(np.uint32), res[t][1].view(np.uint32)), (count, sk, t)
        if cand is not None:  # candidate filter: against the oracle too
            oi, od, oc = res[1]
            for qi in (0, 5, len(queries) - 1):
                qv, qh = oracle.query_leaf(queries[qi])
                want, _ = O.search(oracle, forest, qv, qh, count, sk, 0, cand)
                assert list(oi[qi, :oc[qi]]) == [i for i, _ in want], (count, sk, qi)
                assert_bit_equal(list(od[qi, :oc[qi]]), [d for _, d in want])
    res[1] = index.search(1500, queries=queries, search_k=20000, raw=True)
    oi, od, oc = res[1]
    for qi in (0, len(queries) - 1):
        qv, qh = oracle.query_leaf(queries[qi])
        want, _ = O.search(oracle, forest, qv, qh, 1500, 20000)
        assert list(oi[qi, :oc[qi]]) == [i for i, _ in want]
        assert_bit_equal(list(od[qi, :oc[qi]]), [d for _, d in want])


def test_device_search_leaf_tiles_leave_non_finite_distances_to_the_sorted_path():
    """Squared distances that overflow to +inf: reader.rs:611-621 treats such candidates by their position in the sorted
    list, so the submission is redone by the sorted path; the answers are those of the oracle either way."""
    from arroy_amd import Dataset
    from arroy_amd._lib import tuning
    n, dims = 3000, 32
    rng = np.random.default_rng(9)
    vecs = rng.standard_normal((n, dims)).astype(np.float32)
    vecs[::7] *= np.float32(3e19)
    ds = Dataset(D.Euclidean, dims, n)
    ds.upload_vectors(np.arange(n, dtype=np.uint32), vecs)
    ds.finalize()
    oracle = O.Data(0, vecs)
    forest = ds.build_forest([1, 2, 3, 4], split_after=50)
    index = ds.create_index(forest)
    queries = vecs[[0, 7, 8, 14]]
    res = {}
    for t in (1, 0):
        with tuning(AH_SEARCH_TILES=t):
            res[t] = index.search(40, queries=queries, search_k=2000, raw=True)
    assert np.array_equal(res[1][0], res[0][0]) and np.array_equal(res[1][1].view(np.uint32), res[0][1].view(np.uint32))
    assert np.isinf(res[1][1]).any()
    for qi in range(4):
        qv, qh = oracle.query_leaf(queries[qi])
        want, _ = O.search(oracle, forest, qv, qh, 40, 2000)
        assert list(res[1][0][qi, :res[1][2][qi]]) == [i for i, _ in want]


@pytest.mark.parametrize("n_trees,split_after,search_k", [(1, 40, 500), (3, 8, 3000), (8, 200, 0), (20, 30, 2000),
                                                         (37, 100, 2**62)])
def test_device_search_wave_descent_takes_the_same_candidates(n_trees, split_after, search_k):
    """k_descend_wave (8 octets search the trees of one query side by side and settle the leaves by key) against
    k_descend (the sequential queue) and the oracle: fewer trees than octets, leaves small enough to overflow the leaf
    lists (those queries go back to k_descend), an exhaustive search_k."""
    from arroy_amd._lib import tuning
    n, dims = 12000, 48
    ds, oracle, vecs, ids = make_data(D.Euclidean, n, dims, seed=1000 + n_trees)
    forest = ds.build_forest(list(range(n_trees)), split_after=split_after)
    index = ds.create_index(forest)
    rng = np.random.default_rng(n_trees)
    queries = np.concatenate([vecs[rng.integers(n, size=20)], rng.standard_normal((20, dims)).astype(np.float32)])
    res = {}
    for w in (1, 0):
        for t in (1, 0):
            with tuning(AH_SEARCH_WAVE=w, AH_SEARCH_TILES=t):
                res[w, t] = index.search(30, queries=queries, search_k=search_k, raw=True)
    for key in res:
        assert np.array_equal(res[key][2], res[0, 0][2]) and np.array_equal(res[key][0], res[0, 0][0]), key
        assert np.array_equal(res[key][1].view(np.uint32), res[0, 0][1].view(np.uint32)), key
    oi, od, oc = res[1, 1]
    for qi in (0, 19, 20, 39):
        qv, qh = oracle.query_leaf(queries[qi])
        want, _ = O.search(oracle, forest, qv, qh, 30, search_k)
        assert list(oi[qi, :oc[qi]]) == [i for i, _ in want]
        assert_bit_equal(list(od[qi, :oc[qi]]), [d for _, d in want])


def test_device_search_wave_descent_with_equal_keys():
    """Degenerate rows (every split fails: `normal: None`, margin 0 for both children) make every key of a tree equal;
    which leaf pops first is then decided by node ids and by which parent was popped first, so those queries are left
    to the sequential queue and still match the oracle."""
    from arroy_amd import Dataset
    from arroy_amd._lib import tuning
    n, dims = 800, 32
    vecs = np.ones((n, dims), dtype=np.float32)
    vecs[:40] = np.random.default_rng(3).standard_normal((40, dims)).astype(np.float32)
    ds = Dataset(D.Euclidean, dims, n)
    ds.upload_vectors(np.arange(n, dtype=np.uint32), vecs)
    ds.finalize()
    oracle = O.Data(0, vecs)
    forest = ds.build_forest([9, 10, 11], split_after=16)
    assert forest.stats["dummy_normals"] > 0
    index = ds.create_index(forest)
    queries = vecs[[0, 1, 100, 799]]
    for sk in (50, 200, 700):
        res = {}
        for w in (1, 0):
            with tuning(AH_SEARCH_WAVE=w):
                res[w] = index.search(20, queries=queries, search_k=sk, raw=True)
        assert np.array_equal(res[1][0], res[0][0]) and np.array_equal(res[1][1].view(np.uint32), res[0][1].view(np.uint32))
        for qi in range(4):
            qv, qh = oracle.query_leaf(queries[qi])
            want, _ = O.search(oracle, forest, qv, qh, 20, sk)
            assert list(res[1][0][qi, :res[1][2][qi]]) == [i for i, _ in want]


@pytest.mark.parametrize("metric,dims", [(4, 31), (6, 17), (5, 64)])
def test_device_search_wave_descent_equal_keys_between_trees(metric, dims):
    """Binary-quantized margins are small integers (plus a bias), so leaves of different trees often carry the same key.
    Which of them the sequential loop takes depends on their pop order (node ids): e.g. leaves of 83 and 227 ids with
    one key and search_k = 90 -- the one popped first may end the search alone.  k_descend_wave must leave such queries
    to the sequential queue: every query of 300, for several search_k, against k_descend."""
    from arroy_amd._lib import tuning
    cls = D.BY_METRIC[metric]
    n = 2049
    ds, oracle, vecs, ids = make_data(cls, n, dims, seed=4242 + metric)
    forest = ds.build_forest(list(range(17)), split_after=300)
    index = ds.create_index(forest)
    rng = np.random.default_rng(12)
    queries = np.concatenate([vecs[rng.integers(n, size=150)], rng.standard_normal((150, dims)).astype(np.float32)])
    for count, sk in [(7, 30), (3, 1), (20, 150), (10, 400)]:
        res = {}
        for w in (1, 0):
            with tuning(AH_SEARCH_WAVE=w):
                res[w] = index.search(count, queries=queries, search_k=sk, raw=True)
        assert np.array_equal(res[1][2], res[0][2]) and np.array_equal(res[1][0], res[0][0]), (count, sk)
        assert np.array_equal(res[1][1].view(np.uint32), res[0][1].view(np.uint32)), (count, sk)
    oi, od, oc = res[0]
    for qi in (0, 299):
        qv, qh = oracle.query_leaf(queries[qi])
        want, _ = O.search(oracle, forest, qv, qh, 10, 400)
        assert list(oi[qi, :oc[qi]]) == [i for i, _ in want]


def _equal_keys_fixture():
    """tests/golden/search_equal_keys_bq.npz: a BinaryQuantizedEuclidean forest (17 trees, 2049 items, 31 dimensions) and a
    query whose two best leaves (83 and 227 ids, in trees of different octets) carry the same key 5.0 while search_k = 90:
    the sequential queue pops the 227-id leaf first (bigger node id) and stops there.  Found by scripts/fuzz_gpu.py; the
    expected answer in the file is the oracle's (tests/test_oracle_golden.py checks that on the CPU)."""
    import types
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "search_equal_keys_bq.npz"))
    forest = types.SimpleNamespace(n_trees=len(g["roots"]), roots=g["roots"], nodes=g["nodes"], normals=g["normals"],
                                   normal_stride=int(g["normal_stride"]), _vec_off=int(g["vec_off"]), _hdr_off=int(g["hdr_off"]),
                                   descendants=g["descendants"])
    return g, forest


def test_device_search_wave_descent_equal_keys_across_the_cut():
    from arroy_amd import Dataset
    from arroy_amd._lib import tuning
    from arroy_amd.dataset import Index
    g, forest = _equal_keys_fixture()
    cls = D.BY_METRIC[int(g["metric"])]
    ds = Dataset(cls, g["vecs"].shape[1], len(g["vecs"]))
    ds.upload_vectors(g["ids"], g["vecs"])
    ds.finalize()
    index = Index(ds, None, view=O.forest_view(forest))
    queries = np.stack([g["query"]] * 9)  # more than one block of the octet-per-query kernel
    for wave in (1, 0):
        with tuning(AH_SEARCH_WAVE=wave):
            oi, od, oc = index.search(int(g["count"]), queries=queries, search_k=int(g["sk"]), raw=True)
        for qi in range(len(queries)):
            assert list(oi[qi, :oc[qi]]) == list(g["want_ids"]), f"wave={wave}"
            assert_bit_equal(list(od[qi, :oc[qi]]), list(g["want_dists"]))


# ---- incremental paths: routing through existing trees + sub-tree builds (src/writer.rs:660-739, 1398-1459) ----

@pytest.mark.parametrize("metric", ALL_METRICS)
def test_route_items_equals_oracle_and_finds_the_items_own_leaf(metric):
    cls = D.BY_METRIC[metric]
    n, dims = 3000, 40
    ds, oracle, vecs, ids = make_data(cls, n, dims, seed=50 + metric)
    seeds = [3, 4, 5]
    forest = ds.build_forest(seeds, split_after=64)
    index = ds.create_index(forest)
    rows = np.sort(np.random.default_rng(metric).choice(n, 500, replace=False)).astype(np.uint32)
    got = index.route_items(rows, seeds)
    want = O.route_items(oracle, forest, rows, seeds)
    assert np.array_equal(got, want)
    # every reached node is a Descendants node; without dummy normals an item already in the tree reaches the
    # leaf that holds it (the same side() decisions as during the build)
    for t in range(3):
        for i, r in enumerate(rows):
            nd = forest.nodes[int(got[t, i])]
            assert nd["kind"] == 1
            if forest.stats["dummy_normals"] == 0:
                assert int(r) in set(int(x) for x in forest.descendants_of(int(got[t, i])))


def test_route_items_through_dummy_normals():
    """`normal: None` nodes route by the policy coin (degenerate data: all splits fail)."""
    from arroy_amd import Dataset
    n, dims = 800, 32
    vecs = np.ones((n, dims), dtype=np.float32)
    ds = Dataset(D.Euclidean, dims, n)
    ds.upload_vectors(np.arange(n), vecs)
    ds.finalize()
    forest = ds.build_forest([9], split_after=16)
    assert forest.stats["dummy_normals"] > 0
    index = ds.create_index(forest)
    o = O.Data(0, vecs)
    rows = np.arange(0, n, 3, dtype=np.uint32)
    assert np.array_equal(index.route_items(rows, [77]), O.route_items(o, forest, rows, [77]))


@pytest.mark.parametrize("metric", [0, 2, 3, 6])
def test_build_subtrees_equals_oracle(metric):
    cls = D.BY_METRIC[metric]
    n, dims = 5000, 32
    ds, oracle, vecs, ids = make_data(cls, n, dims, seed=70 + metric)
    rng = np.random.default_rng(metric)
    subsets = [np.sort(rng.choice(n, m, replace=False)).astype(np.uint32) for m in (1200, 40, 333, 2, 0, 2500)]
    seeds = [100 + i for i in range(len(subsets))]
    forest = ds.build_subtrees(subsets, seeds, split_after=48)
    assert forest.n_trees == len(subsets)
    for t, (sub, seed) in enumerate(zip(subsets, seeds)):
        ref = oracle.build_tree(48, seed, rows=sub)
        assert forest.canonical(t) == ref.canonical(), f"sub-tree {t} ({len(sub)} items) differs from the oracle"
    with pytest.raises(__import__("arroy_amd").ArroyHipError):
        ds.build_subtrees([[5, 3, 9]], [1], split_after=1)  # not ascending


def test_index_from_caller_owned_arrays():
    """`ah_index_create_from_view`: tree nodes that never were an ah_forest (here: built by the CPU oracle, records in
    the oracle's [header][vector] layout) are searchable on device; garbage views are rejected, not dereferenced."""
    from arroy_amd import ArroyHipError, Index
    n, dims = 2500, 40
    ds, oracle, vecs, ids = make_data(D.Euclidean, n, dims, seed=8)
    tree = oracle.build_tree(50, 99).as_forest(oracle)
    view = O.forest_view(tree)
    index = Index(ds, None, view=view)
    queries = np.random.default_rng(3).standard_normal((10, dims)).astype(np.float32)
    got = index.search(10, queries=queries, search_k=400)
    for qi in range(10):
        qv, qh = oracle.query_leaf(queries[qi])
        want, _ = O.search(oracle, tree, qv, qh, 10, 400)
        assert [i for i, _ in got[qi]] == [i for i, _ in want]
    tree.nodes["left"][int(tree.roots[0])] = 10**6  # child index out of range
    with pytest.raises(ArroyHipError):
        Index(ds, None, view=O.forest_view(tree))


# ---- staging, concurrency -----------------------------------------------------------------------------------

def test_staging_many_chunks_records_and_vectors_agree():
    """10k x 768 cosine: several 8 MiB staging chunks through both upload paths (LMDB-style records with the stored
    header vs raw vectors with the header computed on device) give the same dataset."""
    from arroy_amd import Dataset
    n, dims = 10_000, 768
    vecs = O.synth(3, 1, n, dims)
    od = O.Data(O.COSINE, vecs)
    a = Dataset(D.Cosine, dims, n)
    a.upload_vectors(np.arange(n, dtype=np.uint32), vecs)
    a.finalize()
    assert_bit_equal(a.read_headers().ravel(), od.headers.ravel(), "norms computed on device == oracle new_header")
    records = [b"\x00" + od.headers[i].tobytes() + vecs[i].tobytes() for i in range(n)]
    b = Dataset(D.Cosine, dims, n)
    b.upload_records(np.arange(n, dtype=np.uint32), records)
    b.finalize()
    q = O.synth(9, 1, 1, dims)[0]
    assert_bit_equal(a.distances(query=q), b.distances(query=q))
    assert_bit_equal(b.item_vector(n - 1), vecs[n - 1])
    qv, qh = od.query_leaf(q)
    assert_bit_equal(a.distances(query=q), od.distances(qv, qh))


def test_concurrent_queries_from_many_threads():
    """`Reader` is Sync: many host threads share one finalized dataset (each call leases its own stream + scratch)."""
    import threading
    n, dims = 20_000, 96
    ds, oracle, vecs, ids = make_data(D.Cosine, n, dims, seed=4)
    queries = np.random.default_rng(0).standard_normal((16, dims)).astype(np.float32)
    expect = [ds.rerank(25, query=q) for q in queries]
    errors = []

    def worker(tid):
        try:
            for rep in range(30):
                qi = (tid * 7 + rep) % len(queries)
                i, d = ds.rerank(25, query=queries[qi])
                if not (np.array_equal(i, expect[qi][0]) and d.tobytes() == expect[qi][1].tobytes()):
                    errors.append((tid, rep))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]


# ---- numeric corner cases ---------------------------------------------------------------------------------------

@pytest.mark.parametrize("metric", [0, 1, 2, 3])
@pytest.mark.parametrize("scale", [1e-22, 1e-12, 1e12, 3e18])
def test_subnormal_and_huge_magnitudes_stay_bit_exact(metric, scale):
    """Products in the subnormal range (scale^2 ~ 1e-44) and sums that overflow to +inf must match the reference's
    IEEE behaviour: no flush-to-zero, correctly rounded sqrt / div, inf handled like the CPU.  (NaN *payloads* are
    outside the contract: x86 generates 0xFFC00000, the GPU 0x7FC00000; inputs here never produce NaN.)"""
    cls = D.BY_METRIC[metric]
    n, dims = 500, 96
    ds, oracle, vecs, ids = make_data(cls, n, dims, seed=17 + metric, scale=scale)
    q = (np.abs(np.random.default_rng(5).standard_normal(dims)) * scale).astype(np.float32)
    qv, qh = oracle.query_leaf(q)
    with np.errstate(all="ignore"):
        want = oracle.distances(qv, qh)
    got = ds.distances(query=q)
    finite_or_inf = ~np.isnan(want)
    assert finite_or_inf.sum() > n // 2
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert_bit_equal(got[finite_or_inf], want[finite_or_inf], f"scale={scale}")
    oi, od = ds.rerank(20, query=q)
    ei, ed = oracle.rerank(qv, qh, None, 20)
    assert list(oi) == list(ei)


@pytest.mark.parametrize("metric", [3, 4, 6])
def test_upload_records_with_stored_headers_for_dot_and_bq(metric):
    """`ImmutableLeafs` staging for the metrics whose stored header / codec is not trivial: DotProduct records carry
    {extra_dim, norm} written by `preprocess` (src/distance/dot_product.rs:119-165), 1-bit records carry packed words."""
    from arroy_amd import Dataset
    cls = D.BY_METRIC[metric]
    n, dims = 900, 70
    ds_a, oracle, vecs, ids = make_data(cls, n, dims, seed=33 + metric)  # upload_vectors (+ preprocess for dot)
    records = [b"\x00" + oracle.headers[i].tobytes() + oracle.codec[i].tobytes() for i in range(n)]
    ds_b = Dataset(cls, dims, n)
    if metric == 3:  # records nobody vouched for: the build must refuse them (freshly added items carry {0, 0} headers)
        ds_c = Dataset(cls, dims, n)
        ds_c.upload_records(np.arange(n, dtype=np.uint32), records)
        ds_c.finalize()
        from arroy_amd import _lib as ahlib
        with pytest.raises(ahlib.ArroyHipError) as e:
            ds_c.build_forest([5], split_after=30)
        assert e.value.status == 8  # AH_ERR_NEED_PREPROCESS
        ds_c.close()
    ds_b.upload_records(np.arange(n, dtype=np.uint32), records, preprocessed=True)  # headers of a built database
    ds_b.finalize()
    assert_bit_equal(ds_a.read_headers().ravel(), ds_b.read_headers().ravel())
    assert_bit_equal(ds_a.distances(item=3), ds_b.distances(item=3))
    fa, fb = ds_a.build_forest([5], split_after=30), ds_b.build_forest([5], split_after=30)
    assert fa.canonical(0) == fb.canonical(0) == oracle.build_tree(30, 5).canonical()
