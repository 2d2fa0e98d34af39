"""The reference's own reader/writer tests replayed through the arroy-surface mirror (arroy_amd/index.py) on the
GPU.  Only RNG-independent expectations are mirrored (the reference's tree *shape* snapshots depend on rand 0.8's
ChaCha12 stream)."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    import arroy_amd
    from arroy_amd import distances as D
    from arroy_amd import index as I
    assert arroy_amd.device_count() >= 1
    return D, I


def rng():
    return random.Random(42)  # the reference seeds StdRng with [42; 32] (src/tests/mod.rs:105-107)


def fmt(res):
    """NnsRes Display (src/tests/reader.rs:14-29): `id(n): distance(d)` per line."""
    def f(d):
        return str(int(d)) if float(d).is_integer() else repr(float(np.float32(d)))
    return [f"id({i}): distance({f(d)})" for i, d in res]


def test_open_db_with_wrong_dimension(api):
    """src/tests/reader.rs:45-60."""
    D, I = api
    db = I.Database(D.Euclidean)
    w = I.Writer(db, 0, 2)
    w.add_item(0, [0.0, 0.0])
    w.builder(rng()).n_trees(1).build()
    reader = I.Reader.open(db, 0)
    with pytest.raises(I.InvalidVecDimension) as e:
        reader.nns(5).by_vector([1.0, 2.0, 3.0])
    assert str(e.value) == "Invalid vector dimensions. Got 3 but expected 2"


def test_open_db_with_wrong_distance(api):
    """src/tests/reader.rs:61-79."""
    D, I = api
    db = I.Database(D.Euclidean)
    w = I.Writer(db, 0, 2)
    w.add_item(0, [0.0, 0.0])
    w.builder(rng()).n_trees(1).build()
    with pytest.raises(I.UnmatchingDistance) as e:
        I.Reader.open(db.remap_data_type(D.Manhattan), 0)
    assert (e.value.expected, e.value.received) == ("euclidean", "manhattan")
    assert str(e.value) == "Invalid distance provided. Got manhattan but expected euclidean"


@pytest.mark.parametrize("item", [2**32 - 2, 2**32 - 1])
def test_use_u32_max_for_a_vec(api, item):
    """src/tests/writer.rs:141-179: the largest item ids are ordinary ids."""
    D, I = api
    db = I.Database(D.Euclidean)
    w = I.Writer(db, 0, 3)
    w.add_item(item, [0.0, 1.0, 2.0])
    w.builder(rng()).n_trees(1).build()
    reader = I.Reader.open(db, 0)
    st = reader._st
    assert st.trees.roots == [0] and [int(x) for x in st.trees.nodes[0][1]] == [item]
    assert reader.item_ids() == [item]
    assert fmt(reader.nns(5).by_vector([0.0, 1.0, 2.0])) == [f"id({item}): distance(0)"]
    assert fmt(reader.nns(5).candidates([item]).by_item(item)) == [f"id({item}): distance(0)"]
    assert reader.nns(5).candidates([1, 2]).by_item(item) == []


def test_search_in_db_with_a_single_vector(api):
    """src/tests/reader.rs:81-99 (meilisearch#4296): cosine of an item with itself is 0."""
    D, I = api
    db = I.Database(D.Cosine)
    w = I.Writer(db, 0, 3)
    w.add_item(0, [0.00397, 0.553, 0.0])
    w.builder(rng()).build()
    reader = I.Reader.open(db, 0)
    assert fmt(reader.nns(1).by_item(0)) == ["id(0): distance(0)"]


@pytest.mark.parametrize("axis", [0, 1])
def test_two_dimension_on_a_line_and_on_a_column(api, axis):
    """src/tests/reader.rs:101-171: exhaustive searches are independent of the tree shapes."""
    D, I = api
    db = I.Database(D.Euclidean)
    w = I.Writer(db, 0, 2)
    for i in range(100):
        v = [0.0, 0.0]
        v[axis] = float(i)
        w.add_item(i, v)
    w.builder(rng()).n_trees(50).build()
    reader = I.Reader.open(db, 0)
    want = [f"id({i}): distance({i})" for i in range(5)]
    assert fmt(reader.nns(5).search_k(2**63).by_item(0)) == want
    assert fmt(reader.nns(5).by_item(0)) == want          # default search_k = count * n_trees = 250 >= 100 items
    # "if we can't look into enough nodes we find some random points": at least the item itself, sorted output
    res = reader.nns(5).search_k(1).by_item(1)
    assert res[0] == (1, 0.0) and [d for _, d in res] == sorted(d for _, d in res)
    assert reader.n_trees() == 50 and reader.n_items() == 100 and reader.dimensions() == 2
    stats = reader.stats()
    assert stats["leaf"] == 100 and len(stats["tree_stats"]) == 50
    assert all(t["descendants"] == t["split_nodes"] + 1 for t in stats["tree_stats"])


def test_get_item_ids(api):
    """src/tests/reader.rs:173-191."""
    D, I = api
    db = I.Database(D.Euclidean)
    w = I.Writer(db, 0, 2)
    for i in range(10):
        w.add_item(i, [0.0, float(i)])
    w.builder(rng()).n_trees(50).build()
    assert I.Reader.open(db, 0).item_ids() == list(range(10))


def test_filtering(api):
    """src/tests/reader.rs:194-227."""
    D, I = api
    db = I.Database(D.Euclidean)
    w = I.Writer(db, 0, 2)
    for i in range(100):
        w.add_item(i, [0.0, float(i)])
    w.builder(rng()).n_trees(50).build()
    reader = I.Reader.open(db, 0)
    assert fmt(reader.nns(5).candidates(range(0, 2)).by_item(0)) == ["id(0): distance(0)", "id(1): distance(1)"]
    assert fmt(reader.nns(5).candidates(range(98, 1000)).by_item(0)) == ["id(98): distance(98)", "id(99): distance(99)"]


def test_search_in_empty_database(api):
    """src/tests/reader.rs:229-243 (arroy#75)."""
    D, I = api
    db = I.Database(D.Euclidean)
    I.Writer(db, 0, 2).builder(rng()).build()
    assert I.Reader.open(db, 0).nns(10).by_vector([0.0, 0.0]) == []


def test_try_reading_in_a_non_built_database(api):
    """src/tests/reader.rs:245-281 (arroy#74)."""
    D, I = api
    db = I.Database(D.Euclidean)
    w = I.Writer(db, 0, 2)
    w.add_item(0, [0.0, 0.0])
    with pytest.raises(I.MissingMetadata):
        I.Reader.open(db, 0)
    w.builder(rng()).build()
    I.Writer(db, 0, 2).del_item(0)
    with pytest.raises(I.NeedBuild):
        I.Reader.open(db, 0)


def test_binary_quantized_item_vector(api):
    """src/tests/binary_quantized.rs:7-55: item_vector returns the +-1 representation, 0.0 -> +1, -0.1 -> -1."""
    D, I = api
    for dist in (D.BinaryQuantizedCosine, D.BinaryQuantizedEuclidean, D.BinaryQuantizedManhattan):
        db = I.Database(dist)
        w = I.Writer(db, 0, 16)
        vec = [-2.0, -1.0, 0.0, -0.1, 2.0, 2.0, -12.4, 21.2, -2.0, -1.0, 0.0, 1.0, 2.0, 2.0, -12.4, 21.2]
        w.add_item(0, vec)
        w.builder(rng()).build()
        got = I.Reader.open(db, 0).item_vector(0)
        want = [-1.0 if np.signbit(np.float32(x)) else 1.0 for x in vec]
        assert list(got) == want


def test_cancel_build(api):
    """src/tests/writer.rs:1346-1375: the cancel closure aborts the build with BuildCancelled."""
    D, I = api
    from arroy_amd import BuildCancelled
    db = I.Database(D.Euclidean)
    w = I.Writer(db, 0, 2)
    for i in range(100):
        w.add_item(i, [float(i), 0.0])
    with pytest.raises(BuildCancelled):
        w.builder(rng()).cancel(lambda: True).build()


@pytest.mark.parametrize("dist_name", ["Euclidean", "Manhattan", "Cosine", "DotProduct", "BinaryQuantizedCosine",
                                       "BinaryQuantizedEuclidean", "BinaryQuantizedManhattan"])
def test_search_recall_against_exhaustive_search(api, dist_name):
    """Quality check in the spirit of examples/compare_with_hnsw.rs: with a generous search_k the forest
    search returns (almost) the exhaustive top-k, and an exhaustive search_k returns it exactly."""
    D, I = api
    dist = getattr(D, dist_name)
    n, dims, k = 3000, 64, 10
    vecs = np.random.default_rng(5).standard_normal((n, dims)).astype(np.float32)
    db = I.Database(dist)
    w = I.Writer(db, 0, dims)
    for i in range(n):
        w.add_item(i, vecs[i])
    w.builder(rng()).n_trees(10).build()
    reader = I.Reader.open(db, 0)
    st = reader._st
    hits = total = 0
    for q in range(0, 40):
        exact_ids, exact_d = st.dataset.rerank(k, item=q)
        full = reader.nns(k).search_k(2**62).by_item(q)
        assert [i for i, _ in full] == list(exact_ids)
        got = reader.nns(k).search_k(1500).by_item(q)
        hits += len(set(i for i, _ in got) & set(int(i) for i in exact_ids))
        total += k
        qv = reader.nns(k).search_k(2**62).by_vector(vecs[q])
        assert [i for i, _ in qv] == list(exact_ids)
    assert hits / total > (0.5 if dist.binary_quantized else 0.7)


def test_write_one_vector_and_until_there_is_a_descendants(api):
    """src/tests/writer.rs:181-264: up to `dimensions` items the whole tree is one Descendants node per root."""
    D, I = api
    db = I.Database(D.Euclidean)
    w = I.Writer(db, 0, 3)
    w.add_item(0, [0, 1, 2])
    w.builder(rng()).n_trees(1).build()
    st = I.Reader.open(db, 0)._st
    assert st.trees.roots == [0] and len(st.trees.nodes) == 1 and list(st.trees.nodes[0][1]) == [0]
    for i in range(1, 3):
        w.add_item(i, [i, i, i])
    w.builder(rng()).n_trees(1).build()
    st = I.Reader.open(db, 0)._st
    assert st.trees.roots == [0] and len(st.trees.nodes) == 1 and list(st.trees.nodes[0][1]) == [0, 1, 2]
    # one more item than dimensions: the first split appears (src/tests/writer.rs:266-293)
    w.add_item(3, [3, 3, 3])
    w.builder(rng()).n_trees(1).build()
    st = I.Reader.open(db, 0)._st
    assert st.trees.roots == [0]  # the sub-tree's root keeps the descendant's id (src/writer.rs:693-702)
    root = st.trees.nodes[0]
    assert root[0] == "S" and root[4] is not None
    assert ["%.4f" % abs(x) for x in np.frombuffer(root[4], np.float32)] == ["0.5774"] * 3
    got = sorted(sorted(int(x) for x in st.trees.nodes[c][1]) for c in (root[1], root[2]))
    assert sum(len(g) for g in got) == 4 and sorted(sum(got, [])) == [0, 1, 2, 3]


def test_write_multiple_indexes(api):
    """src/tests/writer.rs:323-366: indexes of one database are independent."""
    D, I = api
    db = I.Database(D.Euclidean)
    for index in range(5):
        w = I.Writer(db, index, 3)
        w.add_item(0, [0, 1, 2])
        w.builder(rng()).n_trees(1).build()
    for index in range(5):
        r = I.Reader.open(db, index)
        assert r.item_ids() == [0] and fmt(r.nns(1).by_item(0)) == ["id(0): distance(0)"]


def check_trees(st, split_after=None):
    """Every tree partitions the item set; leaves respect split_after; ids of the store are unique per node."""
    items = set(st.items)
    for r in st.trees.roots:
        seen = []
        for nid in st.trees.subtree_ids(r):
            nd = st.trees.nodes[nid]
            if nd[0] == "D":
                assert list(nd[1]) == sorted(set(int(x) for x in nd[1]))
                if split_after is not None:
                    assert len(nd[1]) <= split_after
                seen += [int(x) for x in nd[1]]
        assert sorted(seen) == sorted(items), "tree %d does not partition the items" % r
    owned = sum((st.trees.subtree_ids(r) for r in st.trees.roots), [])
    assert len(owned) == len(set(owned)) == len(st.trees.nodes)


def check_exhaustive_search(reader, st, k=10, probes=8):
    ids = reader.item_ids()
    for q in ids[:: max(1, len(ids) // probes)]:
        exact_ids, exact_d = st.dataset.rerank(min(k, len(ids)), item=q)
        got = reader.nns(k).search_k(2**62).by_item(q)
        assert [i for i, _ in got] == [int(i) for i in exact_ids]
        assert np.array_equal(np.array([d for _, d in got], np.float32), exact_d)


@pytest.mark.parametrize("dist_name", ["Euclidean", "Cosine", "DotProduct", "BinaryQuantizedCosine"])
def test_incremental_add_and_delete(api, dist_name):
    """src/tests/writer.rs:368-1040 (add_one_item_incrementally*, delete_one_item*, delete_one_leaf_in_a_split,
    delete_document_in_an_empty_index...): after every incremental build each tree still partitions the item
    set, overgrown descendants were re-split, and an exhaustive search equals the brute-force ranking."""
    D, I = api
    dist = getattr(D, dist_name)
    dims, n0 = 24, 900
    g = np.random.default_rng(5)
    vecs = g.standard_normal((n0 + 400, dims)).astype(np.float32)
    db = I.Database(dist)
    w = I.Writer(db, 0, dims)
    for i in range(n0):
        w.add_item(i, vecs[i])
    w.builder(rng()).n_trees(5).build()
    st = I.Reader.open(db, 0)._st
    check_trees(st, dims)
    before = {nid: nd for nid, nd in st.trees.nodes.items() if nd[0] == "S"}
    # add 300 new items, overwrite 50, delete 100
    for i in range(n0, n0 + 300):
        w.add_item(i, vecs[i])
    for i in range(0, 50):
        w.add_item(i, vecs[n0 + 300 + i])
    for i in range(100, 200):
        assert w.del_item(i)
    assert w.need_build()
    w.builder(rng()).n_trees(5).build()
    reader = I.Reader.open(db, 0)
    st = reader._st
    assert reader.n_items() == n0 + 300 - 100 and reader.n_trees() == 5
    check_trees(st, dims)
    # incremental, not a rebuild: the old split planes are still there, byte for byte — except the few whose branch
    # collapsed when items left (`delete_items_in_file`); ids freed by this build are not handed out again in it
    # (a split that collapsed into a Descendants node keeps its id, and is split again — new plane — if the routed
    # items make it overflow)
    survivors = 0
    for nid, nd in before.items():
        now = st.trees.nodes.get(nid)
        if now is not None and now[0] == "S" and now[4] == nd[4] and np.array_equal(now[3], nd[3]):
            survivors += 1
    assert survivors > 0.3 * len(before)  # the lowest splits merge easily: their two leaves fit together again
    check_exhaustive_search(reader, st)
    # delete everything but a handful: back to ONE Descendants root (src/writer.rs:916-962)
    for i in list(st.items)[dims - 2:]:
        w.del_item(i)
    w.builder(rng()).n_trees(5).build()
    reader = I.Reader.open(db, 0)
    st = reader._st
    assert st.trees.roots == [0] and len(st.trees.nodes) == 1 and reader.n_trees() == 1
    check_exhaustive_search(reader, st, k=5)
    # and grow again from the single leaf
    for i in range(n0, n0 + 200):
        w.add_item(i, vecs[i])
    w.builder(rng()).build()
    reader = I.Reader.open(db, 0)
    check_trees(reader._st, dims)
    check_exhaustive_search(reader, reader._st)


def test_add_and_remove_trees(api):
    """src/tests/writer.rs:1042-1170 (delete_extraneous_tree, create_root_split_node_with_empty_child...):
    an explicit n_trees shrinks or grows the forest without touching the surviving trees."""
    D, I = api
    dims = 16
    g = np.random.default_rng(6)
    vecs = g.random((500, dims)).astype(np.float32)
    db = I.Database(D.Manhattan)
    w = I.Writer(db, 0, dims)
    for i in range(500):
        w.add_item(i, vecs[i])
    w.builder(rng()).n_trees(6).build()
    st = I.Reader.open(db, 0)._st
    roots6 = list(st.trees.roots)
    keep = {r: sorted(st.trees.subtree_ids(r)) for r in roots6}
    w.add_item(0, vecs[0])  # mark one item updated so a build is needed
    w.builder(rng()).n_trees(2).build()
    st = I.Reader.open(db, 0)._st
    # delete_extra_trees drops the oldest first with `roots.swap_remove(0)` (src/writer.rs:631-655): of
    # [r0..r5] the survivors are r2 and r1; the roots are sorted again after the delete step (:1001)
    assert st.trees.roots == sorted([roots6[1], roots6[2]])
    check_trees(st, dims)
    w.add_item(1, vecs[1])
    w.builder(rng()).n_trees(4).build()
    reader = I.Reader.open(db, 0)
    st = reader._st
    assert st.trees.roots[:2] == sorted([roots6[1], roots6[2]]) and len(st.trees.roots) == 4
    # the two new trees reuse node ids freed by the four dropped ones (ConcurrentNodeIds, src/parallel.rs:222-254)
    freed = set(sum((keep[r] for r in (roots6[0], roots6[3], roots6[4], roots6[5])), []))
    assert set(st.trees.roots[2:]) <= freed
    check_trees(st, dims)
    check_exhaustive_search(reader, st)


def test_delete_item_not_in_trees_then_search(api):
    """src/tests/writer.rs:560-640 (delete_one_item_in_a_descendant / delete_one_leaf_in_a_split): a deleted item
    never comes back from a search, and an emptied leaf is tolerated by the descent."""
    D, I = api
    db = I.Database(D.Euclidean)
    w = I.Writer(db, 0, 2)
    for i in range(6):
        w.add_item(i, [i, 0])
    w.builder(rng()).n_trees(1).build()
    for victim in (3, 0, 5):
        w.del_item(victim)
        w.builder(rng()).n_trees(1).build()
        reader = I.Reader.open(db, 0)
        got = reader.nns(10).search_k(2**62).by_vector([0, 0])
        assert victim not in [i for i, _ in got] and sorted(i for i, _ in got) == reader.item_ids()


def test_deletions_collapse_branches_like_the_reference(api):
    """`delete_items_in_file` (src/writer.rs:1021-1114), whose exact behaviour is pinned against the reference's
    snapshots by tests/test_oracle_reference_incremental.py (tests/ref_writer.py): the mirror must produce the same
    node table — a split whose child became empty is replaced by its other child, siblings that fit together are
    merged into their parent, everything else keeps its id."""
    import copy
    import sys
    sys.path.insert(0, __file__.rsplit("/", 1)[0])
    from ref_writer import RefWriter
    D, I = api
    dims = 8
    g = np.random.default_rng(11)
    vecs = g.standard_normal((400, dims)).astype(np.float32)
    db = I.Database(D.Euclidean)
    w = I.Writer(db, 0, dims)
    for i in range(400):
        w.add_item(i, vecs[i])
    w.builder(rng()).n_trees(3).build()
    st = I.Reader.open(db, 0)._st
    ref = RefWriter(0, dims)
    ref.nodes = {k: (("D", [int(x) for x in nd[1]]) if nd[0] == "D" else nd) for k, nd in copy.deepcopy(st.trees.nodes).items()}
    roots = list(st.trees.roots)
    victims = [int(x) for x in g.choice(400, 330, replace=False)]  # most items: plenty of empty and mergeable branches
    for v in victims:
        assert w.del_item(v)
    w.builder(rng()).n_trees(3).build()
    st = I.Reader.open(db, 0)._st
    want_roots = sorted(ref._delete_items(r, set(victims), dims)[0] for r in roots)
    assert st.trees.roots == want_roots
    assert sorted(st.trees.nodes) == sorted(ref.nodes)
    for k, nd in ref.nodes.items():
        got = st.trees.nodes[k]
        assert got[0] == nd[0]
        if nd[0] == "D":
            assert [int(x) for x in got[1]] == nd[1]
        else:
            assert (got[1], got[2]) == (nd[1], nd[2]) and got[4] == nd[4]
    check_trees(st)
    check_exhaustive_search(I.Reader.open(db, 0), st)
