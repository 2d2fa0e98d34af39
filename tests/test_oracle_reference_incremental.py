"""Step-by-step replay of the reference's INCREMENTAL writer tests (src/tests/writer.rs, inline insta snapshots) with
the oracle's restatement of the reference's randomness and order (tests/ref_writer.py).  Every database dump the
reference asserts after every `build` — roots, item ids, every tree node with its id, children, bias and normal to
4 decimals, every descendants list — must be reproduced exactly.  Fixtures: tests/golden/reference_golden.json
(`writer_inline_snapshots`, extracted from the reference's test file by tests/golden/make_golden.py).  No GPU."""
import pytest

from oracle import oracle as O
from ref_writer import OracleBackend, RefWriter

SEED = bytes([42] * 32)  # src/tests/mod.rs:105-107


def check(golden, name, step, w):
    want = golden["writer_inline_snapshots"][name]["dumps"][step]
    got = w.dump()
    assert got["roots"] == want["roots"], f"{name} dump {step}: roots"
    assert got["item_ids"] == want["item_ids"], f"{name} dump {step}: items"
    assert sorted(got["trees"], key=int) == sorted(want["trees"], key=int), f"{name} dump {step}: tree node ids"
    for k, node in want["trees"].items():
        assert got["trees"][k] == node, f"{name} dump {step}: tree node {k}"


def line(n):  # items i -> [i, 0] (two dimensions)
    return [(i, [float(i), 0.0]) for i in range(n)]


def test_add_one_item_incrementally(golden):
    w, rng = RefWriter(O.EUCLIDEAN, 2), O.ChaCha12(SEED)
    for i, v in line(6):
        w.add_item(i, v)
    w.build(rng, n_trees=1)
    check(golden, "add_one_item_incrementally", 0, w)
    w.add_item(25, [25.0, 0.0])
    w.build(rng, n_trees=1)
    check(golden, "add_one_item_incrementally", 1, w)
    w.add_item(8, [8.0, 0.0])
    w.build(rng, n_trees=1)
    check(golden, "add_one_item_incrementally", 2, w)


def test_add_one_item_incrementally_to_create_a_split_node(golden):
    w, rng = RefWriter(O.EUCLIDEAN, 2), O.ChaCha12(SEED)
    w.add_item(0, [0.0, 0.0])
    w.add_item(1, [1.0, 0.0])
    w.build(rng, n_trees=1)
    check(golden, "add_one_item_incrementally_to_create_a_split_node", 0, w)
    w.add_item(2, [2.0, 0.0])
    w.build(rng, n_trees=1)
    check(golden, "add_one_item_incrementally_to_create_a_split_node", 1, w)


def test_add_one_item_incrementally_in_small_dbs(golden):
    w, rng = RefWriter(O.EUCLIDEAN, 2), O.ChaCha12(SEED)
    w.build(rng, n_trees=1)
    check(golden, "add_one_item_incrementally_in_an_empty_db", 0, w)
    w.add_item(0, [0.0, 0.0])
    w.build(rng, n_trees=1)
    check(golden, "add_one_item_incrementally_in_an_empty_db", 1, w)
    w, rng = RefWriter(O.EUCLIDEAN, 2), O.ChaCha12(SEED)
    w.add_item(0, [0.0, 0.0])
    w.build(rng, n_trees=1)
    check(golden, "add_one_item_incrementally_in_a_one_item_db", 0, w)
    w.add_item(1, [1.0, 0.0])
    w.build(rng, n_trees=1)
    check(golden, "add_one_item_incrementally_in_a_one_item_db", 1, w)


def test_overwrite_one_item_incremental(golden):
    w, rng = RefWriter(O.EUCLIDEAN, 2), O.ChaCha12(SEED)
    for i, v in line(6):
        w.add_item(i, v)
    w.build(rng, n_trees=1)
    check(golden, "overwrite_one_item_incremental", 0, w)
    w.add_item(3, [6.0, 0.0])
    w.build(rng, n_trees=1)
    check(golden, "overwrite_one_item_incremental", 1, w)


def test_delete_one_item(golden):
    w, rng = RefWriter(O.EUCLIDEAN, 2), O.ChaCha12(SEED)
    for i, v in line(6):
        w.add_item(i, v)
    w.build(rng, n_trees=1)
    check(golden, "delete_one_item", 0, w)
    w.del_item(3)
    w.build(rng, n_trees=1)
    check(golden, "delete_one_item", 1, w)
    w.del_item(1)
    w.build(rng, n_trees=1)
    check(golden, "delete_one_item", 2, w)


def test_delete_in_small_dbs(golden):
    w, rng = RefWriter(O.EUCLIDEAN, 2), O.ChaCha12(SEED)
    w.add_item(0, [0.0, 0.0])
    w.build(rng, n_trees=1)
    check(golden, "delete_one_item_in_a_one_item_db", 0, w)
    w.del_item(0)
    w.build(rng, n_trees=1)
    check(golden, "delete_one_item_in_a_one_item_db", 1, w)
    w, rng = RefWriter(O.EUCLIDEAN, 2), O.ChaCha12(SEED)
    w.add_item(0, [0.0, 0.0])
    w.add_item(1, [1.0, 0.0])
    w.build(rng, n_trees=1)
    check(golden, "delete_one_item_in_a_descendant", 0, w)
    w.del_item(0)
    w.build(rng, n_trees=1)
    check(golden, "delete_one_item_in_a_descendant", 1, w)
    w, rng = RefWriter(O.COSINE, 2), O.ChaCha12(SEED)
    w.add_item(0, [0.0, 0.0])
    w.build(rng)
    check(golden, "delete_one_item_in_a_single_document_database", 0, w)
    w.del_item(0)
    w.build(rng)
    check(golden, "delete_one_item_in_a_single_document_database", 1, w)


def test_delete_one_leaf_in_a_split(golden):
    w, rng = RefWriter(O.EUCLIDEAN, 2), O.ChaCha12(SEED)
    for i, v in line(3):
        w.add_item(i, v)
    w.build(rng, n_trees=1)
    check(golden, "delete_one_leaf_in_a_split", 0, w)
    w.del_item(1)
    w.build(rng, n_trees=1)
    check(golden, "delete_one_leaf_in_a_split", 1, w)


def test_create_root_split_node_with_empty_child(golden):
    w, rng = RefWriter(O.EUCLIDEAN, 2), O.ChaCha12(SEED)
    for i, v in line(6):
        w.add_item(i, v)
    w.build(rng, n_trees=1)
    check(golden, "create_root_split_node_with_empty_child", 0, w)
    w.del_item(1)
    w.del_item(5)
    w.build(rng, n_trees=1)
    check(golden, "create_root_split_node_with_empty_child", 1, w)
    w.del_item(0)
    w.build(rng, n_trees=1)
    check(golden, "create_root_split_node_with_empty_child", 2, w)


def test_reuse_node_id(golden):
    w, rng = RefWriter(O.EUCLIDEAN, 2), O.ChaCha12(SEED)
    for i, v in line(6):
        w.add_item(i, v)
    w.build(rng, n_trees=1)
    check(golden, "reuse_node_id", 0, w)
    w.del_item(4)
    w.build(rng, n_trees=1)
    check(golden, "reuse_node_id", 1, w)
    w.add_item(4, [4.0, 0.0])
    w.build(rng, n_trees=1)
    check(golden, "reuse_node_id", 2, w)
    w.build(rng, n_trees=2)
    check(golden, "reuse_node_id", 3, w)


def test_delete_extraneous_tree(golden):
    w, rng = RefWriter(O.EUCLIDEAN, 4), O.ChaCha12(SEED)
    for i in range(5):
        w.add_item(i, [float(i), 0.0, 0.0, 0.0])
    w.build(rng)
    check(golden, "delete_extraneous_tree", 0, w)
    # the reference's test re-opens the index with `Writer::new(.., 0, 2)`: from here on `dimensions`, hence
    # `fit_in_descendant`, is 2 although the stored vectors have 4 components
    w.build(rng, n_trees=2, split_after=2)
    check(golden, "delete_extraneous_tree", 1, w)
    w.build(rng, n_trees=1, split_after=2)
    check(golden, "delete_extraneous_tree", 2, w)


def test_write_and_update_lot_of_random_points_second_snapshot(golden):
    """src/tests/writer.rs:296-320 + snapshots/arroy__tests__writer__write_and_update_lot_of_random_points-2.snap:
    100 x 30-d Euclidean, 10 trees, then every even item overwritten with a fresh random vector and the index built
    again INCREMENTALLY.  92 tree nodes.  This one depends on everything at once: one rng stream across the item
    generation and both builds, the removal of 50 items from 10 trees with branch collapsing, their re-insertion
    through the surviving planes, node-id reuse, and the ORDER in which the touched descendants are re-split — the
    iteration order of hashbrown maps with the identity hash and rayon's reduce tree (tests/ref_writer.py)."""
    import numpy as np
    g0, g = golden["random_points_10_trees"], golden["random_points_10_trees_updated"]
    rng = O.ChaCha12(SEED)
    w = RefWriter(O.EUCLIDEAN, g0["dims"])
    for i in range(g0["n_items"]):
        w.add_item(i, [rng.gen_f32() for _ in range(g0["dims"])])
    w.build(rng, n_trees=g0["n_trees"])
    for i in range(0, g0["n_items"], 2):
        w.add_item(i, [rng.gen_f32() for _ in range(g0["dims"])])
    w.build(rng, n_trees=g0["n_trees"])
    assert w.roots == g["roots"]
    assert sorted(w.nodes) == sorted(int(k) for k in g["trees"])
    for k, want in g["trees"].items():
        got = w.nodes[int(k)]
        if want["kind"] == "D":
            assert got == ("D", want["descendants"]), f"tree node {k}"
        else:
            assert got[0] == "S" and (got[1], got[2]) == (want["left"], want["right"]), f"tree node {k}"
            assert "%.4f" % got[3][0] == want["bias"], f"bias of tree node {k}"
            assert ["%.4f" % x for x in np.frombuffer(got[4], dtype=np.float32)][:10] == want["vector10"], f"normal of {k}"


def replay_little_memory(golden, backend_factory):
    import numpy as np
    dumps = golden["little_memory"]["dumps"]
    rng = O.ChaCha12(SEED)
    w = RefWriter(O.COSINE, 3, backend_factory=backend_factory)
    for i in range(100):
        w.add_item(i, [rng.gen_f32() for _ in range(3)])
    w.build(rng, n_trees=2, available_memory=0)
    for step, todo in enumerate([None, list(range(0, 100, 2)) + list(range(100, 150))]):
        if todo is not None:
            for i in todo:
                w.add_item(i, [rng.gen_f32() for _ in range(3)])
            w.build(rng, n_trees=3, available_memory=0)
        want = dumps[step]
        assert w.roots == want["roots"]
        assert sorted(w.nodes) == sorted(int(k) for k in want["trees"])
        for k, node in want["trees"].items():
            got = w.nodes[int(k)]
            if node["kind"] == "D":
                assert got == ("D", node["descendants"]), f"dump {step}: tree node {k}"
            else:
                assert got[0] == "S" and (got[1], got[2]) == (node["left"], node["right"]), f"dump {step}: tree node {k}"
                assert ["%.4f" % x for x in np.frombuffer(got[4], dtype=np.float32)] == node["vector"], f"dump {step}: normal {k}"


def test_little_memory_replay_with_the_python_tree_builder(golden):
    """Same snapshots through RefWriter's backend interface (the path the GPU replay uses, here with the oracle as
    backend): the Python restatement of make_tree_in_file and the C one must agree."""
    replay_little_memory(golden, lambda item_ids, vecs: OracleBackend(O.COSINE, item_ids, vecs))


def test_write_and_update_lot_of_random_points_with_little_memory(golden):
    """src/tests/writer.rs:1378-1403 + its two .snap files: COSINE, 3 dimensions, `available_memory(0)` — every task
    first builds a tree over `dimensions + 1` randomly chosen items (`fit_in_memory`, :1536-1584, `gen_range` on u64),
    then feeds the rest through that tree four at a time (`insert_items_in_descendants_from_tmpfile`, :1463-1531) and
    spawns nested tasks for the leaves that overflowed (:725-737), last in first out.  188 tree nodes after the first
    build (2 trees), 409 after 50 overwrites + 50 new items and a third tree.  Besides the order of everything, this
    is the reference's own pin of the COSINE arithmetic: two_means with normalisation, `create_split`, `side` — every
    normal is compared to the 4 decimals the snapshot prints, every descendants list exactly."""
    replay_little_memory(golden, None)


@pytest.mark.parametrize("name,dims,items,n_trees", [
    ("write_one_vector", 3, [(0, [0.0, 1.0, 2.0])], None),
    ("write_one_vector_in_one_tree", 3, [(0, [0.0, 1.0, 2.0])], 1),
    ("write_one_vector_in_multiple_trees", 3, [(0, [0.0, 1.0, 2.0])], 10),
    ("use_u32_max_minus_one_for_a_vec", 3, [(2**32 - 2, [0.0, 1.0, 2.0])], 1),
    ("use_u32_max_for_a_vec", 3, [(2**32 - 1, [0.0, 1.0, 2.0])], 1),
    ("write_vectors_until_there_is_a_descendants", 3, [(i, [float(i)] * 3) for i in range(3)], 1),
    ("write_vectors_until_there_is_a_split", 3, [(i, [float(i)] * 3) for i in range(4)], 1),
])
def test_single_build_inline_snapshots(golden, name, dims, items, n_trees):
    """The one-build tests of src/tests/writer.rs:141-293 through the same replay: whatever `n_trees` says, items that
    fit in one descendant give ONE root with id 0 (`clear_db_and_create_a_single_leaf`, :916-962)."""
    w, rng = RefWriter(O.EUCLIDEAN, dims), O.ChaCha12(SEED)
    for i, v in items:
        w.add_item(i, v)
    w.build(rng, n_trees=n_trees)
    check(golden, name, 0, w)
