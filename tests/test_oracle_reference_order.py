"""Replay of the reference's insta snapshots (src/tests/writer.rs) with the oracle's restatement of the
reference's OWN randomness: rand 0.8 `StdRng` = ChaCha12 seeded with [42; 32] (src/tests/mod.rs:105-107), consumed
in the reference's depth-first order on one rayon thread.  This pins, against values asserted inside
/root/reference: the ChaCha12 stream and `gen::<f32>()`, `seq::index::sample`, `gen_range`, `two_means`,
`create_split` (normal + bias), `side`, the accept rule, node-id allocation order and the descendants lists."""
import numpy as np

from conftest import hex_f32
from oracle import oracle as O

SEED = bytes([42] * 32)


def show(tree):
    """{id: printable node} in the snapshot's formatting (4 decimals)."""
    out = {}
    for k, nd in tree.items():
        if nd[0] == "D":
            out[k] = ("D", nd[1])
        else:
            vec = None if nd[4] is None else ["%.4f" % x for x in np.frombuffer(nd[4], dtype=np.float32)]
            out[k] = ("S", nd[1], nd[2], "%.4f" % nd[3][0], vec)
    return out


def test_write_vectors_until_there_is_a_split():
    """src/tests/writer.rs:266-293."""
    data = O.Data(O.EUCLIDEAN, np.array([[i, i, i] for i in range(4)], dtype=np.float32))
    assert show(O.build_tree_reference_order(data, 0, SEED)) == {
        0: ("S", 1, 2, "-2.3960", ["0.5774", "0.5774", "0.5774"]),
        1: ("D", [0, 1]),
        2: ("D", [2, 3]),
    }


def test_delete_one_leaf_in_a_split_first_build():
    """src/tests/writer.rs:592-616."""
    data = O.Data(O.EUCLIDEAN, np.array([[0, 0], [1, 0], [2, 0]], dtype=np.float32))
    assert show(O.build_tree_reference_order(data, 0, SEED)) == {
        0: ("S", 1, 2, "-0.7143", ["1.0000", "0.0000"]),
        1: ("D", [0]),
        2: ("D", [1, 2]),
    }


def test_create_root_split_node_with_empty_child_first_build():
    """src/tests/writer.rs:1043-1072 (same tree in delete_one_item, :676-706)."""
    data = O.Data(O.EUCLIDEAN, np.array([[i, 0] for i in range(6)], dtype=np.float32))
    assert show(O.build_tree_reference_order(data, 0, SEED)) == {
        0: ("S", 3, 6, "-2.7500", ["1.0000", "0.0000"]),
        1: ("D", [2]),
        2: ("D", [0, 1]),
        3: ("S", 1, 2, "1.0000", ["-1.0000", "0.0000"]),
        4: ("D", [4, 5]),
        5: ("D", [3]),
        6: ("S", 4, 5, "3.7500", ["-1.0000", "0.0000"]),
    }


def random_points(n=100, dims=30):
    """`std::array::from_fn(|_| rng.gen())` for ids 0..n (src/tests/writer.rs:301-304)."""
    rng = O.ChaCha12(SEED)
    return np.array([[rng.gen_f32() for _ in range(dims)] for _ in range(n)], dtype=np.float32)


def test_chacha12_f32_stream_matches_the_lmdb_fixture_bit_for_bit(golden):
    """src/tests/assets/v0_6/large.mdb was written by the same test with the same rng: odd ids still hold the
    vectors of the first batch (even ids were overwritten later), i.e. draws 30*id .. 30*id+29 of the stream."""
    g = golden["large_v0_6"]
    vecs = random_points()
    for row, item in enumerate(g["ids"]):
        if item % 2 == 1:
            assert vecs[item].tobytes() == hex_f32(g["vectors_hex"][row]).tobytes(), f"item {item}"


def test_write_and_update_lot_of_random_points_first_snapshot(golden):
    """src/tests/writer.rs:296-308 + snapshots/arroy__tests__writer__write_and_update_lot_of_random_points.snap:
    100 x 30-d Euclidean, 10 trees — every tree node (ids, children, bias, normal prefix, descendants)."""
    g = golden["random_points_10_trees"]
    vecs = random_points(g["n_items"], g["dims"])
    for item, comps in g["items10"].items():
        assert ["%.4f" % x for x in vecs[int(item)][:10]] == comps
    data = O.Data(O.EUCLIDEAN, vecs)
    tree = show(O.build_tree_reference_order(data, 0, SEED, n_trees=g["n_trees"], skip_u32=g["n_items"] * g["dims"]))
    assert sorted(tree) == sorted(int(k) for k in g["trees"])
    for k, want in g["trees"].items():
        got = tree[int(k)]
        if want["kind"] == "D":
            assert got == ("D", want["descendants"]), f"tree node {k}"
        else:
            assert got[0] == "S" and (got[1], got[2]) == (want["left"], want["right"]), f"tree node {k}"
            assert got[3] == want["bias"], f"bias of tree node {k}"
            assert got[4][:10] == want["vector10"], f"normal of tree node {k}"
