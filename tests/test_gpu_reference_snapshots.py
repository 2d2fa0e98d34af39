"""The reference's incremental and low-memory writer snapshots replayed with every `create_split` and every `side`
computed by the HIP kernels (through the C ABI: `ah_create_split`, `ah_split_sides`), in the reference's own order and
with its own randomness (tests/ref_writer.py).  What the CPU replays (tests/test_oracle_reference_incremental.py) pin
for the oracle, these pin for the GPU directly against values asserted inside /root/reference — including the cosine
two-means / normalisation arithmetic, which no other reference test pins numerically."""
import numpy as np
import pytest

from oracle import oracle as O
from ref_writer import GpuBackend, RefWriter
from test_oracle_reference_incremental import SEED, check, line, replay_little_memory

pytestmark = pytest.mark.gpu


def gpu(distance_name, dims):
    from arroy_amd import distances as D
    cls = getattr(D, distance_name)
    return lambda item_ids, vecs: GpuBackend(cls, dims, item_ids, vecs)


def test_little_memory_cosine_snapshots_through_the_gpu(golden):
    """src/tests/writer.rs:1378-1403: Cosine, `available_memory(0)`, 188 + 409 tree nodes."""
    replay_little_memory(golden, gpu("Cosine", 3))


def test_add_one_item_incrementally_through_the_gpu(golden):
    """src/tests/writer.rs:868-965."""
    w, rng = RefWriter(O.EUCLIDEAN, 2, backend_factory=gpu("Euclidean", 2)), O.ChaCha12(SEED)
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


def test_reuse_node_id_through_the_gpu(golden):
    """src/tests/writer.rs:1123-1242."""
    w, rng = RefWriter(O.EUCLIDEAN, 2, backend_factory=gpu("Euclidean", 2)), O.ChaCha12(SEED)
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


def test_ten_tree_incremental_update_snapshot_through_the_gpu(golden):
    """src/tests/writer.rs:296-320, second dump (…-2.snap): 92 tree nodes after 50 overwrites."""
    g0, g = golden["random_points_10_trees"], golden["random_points_10_trees_updated"]
    rng = O.ChaCha12(SEED)
    w = RefWriter(O.EUCLIDEAN, g0["dims"], backend_factory=gpu("Euclidean", g0["dims"]))
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
