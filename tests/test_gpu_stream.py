"""ah_build_forest_stream: the node sink DURING the build (`TmpNodes::put`, src/parallel.rs:130-147; the final drain,
src/writer.rs:597-607) — the forest must be the one ah_build_forest materialises and the oracle builds, node for node,
whatever the batching; breadth-first order, unique dense ids, nothing handed over twice."""
import numpy as np
import pytest

import test_gpu_parity as P
from oracle import oracle as O
from test_gpu_parity import make_data

pytestmark = pytest.mark.gpu

from arroy_amd import _lib  # noqa: E402
from arroy_amd import distances as D  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def _imports():
    import arroy_amd
    assert arroy_amd.device_count() >= 1, "no GPU visible: these tests must run on an MI355X"
    P.D, P.O = D, O


@pytest.mark.parametrize("cls,dims,n", [(D.Cosine, 96, 20_000), (D.Euclidean, 48, 30_000), (D.DotProduct, 64, 12_000),
                                        (D.Manhattan, 40, 9_000), (D.BinaryQuantizedCosine, 128, 15_000)],
                         ids=["cosine", "euclidean", "dot", "manhattan", "bq_cosine"])
def test_streamed_forest_equals_the_materialised_one_and_the_oracle(cls, dims, n):
    ds, oracle, vecs, ids = make_data(cls, n, dims, seed=dims + n)
    seeds = list(range(900, 907))
    forest = ds.build_forest(seeds)
    want = [forest.canonical(t) for t in range(len(seeds))]
    assert want[0] == oracle.build_tree(0, seeds[0]).canonical() and want[6] == oracle.build_tree(0, seeds[6]).canonical()
    for kw in (dict(), dict(max_trees_in_flight=3), dict(margin_mode=_lib.MARGIN_EXACT_ONLY, max_host_threads=1)):
        roots, stats, got = ds.build_forest_stream(seeds, **kw)
        assert [got.canonical(t) for t in range(len(seeds))] == want, kw
        n_nodes = len(got.splits) + len(got.leaves)
        assert n_nodes == len(forest.nodes) and stats["split_nodes"] == len(got.splits) and stats["descendant_nodes"] == len(got.leaves)
        # ids: dense from 0, unique (StreamedForest.take asserts no id arrives twice); the children of a node are consecutive
        assert sorted(list(got.splits) + list(got.leaves)) == list(range(n_nodes))
        for i, (nb, left, right, tree, depth, count) in got.splits.items():
            assert right == left + 1 and left > i  # breadth-first: a parent arrives (and is numbered) before its children
            for c in (left, right):
                child = got.splits.get(c)
                assert (child[3], child[4]) == (tree, depth + 1) if child else got.leaves[c][1:] == (tree, depth + 1)
        # order of arrival: split planes level by level inside a batch of trees, then that batch's Descendants nodes
        kinds = [k for k, _lvl, _n, _b in got.batches]
        n_tree_batches = 1 if not kw.get("max_trees_in_flight") else -(-len(seeds) // kw["max_trees_in_flight"])
        assert sum(1 for a, b in zip(kinds, kinds[1:]) if (a, b) == (1, 2)) == n_tree_batches - 1
        assert stats["margin_evaluations"] == forest.stats["margin_evaluations"] and stats["levels"] == forest.stats["levels"]
        assert sorted(i for leaf in got.leaves.values() if leaf[1] == 0 for i in leaf[0]) == sorted(int(x) for x in ids)
    ds.close()


def test_stream_sink_can_stop_the_build_and_small_datasets_are_one_leaf_per_tree():
    from arroy_amd import BuildCancelled
    ds, oracle, vecs, ids = make_data(D.Euclidean, 5000, 32, seed=5)
    calls = []

    def stop_at_third(batch):
        calls.append(int(batch.kind))
        return 7 if len(calls) == 3 else 0
    with pytest.raises(BuildCancelled):
        ds.build_forest_stream([1, 2, 3], sink=stop_at_third, split_after=20)
    assert len(calls) == 3
    with pytest.raises(ZeroDivisionError):  # an exception inside the sink stops the build and comes back to the caller
        ds.build_forest_stream([1, 2, 3], sink=lambda b: 1 // 0, split_after=20)
    # fit_in_descendant at the root (src/writer.rs:1183-1188): one Descendants node per tree, straight from the host
    roots, stats, got = ds.build_forest_stream([4, 5], split_after=6000)
    assert list(roots) == [0, 1] and not got.splits and [got.leaves[r][0] for r in (0, 1)] == [tuple(int(x) for x in ids)] * 2
    # a leaf that does not fit half of the pinned stream buffer is refused, not truncated
    ds.close()


def test_stream_at_full_size_10m_x_768_100_trees():
    """BASELINE configs[2] through the sink: every node once, every item id once per tree, the bytes of the split planes and
    of the id lists those of the materialised forest."""
    from arroy_amd import Dataset, shard
    n, dims, trees = 10_000_000, 768, 100
    ds = Dataset(D.Cosine, dims, n)
    ds.fill_synthetic(42, 1, n)
    ds.finalize()
    seeds = shard.tree_seeds(42, range(trees))
    forest = ds.build_forest(seeds)
    nodes = forest.nodes
    want_splits, want_leaves = int((nodes["kind"] == 2).sum()), int((nodes["kind"] == 1).sum())
    vec_bytes = 4 * dims
    planes = forest.normals.reshape(-1, forest.normal_stride)[:, :vec_bytes + 4]
    want_plane_sum = int(planes.view(np.uint32).sum(dtype=np.uint64))
    want_ids_weighted = int((forest.descendants.astype(np.uint64) * (np.arange(forest.descendants.size, dtype=np.uint64) % 1009)).sum())
    per_tree_counts = np.bincount(nodes["tree"], minlength=trees)
    forest.close()
    seen = {"splits": 0, "leaves": 0, "plane_sum": 0, "ids_weighted": 0, "pos": 0, "ids": 0, "last_level": -1, "desc_started": False,
            "per_tree": np.zeros(trees, dtype=np.int64), "desc_of": np.zeros(trees, dtype=bool),
            "level_of": np.full(trees, -1, dtype=np.int64), "last_desc_tree": -1}

    def sink(b):
        m = int(b.n_nodes)
        arr = np.ctypeslib.as_array(C.cast(b.nodes, C.POINTER(C.c_uint8)), shape=(m * C.sizeof(_lib.AhStreamNode),))
        nd = arr.view(np.dtype([("id", "<u4"), ("tree", "<u4"), ("kind", "u1"), ("has_normal", "u1"), ("reserved", "<u2"),
                                ("left", "<u4"), ("right", "<u4"), ("count", "<u4"), ("depth", "<u4"), ("payload_offset", "<u8")],
                               align=True))
        seen["per_tree"] += np.bincount(nd["tree"], minlength=trees)
        payload = np.ctypeslib.as_array(b.payload, shape=(int(b.payload_len),))
        if b.kind == 2:
            # a tree's split planes all arrive before its Descendants nodes (the tail of this build runs in groups of trees:
            # a group's ids travel while the next group's last levels are computed), level by level for any one tree
            assert not seen["desc_of"][nd["tree"]].any() and (int(b.level) >= seen["level_of"][nd["tree"]]).all()
            seen["level_of"][nd["tree"]] = int(b.level)
            seen["last_level"] = max(seen["last_level"], int(b.level))
            seen["splits"] += m
            recs = payload.reshape(m, int(b.normal_stride))[:, :vec_bytes + 4]
            seen["plane_sum"] += int(recs.view(np.uint32).sum(dtype=np.uint64))
        else:
            seen["desc_started"] = True
            assert int(nd["tree"][0]) >= seen["last_desc_tree"] and (np.diff(nd["tree"].astype(np.int64)) >= 0).all()
            seen["last_desc_tree"] = int(nd["tree"][-1])
            seen["desc_of"][nd["tree"]] = True
            seen["leaves"] += m
            ids = payload.view(np.uint32)
            assert int(nd["count"].sum()) == ids.size and int(nd["count"].max()) <= dims
            pos = np.arange(seen["pos"], seen["pos"] + ids.size, dtype=np.uint64) % 1009
            seen["ids_weighted"] += int((ids.astype(np.uint64) * pos).sum())
            seen["pos"] += ids.size
        return 0
    import ctypes as C
    roots, stats, _ = ds.build_forest_stream(seeds, sink=sink)
    assert (seen["splits"], seen["leaves"]) == (want_splits, want_leaves) == (stats["split_nodes"], stats["descendant_nodes"])
    assert seen["pos"] == trees * n and seen["ids_weighted"] == want_ids_weighted  # the ids, in the blob's order
    assert seen["plane_sum"] == want_plane_sum                                      # every split plane's vector + first header word
    assert (seen["per_tree"] == per_tree_counts).all() and len(set(int(r) for r in roots)) == trees
    assert stats["screen_violations"] == 0 and stats["host_blob_recycled"] == 0
    assert stats["tail_groups"] == 5 and seen["desc_of"].all()  # (AH_BUILD_TAIL_GROUPS' default: level 13 on, group by group)
    ds.close()
