"""The tail of a build in groups of trees (AH_BUILD_TAIL_GROUPS; build_batch in arroy_amd/csrc/forest.hip): the last big
level and what follows it run tree group by tree group, so that a group's item ids and normals leave the device under
the next group's kernels.  Trees never interact (src/writer.rs:556-561), so the forest must be the one the
level-by-level loop builds and the oracle builds (`make_tree_in_file`, src/writer.rs:1167-1261), node for node, whatever
the number of groups — materialised and streamed, with item ids that are not the row numbers, for every metric family."""
import time

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


CASES = [(D.Cosine, 96, 24_000, 9), (D.Euclidean, 48, 30_000, 7), (D.DotProduct, 64, 12_000, 5), (D.Manhattan, 40, 9_000, 4),
         (D.BinaryQuantizedCosine, 128, 15_000, 6), (D.BinaryQuantizedEuclidean, 64, 8_000, 3)]


@pytest.mark.parametrize("cls,dims,n,trees", CASES, ids=[c[0].__name__ for c in CASES])
def test_grouped_tail_builds_the_same_forest(cls, dims, n, trees):
    ds, oracle, vecs, ids = make_data(cls, n, dims, seed=3 * dims + n)
    seeds = list(range(4100, 4100 + trees))
    with _lib.tuning(AH_BUILD_TAIL_GROUPS=0):
        plain = ds.build_forest(seeds)
    assert plain.stats["tail_groups"] == 0
    want = [plain.canonical(t) for t in range(trees)]
    assert want[0] == oracle.build_tree(0, seeds[0]).canonical()
    assert want[trees - 1] == oracle.build_tree(0, seeds[trees - 1]).canonical()
    digest = plain.digest()[0]
    # (tunables, build options, must the tail run in groups?)  AUTO cuts the level only where it would run node-major:
    # AH_ROWMAJOR=0 / a forced node-major mode make that every level of these small forests
    configs = [({}, {}, False), ({"AH_ROWMAJOR": 0}, {}, True), ({}, dict(margin_mode=_lib.MARGIN_NODE_MAJOR), True),
               ({}, dict(margin_mode=_lib.MARGIN_EXACT_ONLY, max_host_threads=1), False),
               ({"AH_ROWMAJOR": 0}, dict(max_trees_in_flight=max(2, trees // 2)), True)]
    bq = cls.metric >= 4
    for groups in (2, 3, 4, 32):
        for knobs, kw, must in configs:
            with _lib.tuning(AH_BUILD_TAIL_GROUPS=groups, AH_BUILD_TAIL_MIN_MB=0, AH_SCREEN_VERIFY=1, **knobs):
                forest = ds.build_forest(seeds, **kw)
                if must or bq:
                    assert forest.stats["tail_groups"] >= 2, (groups, knobs, kw, forest.stats["tail_groups"])
                assert forest.stats["screen_violations"] == 0
                assert forest.digest()[0] == digest, (groups, knobs, kw)
                assert forest.stats["margin_evaluations"] == plain.stats["margin_evaluations"]
                assert forest.stats["split_nodes"] == plain.stats["split_nodes"]
                if groups == 3:
                    assert [forest.canonical(t) for t in range(trees)] == want
                roots, stats, got = ds.build_forest_stream(seeds, **kw)
                assert stats["tail_groups"] == forest.stats["tail_groups"]
                assert [got.canonical(t) for t in range(trees)] == want, (groups, knobs, kw)
                n_nodes = len(got.splits) + len(got.leaves)
                assert n_nodes == len(plain.nodes) and sorted(list(got.splits) + list(got.leaves)) == list(range(n_nodes))
                for i, (nb, left, right, tree, depth, count) in got.splits.items():
                    assert right == left + 1 and left > i  # a parent arrives (and is numbered) before its children
                # the Descendants nodes still arrive in ascending (tree, position) order, a group of trees at a time
                leaf_trees = [leaf[1] for leaf in got.leaves.values()]  # (a dict keeps the order of arrival)
                assert leaf_trees == sorted(leaf_trees)
    ds.close()


def test_grouped_tail_with_caller_ids_and_a_cancel_flag():
    """Item ids that are not row numbers go through k_rows_to_ids group by group; a raised cancel flag still ends the call."""
    from arroy_amd import BuildCancelled
    n, dims = 20_000, 64
    ids = np.sort(np.random.default_rng(77).permutation(5 * n)[:n]).astype(np.uint32)  # (uploads ascend)
    ds, oracle, vecs, ids = make_data(D.Euclidean, n, dims, seed=77, ids=ids)
    seeds = list(range(50, 58))
    with _lib.tuning(AH_BUILD_TAIL_GROUPS=0):
        plain = ds.build_forest(seeds)
    assert plain.canonical(3) == oracle.build_tree(0, seeds[3]).canonical()
    with _lib.tuning(AH_BUILD_TAIL_GROUPS=4, AH_BUILD_TAIL_MIN_MB=0, AH_ROWMAJOR=0):
        forest = ds.build_forest(seeds)
        assert forest.stats["tail_groups"] == 4 and forest.digest()[0] == plain.digest()[0]
        assert [forest.canonical(t) for t in range(8)] == [plain.canonical(t) for t in range(8)]
        roots, stats, got = ds.build_forest_stream(seeds)
        assert stats["tail_groups"] == 4 and [got.canonical(t) for t in range(8)] == [plain.canonical(t) for t in range(8)]
        seen = []

        def progress(level, nodes_done, items_routed):
            seen.append(level)
            if len(seen) >= plain.stats["levels"] - 2:  # (called under the first group's first level: hold the library's
                time.sleep(0.02)                        # thread there until the watcher thread has raised the flag)

        with pytest.raises(BuildCancelled):
            ds.build_forest(seeds, cancel=lambda: len(seen) >= plain.stats["levels"] - 2, progress=progress)
        assert ds.build_forest(seeds).digest()[0] == plain.digest()[0]  # and the dataset builds again afterwards
    ds.close()
