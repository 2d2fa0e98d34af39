"""Allocation failures never cross the C ABI as anything but a status code (src/writer.rs:799-827: the reference catches its
workers' panics and returns Error::Panic; include/arroy_hip.h: "no C++ exception crosses the ABI").

`AH_FAIL_ALLOC_AFTER = n` (a tunable of the library, common.h) makes the n-th allocation counted from that moment fail:
device blocks of the caching allocator, pinned host memory, the host blobs of a forest, and — while the calling thread is
inside an entry point — the library's own `operator new` (std::vector, std::string, new).  Every test sweeps n = 1, 2, 3 ...
over one entry point until a call goes through without the counter firing, i.e. until EVERY allocation of that call has been
the failing one once, and asserts after each call:

* the process is still alive and the call returned AH_OK, AH_ERR_OUT_OF_MEMORY or AH_ERR_DEVICE (a failure of an optional
  allocation — the screens' copies — may leave the call successful);
* nothing leaked: the bytes of HBM the library has handed out are not above their value before the sweep;
* the library still works afterwards: the same call without the fault returns the reference answer bit for bit."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu

from arroy_amd import Dataset, Index, _lib  # noqa: E402
from arroy_amd import distances as D  # noqa: E402

N, DIMS, TREES = 20_000, 64, 4
OK, DEVICE, OOM = 0, 3, 4


@pytest.fixture(scope="module")
def world():
    import arroy_amd
    assert arroy_amd.device_count() >= 1, "no GPU visible: these tests must run on an MI355X"
    vecs = O.synth(11, 1, N, DIMS)
    ds = Dataset(D.Cosine, DIMS, N)
    ds.upload_vectors(np.arange(N, dtype=np.uint32), vecs)
    ds.finalize()
    seeds = [101, 102, 103, 104]
    forest = ds.build_forest(seeds)
    index = ds.create_index(forest)
    queries = vecs[[5, 77, 1234, 19_999]] + np.float32(1e-3)
    yield ds, vecs, seeds, forest, index, queries
    _lib.tuning_set("AH_FAIL_ALLOC_AFTER", 0)
    index.close()
    forest.close()
    ds.close()


def sweep_once(op, cleanup, limit):
    seen = []
    for n in range(1, limit):
        _lib.tuning_set("AH_FAIL_ALLOC_AFTER", n)
        try:
            res, status = op(), OK
        except _lib.ArroyHipError as e:
            res, status = None, e.status
        finally:
            left = _lib.tuning_get("AH_FAIL_ALLOC_AFTER")[0]
            _lib.tuning_set("AH_FAIL_ALLOC_AFTER", 0)
        assert status in (OK, DEVICE, OOM), (n, status)
        if status != OK:
            assert _lib.lib().ah_last_error() != b"", n
        if res is not None and cleanup is not None:
            cleanup(res)
        if left > 0:  # the counter never fired: the call has no n-th allocation
            assert status == OK
            return seen
        seen.append(status)
    raise AssertionError(f"more than {limit} allocations in one call?")


def sweep(op, cleanup=None, limit=2000):
    """op() with the n-th allocation failing, n = 1, 2, ...; returns the statuses seen before the clean pass.  Leaks: a call may
    legitimately leave the calling thread's scratch larger (or, after a failed growth, empty), so the bytes the library holds
    are compared between two passes that end in the same state — the clean call of a first sweep and the clean call of a
    second one: whatever a FAILING call leaked in between would still be there."""
    seen = sweep_once(op, cleanup, limit)
    live1, _ = _lib.device_cache_stats(0)
    again = sweep_once(op, cleanup, limit)
    live2, _ = _lib.device_cache_stats(0)
    assert live2 <= live1, f"{len(again)} failing calls leaked {live2 - live1} bytes of HBM"
    return seen


def test_index_create_from_view_survives_every_allocation_failure(world):
    ds, _vecs, _seeds, forest, index, queries = world
    view = forest.view_struct()
    seen = sweep(lambda: Index(ds, None, view=view), cleanup=lambda ix: ix.close())
    assert len(seen) >= 6 and OOM in seen, seen  # the node / root / descendant / normal buffers and the host vectors
    again = Index(ds, None, view=view)
    assert again.search(10, queries=queries, search_k=800) == index.search(10, queries=queries, search_k=800)
    again.close()


def test_search_batch_survives_every_allocation_failure(world):
    _ds, vecs, _seeds, _forest, index, queries = world
    # a warmed-up search allocates nothing (its scratch is kept by the calling thread's context), so every sweep submits a
    # batch larger than any before it: the scratch must grow, and a failed growth leaves it empty for the next attempt
    nq = 256
    for knobs in (dict(), dict(AH_SEARCH_WAVE=0, AH_SEARCH_TILES=0), dict(AH_SEARCH_SCREEN=0)):
        nq *= 2
        big = vecs[np.arange(nq) * 97 % N] + np.float32(1e-3)
        with _lib.tuning(**knobs):
            seen = sweep(lambda: index.search(10, queries=big, search_k=800, raw=True))
            got = index.search(10, queries=big, search_k=800, raw=True)
        # (contexts — stream + scratch — are recycled across datasets and tests of one process: a submission that fits the
        # scratch an earlier test left behind has nothing to allocate, and then there is nothing to fail)
        assert all(s in (OOM, DEVICE) for s in seen), (knobs, seen)
        for lo in range(0, nq, 16):  # the same queries in small submissions (the path the other tests pin to the oracle)
            part = index.search(10, queries=big[lo:lo + 16], search_k=800, raw=True)
            assert np.array_equal(part[0], got[0][lo:lo + 16]) and part[1].tobytes() == got[1][lo:lo + 16].tobytes(), (knobs, lo)
    # with a candidate filter (its own device buffers)
    cand = np.arange(0, N, 3, dtype=np.uint32)
    want_f = index.search(5, queries=queries, search_k=600, candidates=cand, candidates_sorted=True, raw=True)
    big = vecs[np.arange(512) * 89 % N] + np.float32(1e-3)
    sweep(lambda: index.search(5, queries=big, search_k=600, candidates=cand, candidates_sorted=True, raw=True))
    got_f = index.search(5, queries=queries, search_k=600, candidates=cand, candidates_sorted=True, raw=True)
    assert np.array_equal(got_f[0], want_f[0]) and got_f[1].tobytes() == want_f[1].tobytes()


def test_route_items_survives_every_allocation_failure(world):
    _ds, _vecs, seeds, _forest, index, _queries = world
    ids = np.arange(0, N, 7, dtype=np.uint32)
    seen = sweep(lambda: index.route_items(ids, seeds))  # (it lives in the calling thread's scratch: nothing to allocate once that is large enough)
    assert all(s in (OOM, DEVICE) for s in seen), seen
    a, b = index.route_items(ids, seeds), index.route_items(ids[::-1].copy(), seeds)
    assert np.array_equal(a, b[:, ::-1])  # an item's leaf does not depend on the submission it travelled in


def test_rerank_paths_survive_every_allocation_failure(world):
    ds, vecs, _seeds, _forest, _index, queries = world
    od = O.Data(O.COSINE, vecs)
    ids = np.arange(0, N, 2, dtype=np.uint32)
    seen = sweep(lambda: ds.rerank(10, query=queries[0], sorted_ids=ids))
    got = ds.rerank(10, query=queries[0], sorted_ids=ids)
    want = od.rerank(*od.query_leaf(queries[0]), ids, 10)
    assert got[0].tolist() == want[0].tolist() and got[1].tobytes() == want[1].tobytes()
    lists = [ids, ids[::3], ids[5::2], ids[:4000]] * 8
    qs = np.concatenate([queries] * 8)
    seen_b = sweep(lambda: ds.rerank_batch(qs, lists, 10))
    assert all(s in (OOM, DEVICE) for s in seen + seen_b), (seen, seen_b)
    got_b = ds.rerank_batch(qs, lists, 10)
    for i in (0, 1, 2, 3, 17, 31):
        w = od.rerank(*od.query_leaf(qs[i]), lists[i], 10)
        assert got_b[0][i].tolist() == w[0].tolist() and got_b[1][i].tobytes() == w[1].tobytes(), i
    sweep(lambda: ds.distances(query=vecs[3]))
    assert ds.distances(query=vecs[3]).tobytes() == od.distances(*od.query_leaf(vecs[3])).tobytes()


def test_forest_builds_survive_every_allocation_failure(world):
    ds, _vecs, seeds, forest, _index, _queries = world
    want = forest.digest()[0]
    seen = sweep(lambda: ds.build_forest(seeds), cleanup=lambda f: f.close())
    assert OOM in seen, seen
    f = ds.build_forest(seeds)
    assert f.digest()[0] == want
    f.close()
    ref = [forest.canonical(t) for t in range(TREES)]
    seen = sweep(lambda: ds.build_forest_stream(seeds))
    assert OOM in seen, seen
    _roots, _stats, streamed = ds.build_forest_stream(seeds)
    assert [streamed.canonical(t) for t in range(TREES)] == ref
    # the same with the tail of the build in groups of trees (its third node table and the groups' read-back are allocations
    # of their own, made in the middle of the level loop)
    with _lib.tuning(AH_BUILD_TAIL_GROUPS=3, AH_BUILD_TAIL_MIN_MB=0, AH_ROWMAJOR=0):
        f = ds.build_forest(seeds)
        assert f.stats["tail_groups"] == 3 and f.digest()[0] == want
        f.close()
        seen = sweep(lambda: ds.build_forest(seeds), cleanup=lambda f: f.close())
        assert OOM in seen, seen
        f = ds.build_forest(seeds)
        assert f.digest()[0] == want
        f.close()
        seen = sweep(lambda: ds.build_forest_stream(seeds))
        assert OOM in seen, seen
        _roots, stats, streamed = ds.build_forest_stream(seeds)
        assert stats["tail_groups"] == 3 and [streamed.canonical(t) for t in range(TREES)] == ref


def test_dataset_staging_survives_every_allocation_failure(world):
    _ds, vecs, _seeds, _forest, _index, _queries = world
    n = 5000

    def make():
        d = Dataset(D.Cosine, DIMS, n)
        try:
            d.upload_vectors(np.arange(n, dtype=np.uint32), vecs[:n])
            d.finalize()
        except BaseException:
            d.close()
            raise
        return d
    sweep(make, cleanup=lambda d: d.close())
    d = make()
    od = O.Data(O.COSINE, vecs[:n])
    assert d.distances(query=vecs[3]).tobytes() == od.distances(*od.query_leaf(vecs[3])).tobytes()
    d.close()
