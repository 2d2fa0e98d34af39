"""The SMALL submissions (arroy's API is one query per call: `QueryBuilder::by_vector` -> `nns_by_leaf`, src/reader.rs:46-75,
317-401) take their own device path since round 5 — a block of 32 octets descends (and prepares the query leaf from the caller's
pinned buffer), one block places the leaf visits and makes the binary16 copies, the leaf tiles keep whole rows in flight, the
selection flags duplicates itself and writes results and status straight into pinned memory; `ah_rerank_by_vector` of a short list
selects with one launch (k_topk_small).  Every one of those has a switch: the answers with any of them off, all of them off, and
the oracle's must be the same bits; and what the small kernels cannot hold must fall back, not truncate."""
import numpy as np
import pytest

from arroy_amd import _lib
from arroy_amd import distances as D
from oracle import oracle as O

pytestmark = pytest.mark.gpu

SMALL_KNOBS = ["AH_SEARCH_BLOCK_MAX_QUERIES", "AH_SEARCH_SMALL_UNITS_MAX_QUERIES", "AH_SEARCH_SMALL_TILES_MAX_QUERIES",
               "AH_SEARCH_FUSED_FLAG", "AH_SEARCH_FUSED_PREPARE", "AH_SEARCH_SINGLE_FUSED", "AH_SEARCH_MULTI",
               # round 6: who copies a single query's ids, the tile launch's grid (flat list for one query, item list for a few), the
               # status block wiped by the selection, the wait on the pinned status word
               "AH_SEARCH_MULTI_IDS_BY_TILES", "AH_SEARCH_FLAT_TILES", "AH_SEARCH_ITEM_LIST", "AH_SEARCH_STATUS_WIPE", "AH_SEARCH_SPIN_WAIT",
               "AH_SEARCH_MULTI_OWN_UNITS"]


def same(a, b):
    return np.array_equal(a[0], b[0]) and a[1].tobytes() == b[1].tobytes() and np.array_equal(a[2], b[2])


@pytest.fixture(scope="module", params=[(D.DotProduct, O.DOT_PRODUCT), (D.Cosine, O.COSINE), (D.Euclidean, O.EUCLIDEAN)],
                ids=["dot", "cosine", "euclidean"])
def world(request):
    from arroy_amd import Dataset, shard
    metric, ometric = request.param
    n, dims, trees = 60_000, 200, 12
    vecs = O.synth(7, 2, n, dims)
    ds = Dataset(metric, dims, n)
    ds.upload_vectors(np.arange(n, dtype=np.uint32), vecs)
    od = O.Data(ometric, vecs)
    if metric is D.DotProduct:
        ds.preprocess_dot()
        od.preprocess_dot()
    ds.finalize()
    forest = ds.build_forest(shard.tree_seeds(7, range(trees)))
    index = ds.create_index(forest)
    rng = np.random.default_rng(3)
    queries = (vecs[rng.choice(n, 80, replace=False)] + rng.standard_normal((80, dims)).astype(np.float32) * np.float32(0.1)).astype(np.float32)
    yield ds, od, forest, index, queries, vecs
    index.close()
    forest.close()
    ds.close()


def oracle_search(od, forest, q, count, sk, cand=None):
    qv, qh = od.query_leaf(q)
    want, _ = O.search(od, forest, qv, qh, count, sk, 0, cand, candidates_sorted=True, want_candidates=False)
    return want


def test_small_submissions_equal_the_oracle_and_every_switch_setting(world):
    _ds, od, forest, index, queries, _vecs = world
    count, sk = 25, 1500
    for nq in (1, 2, 7, 8, 9, 33, 64):
        qs = queries[:nq]
        index.stats(reset=True)
        base = index.search(count, queries=qs, search_k=sk, raw=True)
        st = index.stats()
        assert st["descent_block"] == nq and st["rerank_tiles"] == nq and st["fallback_chunks"] == 0, (nq, st)
        # round 6: up to 32 queries a call, the 12 trees of a query are dealt over two blocks (k_descend_multi)
        assert st["descent_multi"] == (nq if nq <= 32 else 0), (nq, st)
        for qi in range(nq):
            want = oracle_search(od, forest, qs[qi], count, sk)
            assert int(base[2][qi]) == len(want) and list(base[0][qi, :len(want)]) == [i for i, _ in want], (nq, qi)
            assert base[1][qi, :len(want)].tobytes() == np.array([d for _, d in want], dtype=np.float32).tobytes(), (nq, qi)
        for knob in SMALL_KNOBS:  # one off at a time
            with _lib.tuning(**{knob: 0}):
                index.stats(reset=True)
                assert same(index.search(count, queries=qs, search_k=sk, raw=True), base), (nq, knob)
                # ... and none of the combinations is "saved" by the fall-back (round 6: block descent on + k_units_small off
                # used to screen against binary16 copies of query leaves nobody had prepared yet)
                assert index.stats()["fallback_chunks"] == 0, (nq, knob, index.stats())
        with _lib.tuning(**{k: 0 for k in SMALL_KNOBS}):  # the big submissions' path on the same queries
            assert same(index.search(count, queries=qs, search_k=sk, raw=True), base), nq
        # the same queries one per call
        for qi in (0, nq - 1):
            one = index.search(count, queries=qs[qi:qi + 1], search_k=sk, raw=True)
            assert np.array_equal(one[0][0], base[0][qi]) and one[1][0].tobytes() == base[1][qi].tobytes()


def test_small_submissions_by_item_and_under_a_filter(world):
    _ds, od, forest, index, queries, vecs = world
    n = vecs.shape[0]
    count, sk = 10, 800
    items = np.array([3, 59_999, 1234, 777, 31_000], dtype=np.uint32)
    got = index.search(count, items=items, search_k=sk, raw=True)
    with _lib.tuning(**{k: 0 for k in SMALL_KNOBS}):
        assert same(index.search(count, items=items, search_k=sk, raw=True), got)
    for share in (0.5, 0.05):
        cand = np.arange(0, n, int(1 / share), dtype=np.uint32)
        got = index.search(count, queries=queries[:5], search_k=sk, candidates=cand, candidates_sorted=True, raw=True)
        for qi in range(5):
            want = oracle_search(od, forest, queries[qi], count, sk, cand)
            assert list(got[0][qi, :got[2][qi]]) == [i for i, _ in want], (share, qi)
        with _lib.tuning(**{k: 0 for k in SMALL_KNOBS}):
            assert same(index.search(count, queries=queries[:5], search_k=sk, candidates=cand, candidates_sorted=True, raw=True), got)


def test_one_query_on_several_compute_units_many_trees_repeated_calls_and_overflow():
    """k_descend_multi (round 6): 44 trees -> six blocks of one descent wave per query.  The control block the blocks talk through
    is wiped by the last block of every query: the same call twenty times over gives the same bits (a stale slot would add a
    phantom leaf).  A search_k that makes a queue pop more than the 32 leaves a list holds raises the failure word: every block
    leaves, the submission is redone the long way, and the next small call finds the control block clean."""
    from arroy_amd import Dataset, shard
    n, dims, trees = 40_000, 64, 44
    vecs = O.synth(11, 2, n, dims)
    ds = Dataset(D.Cosine, dims, n)
    ds.upload_vectors(np.arange(n, dtype=np.uint32), vecs)
    ds.finalize()
    od = O.Data(O.COSINE, vecs)
    forest = ds.build_forest(shard.tree_seeds(11, range(trees)))
    index = ds.create_index(forest)
    rng = np.random.default_rng(5)
    queries = (vecs[rng.choice(n, 24, replace=False)] + rng.standard_normal((24, dims)).astype(np.float32) * np.float32(0.05)).astype(np.float32)
    count, sk = 20, 2000
    try:
        for nq in (1, 3, 8):
            index.stats(reset=True)
            base = index.search(count, queries=queries[:nq], search_k=sk, raw=True)
            st = index.stats()
            assert st["descent_multi"] == nq and st["fallback_chunks"] == 0, (nq, st)
            for qi in range(nq):
                want = oracle_search(od, forest, queries[qi], count, sk)
                assert list(base[0][qi, :base[2][qi]]) == [i for i, _ in want], (nq, qi)
                assert base[1][qi, :len(want)].tobytes() == np.array([d for _, d in want], dtype=np.float32).tobytes(), (nq, qi)
            for _ in range(20):
                assert same(index.search(count, queries=queries[:nq], search_k=sk, raw=True), base), nq
            with _lib.tuning(AH_SEARCH_MULTI=0):
                assert same(index.search(count, queries=queries[:nq], search_k=sk, raw=True), base), nq
        # different queries call after call (nothing of the previous call may survive in the control block)
        for i in range(16):
            got = index.search(count, queries=queries[i:i + 2], search_k=sk, raw=True)
            with _lib.tuning(AH_SEARCH_MULTI=0):
                assert same(index.search(count, queries=queries[i:i + 2], search_k=sk, raw=True), got), i
        # overflow: leaves of <= 64 ids, search_k = n / 2 -> hundreds of leaves per tree
        with _lib.tuning(AH_SEARCH_SMALL_GATE=0):
            index.stats(reset=True)
            big = index.search(count, queries=queries[:2], search_k=n // 2, raw=True)
            st = index.stats()
            assert st["fallback_chunks"] >= 1 and st["fallback_queue"] >= 1 and st["descent_multi"] == 0, st
        for qi in range(2):
            want = oracle_search(od, forest, queries[qi], count, n // 2)
            assert list(big[0][qi, :big[2][qi]]) == [i for i, _ in want], qi
        index.stats(reset=True)
        again = index.search(count, queries=queries[:3], search_k=sk, raw=True)
        assert index.stats()["descent_multi"] == 3
        with _lib.tuning(AH_SEARCH_MULTI=0):
            assert same(index.search(count, queries=queries[:3], search_k=sk, raw=True), again)
        # more leaves than the one-scan item list of the single query's tile launch covers (128): every block walks the visits
        index.stats(reset=True)
        wide = index.search(count, queries=queries[:1], search_k=8000, raw=True)
        st = index.stats()
        assert st["descent_multi"] == 1 and st["fallback_chunks"] == 0 and st["tile_visits"] > 128, st
        want = oracle_search(od, forest, queries[0], count, 8000)
        assert list(wide[0][0, :wide[2][0]]) == [i for i, _ in want]
        assert wide[1][0, :len(want)].tobytes() == np.array([d for _, d in want], dtype=np.float32).tobytes()
        with _lib.tuning(AH_SEARCH_FLAT_TILES=0):
            assert same(index.search(count, queries=queries[:1], search_k=8000, raw=True), wide)
        # ... every time: the blocks' agreement on when to stop must keep up with a descent that opens 170 leaves (its first version
        # compared every known leaf with every other; one pass fell behind the descent, the lists grew until a queue's overflowed,
        # and this very call took the long way every other time — after calls that had failed, or not)
        for rep in range(40):
            with _lib.tuning(AH_SEARCH_SMALL_GATE=0):
                index.search(count, queries=queries[:2], search_k=n // 2, raw=True)
            index.stats(reset=True)
            assert same(index.search(count, queries=queries[:1], search_k=8000, raw=True), wide), rep
            st = index.stats()
            assert st["descent_multi"] == 1 and st["fallback_chunks"] == 0, (rep, st)
        # under a filter (the kept ids of a leaf are copied octet by octet) and by item
        cand = np.arange(0, n, 2, dtype=np.uint32)
        got = index.search(count, queries=queries[:4], search_k=sk, candidates=cand, candidates_sorted=True, raw=True)
        for qi in range(4):
            want = oracle_search(od, forest, queries[qi], count, sk, cand)
            assert list(got[0][qi, :got[2][qi]]) == [i for i, _ in want], qi
        items = np.array([5, 39_999, 1234], dtype=np.uint32)
        got = index.search(count, items=items, search_k=sk, raw=True)
        with _lib.tuning(AH_SEARCH_MULTI=0):
            assert same(index.search(count, items=items, search_k=sk, raw=True), got)
    finally:
        index.close()
        forest.close()
        ds.close()


def test_one_query_calls_from_many_threads_at_once(world):
    """`Reader: Sync` — arroy's readers call `nns_by_vector` from many threads.  Every call leases a context of the dataset (its
    own stream, scratch, pinned buffer and — round 6 — control block of `k_descend_multi`); a context serves one thread after the
    other, so the status block a call finds "wiped by the previous small submission" was wiped by ANOTHER thread's call, the host
    polls a pinned word while seven other streams are busy, and calls of one and of three queries alternate on the same scratch.
    Eight threads x 60 calls against the answers of the big submissions' path."""
    import threading
    _ds, _od, _forest, index, queries, _vecs = world
    count, sk = 25, 1500
    with _lib.tuning(**{k: 0 for k in SMALL_KNOBS}):
        want = index.search(count, queries=queries, search_k=sk, raw=True)
    errors = []

    def worker(tid):
        try:
            for rep in range(60):
                nq = 1 if (rep + tid) % 3 else 3
                a = (tid * 11 + rep * 5) % (len(queries) - nq)
                got = index.search(count, queries=queries[a:a + nq], search_k=sk, raw=True)
                ok = np.array_equal(got[0], want[0][a:a + nq]) and got[1].tobytes() == want[1][a:a + nq].tobytes() and \
                    np.array_equal(got[2], want[2][a:a + nq])
                if not ok:
                    errors.append((tid, rep, nq, a))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    index.stats(reset=True)
    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:4]
    st = index.stats()
    assert st["fallback_chunks"] == 0 and st["descent_multi"] == st["queries"] == 8 * 60 + 8 * 20 * 2, st


def test_more_visits_than_the_one_block_unit_builder_holds_fall_back(world):
    """k_units_small sorts at most 2048 leaf visits: 64 queries with a search_k that makes each of them pop more than 32 leaves
    overflow it — bit 5 of the status word, the submission is redone on the sorted path, the answers are the oracle's."""
    _ds, od, forest, index, queries, vecs = world
    n = vecs.shape[0]
    count, sk = 10, n // 3
    with _lib.tuning(AH_SEARCH_SMALL_GATE=0):  # as before round 6: the call starts on the small kernels whatever it will open
        index.stats(reset=True)
        got = index.search(count, queries=queries[:64], search_k=sk, raw=True)
        st = index.stats()
    assert st["fallback_chunks"] >= 1 or st["tile_visits"] <= 2048, st
    # default: the host's estimate of the leaves a query opens (search_k / mean leaf + trees) sends such a call past the small
    # kernels from the start — the same answers, no chunk redone (round-5 advice)
    index.stats(reset=True)
    gated = index.search(count, queries=queries[:64], search_k=sk, raw=True)
    st = index.stats()
    assert st["fallback_chunks"] == 0 and st["tile_visits"] > 2048, st
    assert all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(got, gated))
    for qi in (0, 17, 63):
        want = oracle_search(od, forest, queries[qi], count, sk)
        assert list(got[0][qi, :got[2][qi]]) == [i for i, _ in want], qi
        assert got[1][qi, :len(want)].tobytes() == np.array([d for _, d in want], dtype=np.float32).tobytes(), qi


def test_rerank_of_one_short_list_one_launch_equals_the_general_path_and_the_oracle(world):
    ds, od, _forest, _index, queries, vecs = world
    n = vecs.shape[0]
    rng = np.random.default_rng(11)
    for n_ids, k in ((1, 1), (5, 10), (1000, 100), (10_000, 100), (16_384, 1024), (16_385, 10), (12_000, 1025)):
        ids = np.sort(rng.choice(n, n_ids, replace=False)).astype(np.uint32)
        for q in (queries[0], vecs[int(ids[0])]):
            got = ds.rerank(k, query=q, sorted_ids=ids)
            want = od.rerank(*od.query_leaf(q), ids, k)
            assert got[0].tolist() == want[0].tolist() and got[1].tobytes() == want[1].tobytes(), (n_ids, k)
            with _lib.tuning(AH_RERANK_SMALL=0):
                gen = ds.rerank(k, query=q, sorted_ids=ids)
            assert gen[0].tolist() == got[0].tolist() and gen[1].tobytes() == got[1].tobytes(), (n_ids, k)
    # by item, and "all items" of a small dataset
    got = ds.rerank(7, item=4242, sorted_ids=np.arange(0, n, 9, dtype=np.uint32))
    want = od.rerank(*od.item_leaf(4242), np.arange(0, n, 9, dtype=np.uint32), 7) if hasattr(od, "item_leaf") else None
    if want is not None:
        assert got[0].tolist() == want[0].tolist() and got[1].tobytes() == want[1].tobytes()
    # unsorted / repeated ids are still refused
    with pytest.raises(_lib.ArroyHipError):
        ds.rerank(3, query=queries[0], sorted_ids=np.array([5, 4, 9], dtype=np.uint32))
    with pytest.raises(_lib.ArroyHipError):
        ds.rerank(3, query=queries[0], sorted_ids=np.array([5, 5, 9], dtype=np.uint32))
    with pytest.raises(_lib.ArroyHipError):  # an id that is not stored
        ds.rerank(3, query=queries[0], sorted_ids=np.array([5, n + 10], dtype=np.uint32))


def test_rerank_one_launch_falls_back_on_ties_beyond_its_capacity_and_on_non_finite_distances():
    """2000 copies of one vector: all distances equal, the k-th key's bin holds more than 1024 keys — k_topk_small raises bit 3
    and the general selection answers.  A query with an inf component: non-finite distances, bit 2, the reference's rule on
    positions (src/reader.rs:611-621) is the general path's business."""
    from arroy_amd import Dataset
    n, dims = 3000, 64
    vecs = O.synth(5, 1, n, dims)
    vecs[:2000] = vecs[0]
    for metric, ometric in ((D.Euclidean, O.EUCLIDEAN), (D.DotProduct, O.DOT_PRODUCT)):
        ds = Dataset(metric, dims, n)
        ds.upload_vectors(np.arange(n, dtype=np.uint32), vecs)
        ds.finalize()
        od = O.Data(ometric, vecs)
        ids = np.arange(n, dtype=np.uint32)
        for k in (10, 1500):
            got = ds.rerank(k, query=vecs[1], sorted_ids=ids)
            want = od.rerank(*od.query_leaf(vecs[1]), ids, k)
            assert got[0].tolist() == want[0].tolist() and got[1].tobytes() == want[1].tobytes(), k
        bad = vecs[5].copy()
        bad[3] = np.inf
        got = ds.rerank(20, query=bad, sorted_ids=ids)
        with _lib.tuning(AH_RERANK_SMALL=0):
            gen = ds.rerank(20, query=bad, sorted_ids=ids)
        assert got[0].tolist() == gen[0].tolist() and got[1].tobytes() == gen[1].tobytes()
        want = od.rerank(*od.query_leaf(bad), ids, 20)
        assert got[0].tolist() == want[0].tolist()
        ds.close()


def test_five_thousand_one_query_calls_in_a_row_never_read_a_result_early():
    """The one-query call ends with the host polling a status word its last kernel writes into pinned memory AFTER the results
    (system-scope release), and starts without a memset because the previous call's selection wiped the status block: a result read
    before it landed, or a stale word, would show as a mismatch somewhere in a few thousand back-to-back calls
    (scripts/stress_one_query.py: 30 000 calls, 0 mismatches, on the round's box)."""
    import subprocess
    import sys

    from conftest import ROOT
    out = subprocess.run([sys.executable, "scripts/stress_one_query.py", "5000"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and " 0 mismatches" in out.stdout, out.stdout[-500:] + out.stderr[-500:]
