"""ah_search_batch against the oracle AT THE SHAPES THAT ARE BENCHMARKED (round-3 review, item 1): the whole
`Reader::nns_by_leaf` (src/reader.rs:317-401) — wave descent, dedup by LDS bitmap / hash set, leaf tiles, candidate
filters — on 1M x 1536 dot product / 20 trees and on the 10M x 768 cosine / 100-tree index, search_k = 10 000,
count = 100, with `ah_index_search_stats` proving which tier served the queries."""
import numpy as np
import pytest

from arroy_amd import _lib
from arroy_amd import distances as D
from oracle import oracle as O

pytestmark = pytest.mark.gpu

COMBOS = [(1, 1), (1, 0), (0, 1), (0, 0)]  # AH_SEARCH_WAVE x AH_SEARCH_TILES


def assert_equals_oracle(got, oracle, forest, queries, picks, count, search_k, cand=None, what=""):
    oi, od, oc = got
    for qi in picks:
        qv, qh = oracle.query_leaf(queries[qi])
        want, _ = O.search(oracle, forest, qv, qh, count, search_k, 0, cand, candidates_sorted=True, want_candidates=False)
        assert int(oc[qi]) == len(want), (what, qi, int(oc[qi]), len(want))
        assert list(oi[qi, :oc[qi]]) == [i for i, _ in want], (what, qi)
        wd = np.array([d for _, d in want], dtype=np.float32)
        assert od[qi, :oc[qi]].view(np.uint32).tolist() == wd.view(np.uint32).tolist(), (what, qi)


def query_sets(vecs, rng, n_each):
    """(clustered, distinct): near copies of a few items (they share every leaf: what the leaf tiles exploit), and items
    that have nothing to do with each other."""
    n, dims = vecs.shape
    bases = rng.choice(n, n_each // 8, replace=False)
    clustered = np.repeat(vecs[bases], 8, axis=0) + rng.standard_normal((n_each, dims)).astype(np.float32) * np.float32(0.05)
    distinct = vecs[rng.choice(n, n_each, replace=False)] + rng.standard_normal((n_each, dims)).astype(np.float32) * np.float32(0.05)
    return clustered.astype(np.float32), distinct.astype(np.float32)


def test_search_equals_oracle_at_the_benchmarked_shape_1m_x_1536_dot_20_trees():
    """BASELINE configs[3] as bench.py's `search` leg runs it: 1M x 1536 dot product (after `preprocess`), 20 trees,
    search_k = 10 000, count = 100."""
    from arroy_amd import Dataset, shard
    n, dims, trees, count, sk = 1_000_000, 1536, 20, 100, 10_000
    ds = Dataset(D.DotProduct, dims, n)
    ds.fill_synthetic(42, 1, n)
    ds.preprocess_dot()
    ds.finalize()
    vecs = O.synth(42, 1, n, dims)
    oracle = O.Data(O.DOT_PRODUCT, vecs)
    oracle.preprocess_dot()
    forest = ds.build_forest(shard.tree_seeds(42, range(trees)))
    index = ds.create_index(forest)
    rng = np.random.default_rng(4)
    clustered, distinct = query_sets(vecs, rng, 64)
    queries = np.concatenate([clustered, distinct])
    nq = len(queries)
    every = range(nq)
    # all four combinations of descent and re-rank: the same bits, and the oracle's
    res = {}
    for wave, tiles in COMBOS:
        with _lib.tuning(AH_SEARCH_WAVE=wave, AH_SEARCH_TILES=tiles):
            index.stats(reset=True)
            res[wave, tiles] = index.search(count, queries=queries, search_k=sk, raw=True)
            st = index.stats()
        assert st["queries"] == nq and st["calls"] == 1, st
        assert st["descent_wave_small"] + st["descent_wave_big"] + st["descent_octet_lds"] + st["descent_octet_global"] == nq, st
        assert st["descent_block"] == 0, st  # 128 queries: above AH_SEARCH_BLOCK_MAX_QUERIES
        if wave:  # the small queues of the wave descent hold an unfiltered search_k = 10 000 query
            assert st["descent_wave_small"] >= 0.9 * nq, st
        else:
            assert st["descent_octet_lds"] == nq and st["descent_wave_small"] + st["descent_wave_big"] == 0, st
        if tiles:  # 1M ids fit the LDS bitmap
            assert st["rerank_tiles"] == nq and st["dedup_flag_bitmap"] == nq and st["fallback_chunks"] == 0, st
            assert st["tile_visits"] > 5 * nq and st["tile_units_16"] + st["tile_units_8"] > 0 and st["tile_units_4"] > 0, st
        else:
            assert st["rerank_sorted"] == nq and st["dedup_sorted_bitmap"] == nq and st["rerank_tiles"] == 0, st
    for combo in COMBOS[1:]:
        for a, b in zip(res[1, 1], res[combo]):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), combo
    assert_equals_oracle(res[1, 1], oracle, forest, queries, every, count, sk, what="unfiltered")
    assert int(res[1, 1][2].min()) == count
    # the certified top-k screen of the tile re-rank (binary16 rows first, f32 for the survivors): on by default above
    index.stats(reset=True)
    index.search(count, queries=queries, search_k=sk, raw=True)
    st = index.stats()
    assert st["rerank_screened"] == nq and count * nq <= st["screen_survivors"] <= 0.2 * sk * nq, st
    # ... and, round 6, on the int8 copy of the rows before that (a big submission): the same bits with the int8 stage off,
    # fewer survivors then (the binary16 bound is tighter), and the counters say which stage served the queries
    assert st["rerank_screened8"] == nq and st["screen8_retried_chunks"] == 0, st
    survivors8 = st["screen_survivors"]
    with _lib.tuning(AH_SEARCH_SCREEN8=0):
        index.stats(reset=True)
        half = index.search(count, queries=queries, search_k=sk, raw=True)
        st = index.stats()
    assert st["rerank_screened"] == nq and st["rerank_screened8"] == 0 and count * nq <= st["screen_survivors"] < survivors8, st
    for a, b in zip(res[1, 1], half):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "int8 stage off"
    with _lib.tuning(AH_SEARCH_SCREEN=0):
        index.stats(reset=True)
        plain = index.search(count, queries=queries, search_k=sk, raw=True)
        st = index.stats()
    assert st["rerank_screened"] == 0 and st["rerank_tiles"] == nq, st
    for a, b in zip(res[1, 1], plain):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "screen off"
    # by_item: the stored leaves (header included) as queries
    items = rng.choice(n, 32, replace=False).astype(np.uint32)
    gi = index.search(count, items=items, search_k=sk, raw=True)
    for qi in range(len(items)):
        qv, qh = oracle.item_leaf(int(items[qi]))
        want, _ = O.search(oracle, forest, qv, qh, count, sk, want_candidates=False)
        assert list(gi[0][qi]) == [i for i, _ in want]
        assert gi[1][qi].view(np.uint32).tolist() == np.array([d for _, d in want], np.float32).view(np.uint32).tolist()
    # the descent of the SMALL submissions (one block of 32 octets per query, k_descend_block): the same 128 queries through it,
    # as one call and one query per call (arroy's own API shape, src/reader.rs:46-75) — the bits of the wave descent
    with _lib.tuning(AH_SEARCH_BLOCK_MAX_QUERIES=1024):
        index.stats(reset=True)
        blk = index.search(count, queries=queries, search_k=sk, raw=True)
        st = index.stats()
    assert st["descent_block"] >= 0.9 * nq and st["descent_block"] + st["descent_wave_big"] + st["descent_octet_lds"] + \
        st["descent_octet_global"] == nq and st["descent_wave_small"] == 0, st
    for a, b in zip(res[1, 1], blk):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    index.stats(reset=True)
    for qi in (0, 17, 64, 99, 127):
        one = index.search(count, queries=queries[qi:qi + 1], search_k=sk, raw=True)
        assert np.array_equal(one[0][0], res[1, 1][0][qi]) and one[1][0].tobytes() == res[1, 1][1][qi].tobytes(), qi
    assert index.stats()["descent_block"] == 5
    # `QueryBuilder::candidates`: half, 10 % and 3 % of the items
    for share in (0.5, 0.10, 0.03):
        keep = np.sort(rng.choice(n, int(n * share), replace=False)).astype(np.uint32)
        index.stats(reset=True)
        got = index.search(count, queries=queries, search_k=sk, candidates=keep, candidates_sorted=True, raw=True)
        st = index.stats()
        assert st["filtered_queries"] == nq, (share, st)
        served = st["descent_wave_small"] + st["descent_wave_big"] + st["descent_octet_lds"] + st["descent_octet_global"]
        assert served == nq and st["rerank_tiles"] + st["rerank_sorted"] == nq, (share, st)
        if share == 0.5:    # the small queues first, the kept count of every leaf from one pass per submission
            assert st["descent_wave_small"] >= 0.75 * nq and st["leaf_kept_passes"] == 1 and st["rerank_tiles"] == nq, st
        elif share == 0.10:  # a query pops ~10x the nodes: straight to the big queues
            assert st["descent_wave_small"] == 0 and st["descent_wave_big"] > 0 and st["leaf_kept_passes"] == 1, st
        else:               # under 5 %: the sequential descent, no pass over the forest
            assert st["descent_wave_small"] + st["descent_wave_big"] == 0 and st["leaf_kept_passes"] == 0, st
        with _lib.tuning(AH_SEARCH_WAVE=0, AH_SEARCH_TILES=0):
            slow = index.search(count, queries=queries, search_k=sk, candidates=keep, candidates_sorted=True, raw=True)
        if share == 0.5:  # ... and the block descent under the same filter
            with _lib.tuning(AH_SEARCH_BLOCK_MAX_QUERIES=1024):
                index.stats(reset=True)
                blk = index.search(count, queries=queries, search_k=sk, candidates=keep, candidates_sorted=True, raw=True)
                assert index.stats()["descent_block"] >= 0.75 * nq
            for a, b in zip(got, blk):
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), share
        for a, b in zip(got, slow):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), share
        assert_equals_oracle(got, oracle, forest, queries, range(0, nq, 4), count, sk, cand=keep, what=f"filter {share}")
        assert np.isin(got[0][got[0] != 0xFFFFFFFF], keep).all()
    index.close()
    forest.close()
    ds.close()


def test_search_equals_oracle_on_the_headline_index_10m_x_768_cosine_100_trees():
    """The forest of BASELINE configs[2] (10M x 768 cosine, 100 trees) searched on the device: ids beyond the LDS bitmap
    (hash-set dedup), 3.4 M nodes, 1e9 descendant ids."""
    from arroy_amd import Dataset, shard
    n, dims, trees, count, sk = 10_000_000, 768, 100, 100, 10_000
    ds = Dataset(D.Cosine, dims, n)
    ds.fill_synthetic(42, 1, n)
    ds.finalize()
    forest = ds.build_forest(shard.tree_seeds(42, range(trees)))
    index = ds.create_index(forest)
    vecs = O.synth(42, 1, n, dims)
    oracle = O.Data(O.COSINE, vecs)
    rng = np.random.default_rng(5)
    clustered, distinct = query_sets(vecs, rng, 16)
    queries = np.concatenate([clustered, distinct])
    nq = len(queries)
    res = {}
    for wave, tiles in COMBOS:
        with _lib.tuning(AH_SEARCH_WAVE=wave, AH_SEARCH_TILES=tiles):
            index.stats(reset=True)
            res[wave, tiles] = index.search(count, queries=queries, search_k=sk, raw=True)
            st = index.stats()
        assert st["queries"] == nq, st
        if wave:  # (32 queries: a block per query)
            assert st["descent_block"] + st["descent_wave_big"] >= 0.9 * nq and st["descent_wave_small"] == 0, st
        if tiles:  # 10M ids: the hash set of the candidates, not the bitmap
            assert st["rerank_tiles"] == nq and st["dedup_flag_hash"] == nq and st["dedup_flag_bitmap"] == 0, st
        else:
            assert st["rerank_sorted"] == nq and st["dedup_sort_lds"] + st["dedup_sort_global"] == nq, st
    for combo in COMBOS[1:]:
        for a, b in zip(res[1, 1], res[combo]):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), combo
    assert_equals_oracle(res[1, 1], oracle, forest, queries, range(nq), count, sk, what="10M unfiltered")
    index.stats(reset=True)
    index.search(count, queries=queries, search_k=sk, raw=True)
    st = index.stats()
    assert st["rerank_screened"] == nq and count * nq <= st["screen_survivors"] <= 0.2 * sk * nq, st
    # the int8 first stage (round 6) serves the BIG submissions: the 32 queries three times over in one call (Cosine: the
    # candidate's stored norm and its row scale both travel with the screen value) — the bits of the 32-query call
    # arroy's own shape on the headline index: ONE query a call (and five) — 100 trees dealt over 13 blocks of one descent wave each
    # (`k_descend_multi`), the single query's ids copied by its flat tile launch — the rows of the 32-query call, bit for bit
    index.stats(reset=True)
    for qi in (0, 13, nq - 1):
        one = index.search(count, queries=queries[qi:qi + 1], search_k=sk, raw=True)
        for a, b in zip(res[1, 1], one):
            assert np.array_equal(a[qi:qi + 1].view(np.uint32), b.view(np.uint32)), f"one query a call: {qi}"
    five = index.search(count, queries=queries[7:12], search_k=sk, raw=True)
    for a, b in zip(res[1, 1], five):
        assert np.array_equal(a[7:12].view(np.uint32), b.view(np.uint32)), "five queries a call"
    st = index.stats()
    assert st["descent_multi"] == 8 and st["fallback_chunks"] == 0, st
    big = np.concatenate([queries, queries, queries])
    index.stats(reset=True)
    got8 = index.search(count, queries=big, search_k=sk, raw=True)
    st = index.stats()
    assert st["rerank_screened8"] + 96 * st["screen8_retried_chunks"] == 96 and st["rerank_screened"] == 96, st
    for rep in range(3):
        for a, b in zip(res[1, 1], got8):
            assert np.array_equal(a.view(np.uint32), b[rep * nq:(rep + 1) * nq].view(np.uint32)), "int8 first stage"
    with _lib.tuning(AH_SEARCH_SCREEN=0):
        plain = index.search(count, queries=queries, search_k=sk, raw=True)
    for a, b in zip(res[1, 1], plain):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "screen off"
    keep = np.sort(rng.choice(n, n // 2, replace=False)).astype(np.uint32)
    index.stats(reset=True)
    got = index.search(count, queries=queries, search_k=sk, candidates=keep, candidates_sorted=True, raw=True)
    st = index.stats()
    assert st["leaf_kept_passes"] == 1 and st["rerank_tiles"] == nq, st
    assert_equals_oracle(got, oracle, forest, queries, range(0, nq, 2), count, sk, cand=keep, what="10M filter half")
    # a submission too small to pay for the pass over 1e9 descendant ids takes the sequential descent
    index.stats(reset=True)
    few = index.search(count, queries=queries[:4], search_k=sk, candidates=keep, candidates_sorted=True, raw=True)
    st = index.stats()
    assert st["leaf_kept_passes"] == 0 and st["descent_octet_lds"] + st["descent_octet_global"] == 4, st
    for a, b in zip(few, got):
        assert np.array_equal(a.view(np.uint32), b[:4].view(np.uint32))
    index.close()
    forest.close()
    ds.close()
