"""Pin the CPU oracle against the reference's own golden vectors.

Every expectation below is a value asserted by a test *inside* /root/reference; the fixture
file tests/golden/reference_golden.json was extracted from the reference's binary test assets
by tests/golden/make_golden.py.  No GPU needed.
"""
import math

import numpy as np
import pytest

from conftest import hex_f32
from oracle import oracle as O


def rust_display_f32(x: float) -> str:
    """Rust `{}` for f32 = shortest decimal that round-trips (what insta snapshots print)."""
    x = np.float32(x)
    if x == np.floor(x) and abs(x) < 1e16:
        return str(int(x))
    for prec in range(1, 12):
        s = np.format_float_positional(x, precision=prec, unique=False, trim="-")
        if np.float32(float(s)) == x:
            return np.format_float_positional(x, unique=True, trim="-")
    return repr(float(x))


# --- src/spaces/simple_avx.rs:112-153, simple_sse.rs:112-151 -----------------------------------

def test_avx_unit_vectors_simd_equals_scalar(golden):
    v1, v2 = hex_f32(golden["avx_unit"]["v1_hex"]), hex_f32(golden["avx_unit"]["v2_hex"])
    assert v1.size == 70
    for tier in ("avx_emul", "avx_real", "auto"):
        if tier == "avx_real" and not O.lib().ao_cpu_has_avx_fma():
            continue
        assert O.euclid(v1, v2, tier) == O.euclid(v1, v2, "scalar")
        assert O.dot(v1, v2, tier) == O.dot(v1, v2, "scalar")


def test_sse_unit_vectors_simd_equals_scalar(golden):
    v1, v2 = hex_f32(golden["sse_unit"]["v1_hex"]), hex_f32(golden["sse_unit"]["v2_hex"])
    assert v1.size == 54
    assert O.euclid(v1, v2, "sse") == O.euclid(v1, v2, "scalar")
    assert O.dot(v1, v2, "sse") == O.dot(v1, v2, "scalar")


def test_avx_emulation_matches_real_instructions():
    if not O.lib().ao_cpu_has_avx_fma():
        pytest.skip("host has no AVX+FMA")
    rng = np.random.default_rng(7)
    for d in (32, 33, 63, 64, 70, 128, 767, 768, 1536):
        for _ in range(20):
            u = rng.standard_normal(d).astype(np.float32)
            v = rng.standard_normal(d).astype(np.float32)
            assert O.dot(u, v, "avx_emul").tobytes() == O.dot(u, v, "avx_real").tobytes()
            assert O.euclid(u, v, "avx_emul").tobytes() == O.euclid(u, v, "avx_real").tobytes()


def test_dispatch_thresholds():
    """src/spaces/simple.rs:10,17: AVX from 32 dims, SSE from 16, scalar below."""
    rng = np.random.default_rng(3)
    for d, tier in ((8, "scalar"), (15, "scalar"), (16, "sse"), (31, "sse"), (32, "avx_emul"), (100, "avx_emul")):
        u = rng.standard_normal(d).astype(np.float32)
        v = rng.standard_normal(d).astype(np.float32)
        assert O.dot(u, v).tobytes() == O.dot(u, v, tier).tobytes()
        assert O.euclid(u, v).tobytes() == O.euclid(u, v, tier).tobytes()


# --- src/tests/upgrade.rs:58-67 and :116-128 with the LMDB fixtures -----------------------------

def _fixture(golden, name):
    g = golden[name]
    vecs = np.stack([hex_f32(h) for h in g["vectors_hex"]])
    hdrs = np.stack([hex_f32(h) for h in g["headers_hex"]])
    return g, O.Data(O.EUCLIDEAN, vecs, ids=g["ids"], headers=hdrs)


def test_large_mdb_item_vector(golden):
    g, data = _fixture(golden, "large_v0_6")
    v0 = hex_f32(g["vectors_hex"][0])
    assert [rust_display_f32(x) for x in v0[:4]] == g["item0_prefix"]


@pytest.mark.parametrize("name", ["large_v0_6", "smol_v0_6"])
def test_golden_nns(golden, name):
    g, data = _fixture(golden, name)
    q, qh = data.query_leaf(np.array(g["query"], dtype=np.float32))
    # search_k=100 >= n_items: every item is a candidate (src/reader.rs:341-374), ids sorted+deduped
    ids, dists = data.rerank(q, qh, None, g["count"])
    got = [[int(i), rust_display_f32(d)] for i, d in zip(ids, dists)]
    assert got == g["expected"]


# --- src/unaligned_vector/binary_quantized_test.rs -----------------------------------------------

def _bits(b):
    return ["{:08b}".format(x) for x in b]


def test_bq_from_slice_bit_pattern():
    """binary_quantized_test.rs:11-27 (test_from_slice)."""
    b = O.bq_quantize([0.1, 0.2, -0.3, 0.4, -0.5, 0.6, -0.7, 0.8, -0.9])
    assert _bits(b) == ["10101011"] + ["00000000"] * 7
    y = O.bq_dequantize(b)  # :29-60 test_to_vec_iter
    assert list(y[:9]) == [1.0, 1.0, -1.0, 1.0, -1.0, 1.0, -1.0, 1.0, -1.0]


def test_bq_unaligned_smol_and_large():
    """binary_quantized_test.rs:99-125 and :127-167."""
    b = O.bq_quantize([-1.0, 2.0, -3.0, 4.0, 5.0])
    assert _bits(b) == ["00011010"] + ["00000000"] * 7
    assert list(O.bq_dequantize(b)[:5]) == [-1.0, 1.0, -1.0, 1.0, 1.0]
    original = np.array([-1.0 if (n % 3 == 0 or n % 5 == 0) else 1.0 for n in range(100)], dtype=np.float32)
    b = O.bq_quantize(original)
    assert _bits(b) == ["10010110", "01101001", "11001011", "10110100", "01100101", "11011010", "00110010",
                        "01101101", "10011001", "10110110", "01001100", "01011011", "00000110", "00000000",
                        "00000000", "00000000"]
    assert np.array_equal(O.bq_dequantize(b)[:100], original)


def test_bq_zero_signs():
    """src/tests/binary_quantized.rs:22-42: 0.0 -> +1, -0.1 -> -1 (is_sign_positive)."""
    b = O.bq_quantize([0.0, -0.0, -0.1, 0.1])
    assert int.from_bytes(b.tobytes(), "little") == 0b1001


def test_bq_basic_and_padding():
    """binary_quantized_test.rs:102-115,131-154: len rounds to 64, padding decodes to -1."""
    x = np.array([1.0, -1.0, 0.5, -0.5, 0.0], dtype=np.float32)
    b = O.bq_quantize(x)
    y = O.bq_dequantize(b)
    assert y.size == 64
    assert list(y[:5]) == [1.0, -1.0, 1.0, -1.0, 1.0]
    assert np.all(y[5:] == -1.0)
    x = np.ones(65, dtype=np.float32)
    b = O.bq_quantize(x)
    assert b.size == 16
    assert int.from_bytes(b[:8].tobytes(), "little") == 2**64 - 1
    assert int.from_bytes(b[8:].tobytes(), "little") == 1


def test_bq_roundtrip_property():
    """binary_quantized_test.rs:169-190 proptest: from_slice/to_vec round trip on -50..50.2."""
    rng = np.random.default_rng(11)
    for n in (0, 1, 63, 64, 65, 200, 515):
        x = rng.uniform(-50, 50.2, n).astype(np.float32)
        y = O.bq_dequantize(O.bq_quantize(x))
        assert y.size == ((n + 63) // 64) * 64
        assert np.array_equal(y[:n], np.where(np.signbit(x), -1.0, 1.0).astype(np.float32))


# --- src/tests/reader.rs -------------------------------------------------------------------------

def test_two_dimension_on_a_line():
    """reader.rs:101-144: items (i, 0) for i in 0..100, query [50,0] -> ids 50,49,51,48,52 dist 0,1,1,2,2."""
    vecs = np.array([[float(i), 0.0] for i in range(100)], dtype=np.float32)
    data = O.Data(O.EUCLIDEAN, vecs)
    q, qh = data.query_leaf(np.array([50.0, 0.0], dtype=np.float32))
    ids, dists = data.rerank(q, qh, None, 5)
    assert list(ids) == [50, 49, 51, 48, 52]
    assert list(dists) == [0.0, 1.0, 1.0, 2.0, 2.0]


def test_cosine_of_zeroish_item_is_zero():
    """reader.rs:81-99: a single item [0,0,0] under Cosine, queried with itself -> distance 0."""
    data = O.Data(O.COSINE, np.zeros((1, 3), dtype=np.float32))
    q, qh = data.query_leaf(np.array([0.00001, 0.00001, 0.00001], dtype=np.float32))
    ids, dists = data.rerank(q, qh, None, 1)
    assert list(ids) == [0] and list(dists) == [0.0]


def test_filtering_candidates():
    """reader.rs:194-227: candidates restrict the re-ranked set."""
    vecs = np.array([[float(i), 0.0] for i in range(100)], dtype=np.float32)
    data = O.Data(O.EUCLIDEAN, vecs)
    q, qh = data.query_leaf(np.array([0.0, 0.0], dtype=np.float32))
    ids, _ = data.rerank(q, qh, np.arange(0, 100, 2, dtype=np.uint32), 10)
    assert list(ids) == list(range(0, 20, 2))


# --- src/tests/reader.rs:283-299 median_top_k_vs_binary_heap -------------------------------------

def test_median_top_k_vs_spec_property():
    rng = np.random.default_rng(5)
    for _ in range(300):
        n = int(rng.integers(1, 1000))
        bits = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
        d = bits.view(np.float32).copy()
        d = d[np.isfinite(d)] if rng.random() < 0.5 else np.where(np.isfinite(d), d, 1.0).astype(np.float32)
        n = d.size
        if n == 0:
            continue
        k = int(rng.integers(1, n + 1))
        ids = np.arange(n, dtype=np.uint32)
        a = O.top_k(d, ids, k)
        b = O.top_k(d, ids, k, spec=True)
        assert np.array_equal(a[0], b[0]) and a[1].tobytes() == b[1].tobytes()


def test_top_k_ties_break_by_id_and_nan_sorts_last():
    d = np.array([1.0, np.nan, 1.0, 0.0, -0.0, np.inf], dtype=np.float32)
    ids = np.array([7, 1, 3, 9, 2, 4], dtype=np.uint32)
    oi, od = O.top_k(d, ids, 6)
    assert list(oi) == [2, 9, 3, 7, 4, 1]


# --- src/tests/writer.rs:266-293 -----------------------------------------------------------------

def test_split_normal_of_points_on_the_diagonal():
    """write_vectors_until_there_is_a_split: 3-d items [i,i,i]; the plane normal prints as
    [0.5774, 0.5774, 0.5774] whatever the sampled pair is (up to sign)."""
    vecs = np.array([[float(i)] * 3 for i in range(4)], dtype=np.float32)
    data = O.Data(O.EUCLIDEAN, vecs)
    nv, nh = data.create_split([0, 3] + [1, 2] * 5)
    n = nv.view(np.float32)
    assert ["%.4f" % abs(x) for x in n] == ["0.5774"] * 3
    # bias consistent with the plane through the midpoint of the two centroids
    p, ph, q, qh = data.two_means([0, 3] + [1, 2] * 5)
    mid = (p.astype(np.float64) + q.astype(np.float64)) / 2
    assert math.isclose(float(nh[0]), -float(np.dot(n.astype(np.float64), mid)), rel_tol=1e-5, abs_tol=1e-6)


def test_target_split_imbalance_formula():
    """src/writer.rs:1348-1353."""
    L = O.lib()
    assert L.ao_split_imbalance(50, 50) == pytest.approx(0.5)
    assert L.ao_split_imbalance(0, 0) == 1.0  # 0/eps = 0 -> max(0, 1)
    assert L.ao_split_imbalance(95, 5) == pytest.approx(0.95)
    assert L.ao_split_imbalance(96, 4) > 0.95


# --- Reader::nns_by_leaf restated (src/reader.rs:317-401), on oracle-built trees ---------------------------------

def test_oracle_search_exhaustive_equals_full_rerank_and_respects_search_k():
    rng = np.random.default_rng(0)
    for metric in (O.EUCLIDEAN, O.COSINE, O.BQ_COSINE):
        vecs = rng.standard_normal((600, 24)).astype(np.float32)
        data = O.Data(metric, vecs)
        forest = data.build_tree(10, 5).as_forest(data)
        qv, qh = data.query_leaf(rng.standard_normal(24).astype(np.float32))
        full, cand = O.search(data, forest, qv, qh, 7, search_k=2**62)
        ids, d = data.rerank(qv, qh, None, 7)
        assert [i for i, _ in full] == list(ids) and len(cand) == 600
        few, cand = O.search(data, forest, qv, qh, 7, search_k=1)   # one leaf: at most split_after candidates
        assert 0 < len(cand) <= 10 and len(few) == min(7, len(cand))
        assert [x for x, _ in few] == [int(i) for i in data.rerank(qv, qh, cand, 7)[0]]
        none, cand = O.search(data, forest, qv, qh, 7, search_k=50, candidates=[])
        assert none == [] and len(cand) == 0
        some, cand = O.search(data, forest, qv, qh, 7, search_k=2**62, candidates=range(0, 600, 5))
        assert set(cand) == set(range(0, 600, 5))


def test_oracle_search_pops_equal_keys_by_node_id():
    """tests/golden/search_equal_keys_bq.npz (found by scripts/fuzz_gpu.py): the two best leaves of the query carry the
    same key and search_k = 90 lies between their sizes (83 / 227 ids).  `BinaryHeap<(OrderedFloat<f32>, NodeId)>`
    (src/reader.rs:338-374) pops the bigger node id first -- the 227-id leaf -- and the search ends there; the oracle
    restates that, and the file's expected answer is what the GPU test of the same fixture must return."""
    import os
    import types
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "search_equal_keys_bq.npz"))
    forest = types.SimpleNamespace(n_trees=len(g["roots"]), roots=g["roots"], nodes=g["nodes"], normals=g["normals"],
                                   normal_stride=int(g["normal_stride"]), _vec_off=int(g["vec_off"]), _hdr_off=int(g["hdr_off"]),
                                   descendants=g["descendants"])
    data = O.Data(int(g["metric"]), g["vecs"], ids=g["ids"])
    qv, qh = data.query_leaf(g["query"])
    got, cand = O.search(data, forest, qv, qh, int(g["count"]), int(g["sk"]))
    assert [i for i, _ in got] == list(g["want_ids"]) and len(cand) == int(g["n_candidates"]) == 227
    assert np.array_equal(np.array([d for _, d in got], dtype=np.float32).view(np.uint32), g["want_dists"].view(np.uint32))
    # the leaf that is NOT taken has the same key: both leaves are children reached with margin 5.0
    leaves = {int(n["count"]) for n in g["nodes"] if n["kind"] == 1}
    assert {83, 227} <= leaves
