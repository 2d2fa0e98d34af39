"""The reference's tests pin numeric distance VALUES only for the Euclidean metric.  For the other
metrics the oracle is cross-checked here against an independent float64 evaluation of the formulas in
src/distance/*.rs (Appendix A.2 of SURVEY.md): relative error <= 1e-5 (the north star's f32 tolerance) for the f32
metrics, exact equality for the integer 1-bit metrics.  No GPU needed."""
import numpy as np
import pytest

from oracle import oracle as O

REL_TOL = 1e-5  # "within 1e-5 relative on f32 distances" (BASELINE.json north_star).  For sums with cancellation (dot
                # products, margins) "relative" is taken against the magnitude of the terms, sum |p_i q_i|: an f32
                # summation cannot be relatively accurate against a result that cancelled to ~0.


def data(n, dims, seed):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((n, dims)).astype(np.float32), rng.standard_normal(dims).astype(np.float32)


def close(got, want, scale=None):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    scale = np.maximum(np.abs(want), 1e-30) if scale is None else np.asarray(scale, np.float64)
    assert np.all(np.abs(got - want) <= REL_TOL * scale + 1e-7), float(np.max(np.abs(got - want) / scale))


@pytest.mark.parametrize("dims", [3, 17, 32, 100, 768, 1536])
def test_f32_metrics_against_float64(dims):
    vecs, q = data(300, dims, dims)
    v64, q64 = vecs.astype(np.float64), q.astype(np.float64)
    # Euclidean: squared distance, sqrt when returned to the user (euclidean.rs:45-47, mod.rs:59-61)
    od = O.Data(0, vecs)
    qv, qh = od.query_leaf(q)
    close(od.distances(qv, qh), ((v64 - q64) ** 2).sum(1))
    # Manhattan: sum |p - q| (manhattan.rs:44-46)
    od = O.Data(1, vecs)
    qv, qh = od.query_leaf(q)
    close(od.distances(qv, qh), np.abs(v64 - q64).sum(1))
    # Cosine: (1 - cos) / 2 (cosine.rs:43-59)
    od = O.Data(2, vecs)
    qv, qh = od.query_leaf(q)
    cos = (v64 @ q64) / (np.linalg.norm(v64, axis=1) * np.linalg.norm(q64))
    close(od.distances(qv, qh), (1.0 - np.clip(cos, -1.0, 1.0)) / 2.0, scale=np.ones(len(vecs)))
    # DotProduct: -dot (dot_product.rs:52-56); normalized_distance = -d = the dot product (:81-83)
    od = O.Data(3, vecs)
    od.preprocess_dot()
    qv, qh = od.query_leaf(q)
    mag = np.abs(v64 * q64).sum(1)
    close(od.distances(qv, qh), -(v64 @ q64), scale=mag)
    ids, dists = od.rerank(qv, qh, None, 10)
    want = np.sort(v64 @ q64)[::-1][:10]
    close(dists, want, scale=np.full(10, mag.max()))
    # its preprocess (dot_product.rs:119-165): extra_dim = sqrt(max_norm^2 - |v|^2), norm = max_norm^2
    norms = np.linalg.norm(v64, axis=1)
    close(od.headers[:, 0] ** 2, np.maximum(norms.max() ** 2 - norms ** 2, 0.0), scale=np.full(len(vecs), norms.max() ** 2))
    close(od.headers[:, 1], np.full(len(vecs), norms.max() ** 2))


@pytest.mark.parametrize("dims", [5, 64, 65, 768, 1000])
def test_one_bit_metrics_are_exact_integers(dims):
    vecs, q = data(200, dims, 7 * dims)
    vb, qb = vecs >= 0, q >= 0            # bit = is_sign_positive (binary_quantized.rs:84-87)
    words = (dims + 63) // 64
    hamming = (vb != qb).sum(1).astype(np.int64)  # padding bits are 0 in both operands
    bqdot = 64 * words - 2 * hamming             # simple.rs:119-131 (padding counts as +1 per bit)
    for metric, want in ((4, 4.0 * hamming / dims),                      # bq_euclidean.rs:117-124, :56-58
                         (5, np.maximum(2.0 * hamming, 0.0) / dims)):   # bq_manhattan.rs:113-120, :55-57
        od = O.Data(metric, vecs)
        qv, qh = od.query_leaf(q)
        raw = od.distances(qv, qh)
        assert np.array_equal(raw, (want * dims).astype(np.float32))  # built distances: exact integers
        ids, dists = od.rerank(qv, qh, None, 200)
        order = np.lexsort((np.arange(200), raw))
        assert list(ids) == list(order)
        assert np.allclose(dists, want[order].astype(np.float32), rtol=1e-6)
    od = O.Data(6, vecs)  # bq_cosine.rs:49-64: norms are sqrt(64 * words) for every vector
    qv, qh = od.query_leaf(q)
    close(od.distances(qv, qh), (1.0 - bqdot / (64.0 * words)) / 2.0)


def test_margins_and_sides_against_float64():
    # margin = bias + dot (Euclidean / Manhattan), dot (Cosine), dot + e_n * e_q (DotProduct); side = sign bit
    vecs, n = data(400, 96, 5)
    v64, n64 = vecs.astype(np.float64), n.astype(np.float64)
    for metric in (0, 1, 2):
        od = O.Data(metric, vecs)
        nh = np.array([0.25 if metric < 2 else 0.0, 0.0], np.float32)
        sides, n_left, margins = od.split_sides(n.view(np.uint8), nh)
        want = v64 @ n64 + float(nh[0])
        ok = np.abs(want) > 1e-3  # away from the plane the sign is not a rounding question
        close(margins[ok], want[ok], scale=np.abs(v64 * n64).sum(1)[ok] + abs(float(nh[0])))
        assert np.array_equal(sides[ok], (want[ok] >= 0).astype(np.uint8))
        assert n_left == int((sides == 0).sum())
