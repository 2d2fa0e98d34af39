"""ctypes binding of the CPU oracle (oracle/libarroy_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
``cpu_baseline`` leg.  The product package ``arroy_amd`` never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libarroy_oracle.so")

EUCLIDEAN, MANHATTAN, COSINE, DOT_PRODUCT, BQ_EUCLIDEAN, BQ_MANHATTAN, BQ_COSINE = range(7)
METRIC_NAMES = {
    EUCLIDEAN: "euclidean", MANHATTAN: "manhattan", COSINE: "cosine", DOT_PRODUCT: "dot-product",
    BQ_EUCLIDEAN: "binary quantized euclidean", BQ_MANHATTAN: "binary quantized manhattan",
    BQ_COSINE: "binary quantized cosine",
}


def build(force: bool = False) -> str:
    """Compile the oracle with the committed recipe (oracle/Makefile)."""
    src = os.path.join(HERE, "arroy_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    return LIB_PATH


class AhNode(C.Structure):
    _fields_ = [("kind", C.c_uint8), ("has_normal", C.c_uint8), ("reserved", C.c_uint16), ("tree", C.c_uint32),
                ("left", C.c_uint32), ("right", C.c_uint32), ("offset", C.c_uint64), ("count", C.c_uint32),
                ("depth", C.c_uint32)]


class AhForestView(C.Structure):
    _fields_ = [("n_trees", C.c_uint32), ("n_nodes", C.c_uint64), ("roots", C.POINTER(C.c_uint32)),
                ("nodes", C.POINTER(AhNode)), ("normals", C.POINTER(C.c_uint8)), ("normals_len", C.c_uint64),
                ("normal_stride", C.c_uint64), ("normal_vector_offset", C.c_uint64), ("normal_header_offset", C.c_uint64),
                ("descendants", C.POINTER(C.c_uint32)),
                ("descendants_len", C.c_uint64)]


class AoRefNode(C.Structure):
    _fields_ = [("id", C.c_uint32), ("kind", C.c_uint8), ("has_normal", C.c_uint8), ("left", C.c_uint32),
                ("right", C.c_uint32), ("offset", C.c_uint64), ("count", C.c_uint32)]


class AoData(C.Structure):
    _fields_ = [("metric", C.c_int), ("dims", C.c_uint32), ("n", C.c_uint64), ("vectors", C.c_void_p),
                ("headers", C.c_void_p), ("ids", C.c_void_p)]


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        f32p, u32p, u8p = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)
        for name in ("ao_dot", "ao_euclid", "ao_dot_scalar", "ao_euclid_scalar", "ao_dot_sse", "ao_euclid_sse",
                     "ao_dot_avx_emul", "ao_euclid_avx_emul", "ao_dot_avx_real", "ao_euclid_avx_real"):
            fn = getattr(L, name)
            fn.restype = C.c_float
            fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.ao_set_tier.argtypes = [C.c_int]
        L.ao_set_use_intrinsics.argtypes = [C.c_int]
        L.ao_bq_bytes.restype = C.c_size_t
        L.ao_bq_bytes.argtypes = [C.c_size_t]
        L.ao_bq_quantize.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.ao_bq_dequantize.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.ao_bq_dot_i32.restype = C.c_int32
        L.ao_bq_dot_i32.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.ao_bq_hamming.restype = C.c_uint32
        L.ao_bq_hamming.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.ao_header_floats.restype = C.c_size_t
        L.ao_header_floats.argtypes = [C.c_int]
        L.ao_vector_bytes.restype = C.c_size_t
        L.ao_vector_bytes.argtypes = [C.c_int, C.c_size_t]
        L.ao_norm_no_header.restype = C.c_float
        L.ao_norm_no_header.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
        L.ao_new_header.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        L.ao_norm.restype = C.c_float
        L.ao_norm.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
        for name in ("ao_built_distance", "ao_non_built_distance", "ao_margin"):
            fn = getattr(L, name)
            fn.restype = C.c_float
            fn.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.ao_normalized_distance.restype = C.c_float
        L.ao_normalized_distance.argtypes = [C.c_int, C.c_float, C.c_size_t]
        L.ao_side_of_margin.restype = C.c_int
        L.ao_side_of_margin.argtypes = [C.c_float]
        L.ao_pq_distance.restype = C.c_float
        L.ao_pq_distance.argtypes = [C.c_float, C.c_float, C.c_int]
        L.ao_new_headers.argtypes = [C.POINTER(AoData)]
        L.ao_distances.argtypes = [C.POINTER(AoData), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        for name in ("ao_top_k", "ao_top_k_spec"):
            fn = getattr(L, name)
            fn.restype = C.c_size_t
            fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
        L.ao_rerank.restype = C.c_size_t
        L.ao_rerank.argtypes = [C.POINTER(AoData), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_size_t,
                                C.c_void_p, C.c_void_p]
        L.ao_split_sides.argtypes = [C.POINTER(AoData), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                     C.POINTER(C.c_uint64), C.c_void_p]
        L.ao_preprocess_dot.argtypes = [C.POINTER(AoData), C.POINTER(C.c_float)]
        L.ao_two_means.restype = C.c_size_t
        L.ao_two_means.argtypes = [C.POINTER(AoData), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ao_create_split.argtypes = [C.POINTER(AoData), C.c_void_p, C.c_void_p, C.c_void_p]
        L.ao_split_imbalance.restype = C.c_double
        L.ao_split_imbalance.argtypes = [C.c_uint64, C.c_uint64]
        L.ao_build_tree.restype = C.c_void_p
        L.ao_build_tree.argtypes = [C.POINTER(AoData), C.c_uint32, C.c_uint64]
        L.ao_build_tree_on.restype = C.c_void_p
        L.ao_build_tree_on.argtypes = [C.POINTER(AoData), C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint64]
        L.ao_tree_view.argtypes = [C.c_void_p, C.POINTER(AhForestView), C.POINTER(C.c_uint32)]
        L.ao_tree_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                       C.POINTER(C.c_uint64)]
        L.ao_tree_free.argtypes = [C.c_void_p]
        L.ao_build_forest_count.restype = C.c_uint64
        L.ao_build_forest_count.argtypes = [C.POINTER(AoData), C.c_uint32, C.c_void_p, C.c_uint32]
        L.ao_synth_fill.argtypes = [C.c_uint64, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p]
        L.ao_synth_value.restype = C.c_float
        L.ao_synth_value.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int]
        L.ao_build_tree_reference_order.restype = C.c_void_p
        L.ao_build_tree_reference_order.argtypes = [C.POINTER(AoData), C.c_uint32, C.c_void_p]
        L.ao_build_forest_reference_order.restype = C.c_void_p
        L.ao_build_forest_reference_order.argtypes = [C.POINTER(AoData), C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64]
        L.ao_ref_tree_nodes.restype = C.c_size_t
        L.ao_ref_tree_nodes.argtypes = [C.c_void_p, C.POINTER(C.POINTER(AoRefNode)), C.POINTER(C.POINTER(C.c_uint8)),
                                        C.POINTER(C.POINTER(C.c_uint32))]
        L.ao_ref_tree_free.argtypes = [C.c_void_p]
        L.ao_rng_from_seed.argtypes = [C.c_void_p, C.c_void_p]
        L.ao_rng_next_u32.restype = C.c_uint32
        L.ao_rng_next_u32.argtypes = [C.c_void_p]
        L.ao_rng_gen_seed.argtypes = [C.c_void_p, C.c_void_p]
        L.ao_rng_gen_bool.restype = C.c_int
        L.ao_rng_gen_bool.argtypes = [C.c_void_p]
        L.ao_rng_gen_range_inclusive_u32.restype = C.c_uint32
        L.ao_rng_gen_range_inclusive_u32.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.ao_rng_index_sample2.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.ao_search.restype = C.c_size_t
        L.ao_search.argtypes = [C.POINTER(AoData), C.POINTER(AhForestView), C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t,
                                C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                C.POINTER(C.c_size_t)]
        L.ao_route_items.argtypes = [C.POINTER(AoData), C.POINTER(AhForestView), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.ao_num_threads.restype = C.c_int
        L.ao_set_num_threads.argtypes = [C.c_int]
        _lib = L
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def is_bq(metric: int) -> bool:
    return metric >= BQ_EUCLIDEAN


def header_floats(metric: int) -> int:
    return 2 if metric == DOT_PRODUCT else 1


def vector_bytes(metric: int, dims: int) -> int:
    return ((dims + 63) // 64) * 8 if is_bq(metric) else 4 * dims


# ---- spaces -------------------------------------------------------------------------------

def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def dot(u, v, tier="auto"):
    u, v = _f32(u), _f32(v)
    name = {"auto": "ao_dot", "scalar": "ao_dot_scalar", "sse": "ao_dot_sse", "avx_emul": "ao_dot_avx_emul",
            "avx_real": "ao_dot_avx_real"}[tier]
    return np.float32(getattr(lib(), name)(_p(u), _p(v), u.size))


def euclid(u, v, tier="auto"):
    u, v = _f32(u), _f32(v)
    name = {"auto": "ao_euclid", "scalar": "ao_euclid_scalar", "sse": "ao_euclid_sse",
            "avx_emul": "ao_euclid_avx_emul", "avx_real": "ao_euclid_avx_real"}[tier]
    return np.float32(getattr(lib(), name)(_p(u), _p(v), u.size))


def bq_quantize(x) -> np.ndarray:
    x = _f32(x)
    out = np.zeros(lib().ao_bq_bytes(x.size), dtype=np.uint8)
    lib().ao_bq_quantize(_p(x), x.size, _p(out))
    return out


def bq_dequantize(b) -> np.ndarray:
    b = np.ascontiguousarray(b, dtype=np.uint8)
    out = np.zeros(b.size * 8, dtype=np.float32)
    lib().ao_bq_dequantize(_p(b), b.size, _p(out))
    return out


# ---- a data set in the oracle's layout ------------------------------------------------------

class Data:
    """Items in the stored layout: codec bytes + headers, ascending ids (id == row unless ids given)."""

    def __init__(self, metric: int, vectors_f32: np.ndarray, ids=None, headers=None, codec_bytes=None):
        L = lib()
        self.metric = metric
        v = _f32(vectors_f32)
        self.n, self.dims = v.shape if v.ndim == 2 else (0, 0)
        if codec_bytes is not None:
            self.codec = np.ascontiguousarray(codec_bytes, dtype=np.uint8)
        elif is_bq(metric):
            self.codec = np.stack([bq_quantize(r) for r in v]) if self.n else np.zeros((0, 0), np.uint8)
        else:
            self.codec = v.view(np.uint8).reshape(self.n, 4 * self.dims)  # no copy: rows may be GBs
        hf = header_floats(metric)
        self.headers = np.zeros((self.n, hf), dtype=np.float32) if headers is None else \
            _f32(headers).reshape(self.n, hf).copy()
        self.ids = None if ids is None else np.ascontiguousarray(ids, dtype=np.uint32)
        self._keep = v
        self._c = AoData(metric, self.dims, self.n, self.codec.ctypes.data, self.headers.ctypes.data,
                         None if self.ids is None else self.ids.ctypes.data)
        if headers is None and self.n:
            L.ao_new_headers(C.byref(self._c))  # D::new_header at add_item time

    def c(self):
        return C.byref(self._c)

    def id_of(self, row):
        return int(row) if self.ids is None else int(self.ids[row])

    # Reader::by_vector query leaf: codec + new_header (src/reader.rs:64-75)
    def query_leaf(self, vector):
        vector = _f32(vector)
        q = bq_quantize(vector) if is_bq(self.metric) else vector.view(np.uint8).copy()
        h = np.zeros(2, dtype=np.float32)
        lib().ao_new_header(self.metric, _p(q), self.dims, _p(h))
        return q, h

    def item_leaf(self, row):
        h = np.zeros(2, dtype=np.float32)
        h[: self.headers.shape[1]] = self.headers[row]
        return self.codec[row].copy(), h

    def distances(self, q, qh, rows=None):
        n = self.n if rows is None else len(rows)
        out = np.zeros(n, dtype=np.float32)
        r = None if rows is None else np.ascontiguousarray(rows, dtype=np.uint32)
        lib().ao_distances(self.c(), _p(q), _p(qh), None if r is None else _p(r), n, _p(out))
        return out

    def rerank(self, q, qh, rows, k):
        n = self.n if rows is None else len(rows)
        r = None if rows is None else np.ascontiguousarray(rows, dtype=np.uint32)
        oi = np.zeros(max(k, 1), dtype=np.uint32)
        od = np.zeros(max(k, 1), dtype=np.float32)
        m = lib().ao_rerank(self.c(), _p(q), _p(qh), None if r is None else _p(r), n, k, _p(oi), _p(od))
        return oi[:m].copy(), od[:m].copy()

    def split_sides(self, nv, nh, rows=None):
        n = self.n if rows is None else len(rows)
        r = None if rows is None else np.ascontiguousarray(rows, dtype=np.uint32)
        sides = np.zeros(n, dtype=np.uint8)
        margins = np.zeros(n, dtype=np.float32)
        nl = C.c_uint64(0)
        nh2 = np.zeros(2, dtype=np.float32)
        nh2[: len(nh)] = nh
        lib().ao_split_sides(self.c(), _p(np.ascontiguousarray(nv)), _p(nh2), None if r is None else _p(r), n,
                             _p(sides), C.byref(nl), _p(margins))
        return sides, int(nl.value), margins

    def create_split(self, sample_rows):
        s = np.ascontiguousarray(sample_rows, dtype=np.uint32)
        assert s.size == 12
        nv = np.zeros(vector_bytes(self.metric, self.dims), dtype=np.uint8)
        nh = np.zeros(2, dtype=np.float32)
        lib().ao_create_split(self.c(), _p(s), _p(nv), _p(nh))
        return nv, nh[: header_floats(self.metric)].copy()

    def two_means(self, sample_rows):
        s = np.ascontiguousarray(sample_rows, dtype=np.uint32)
        fd = vector_bytes(self.metric, self.dims) * 8 if is_bq(self.metric) else self.dims
        p = np.zeros(fd, np.float32); q = np.zeros(fd, np.float32)
        ph = np.zeros(2, np.float32); qh = np.zeros(2, np.float32)
        lib().ao_two_means(self.c(), _p(s), _p(p), _p(ph), _p(q), _p(qh))
        return p, ph, q, qh

    def preprocess_dot(self):
        m = C.c_float(0)
        lib().ao_preprocess_dot(self.c(), C.byref(m))
        return np.float32(m.value)

    def build_tree(self, split_after: int, seed: int, rows=None):
        return Tree(self, split_after, seed, rows)


def top_k(dists, ids, k, spec=False):
    d = _f32(dists)
    i = np.ascontiguousarray(ids, dtype=np.uint32)
    oi = np.zeros(max(k, 1), dtype=np.uint32)
    od = np.zeros(max(k, 1), dtype=np.float32)
    fn = lib().ao_top_k_spec if spec else lib().ao_top_k
    m = fn(_p(d), _p(i), d.size, k, _p(oi), _p(od))
    return oi[:m].copy(), od[:m].copy()


class Tree:
    """One tree built by the oracle's depth-first restatement of make_tree_in_file."""

    def __init__(self, data: Data, split_after: int, seed: int, rows=None):
        L = lib()
        if rows is None:
            h = L.ao_build_tree(data.c(), split_after, seed)
        else:
            r = np.ascontiguousarray(rows, dtype=np.uint32)
            h = L.ao_build_tree_on(data.c(), split_after, seed, _p(r), r.size)
        view = AhForestView()
        root = C.c_uint32(0)
        L.ao_tree_view(h, C.byref(view), C.byref(root))
        n = view.n_nodes
        self.root = int(root.value)
        self.nodes = [(nd.kind, nd.has_normal, nd.left, nd.right, nd.offset, nd.count, nd.depth)
                      for nd in (view.nodes[i] for i in range(n))]
        self.normals = bytes(C.string_at(view.normals, view.normals_len)) if view.normals_len else b""
        self.descendants = np.ctypeslib.as_array(view.descendants, shape=(view.descendants_len,)).copy() \
            if view.descendants_len else np.zeros(0, np.uint32)
        me, rt, dm = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        L.ao_tree_counters(h, C.byref(me), C.byref(rt), C.byref(dm))
        self.margin_evals, self.retries, self.dummy_normals = int(me.value), int(rt.value), int(dm.value)
        self.stride = 4 * header_floats(data.metric) + vector_bytes(data.metric, data.dims)
        L.ao_tree_free(h)

    def as_forest(self, data: "Data"):
        """The tree as a one-tree forest object with the attributes `search()` / `forest_view()` expect."""
        node_dt = np.dtype([("kind", "u1"), ("has_normal", "u1"), ("reserved", "<u2"), ("tree", "<u4"), ("left", "<u4"), ("right", "<u4"),
                            ("offset", "<u8"), ("count", "<u4"), ("depth", "<u4")], align=True)
        f = type("OracleForest", (), {})()
        f.n_trees = 1
        f.roots = np.array([self.root], dtype=np.uint32)
        f.nodes = np.array([(k, hn, 0, 0, l, r, off, cnt, dep) for (k, hn, l, r, off, cnt, dep) in self.nodes], dtype=node_dt)
        f.normals = np.frombuffer(self.normals, dtype=np.uint8).copy() if self.normals else np.zeros(0, np.uint8)
        f.normal_stride = self.stride
        f._hdr_off, f._vec_off = 0, 4 * header_floats(data.metric)
        f.descendants = np.ascontiguousarray(self.descendants, dtype=np.uint32)
        return f

    def canonical(self, node=None):
        """Structure independent of node numbering: nested tuples."""
        import sys
        sys.setrecursionlimit(100000)
        node = self.root if node is None else node
        kind, has_normal, left, right, offset, count, depth = self.nodes[node]
        if kind == 1:
            return ("D", tuple(int(x) for x in self.descendants[offset:offset + count]))
        nb = self.normals[offset:offset + self.stride] if has_normal else None
        return ("S", nb, self.canonical(left), self.canonical(right))


# enum ah_synth_distribution (include/arroy_hip_policy.h)
SYNTH_UNIFORM_01, SYNTH_UNIFORM_PM1, SYNTH_NORMAL, SYNTH_NORMAL_OUTLIERS, SYNTH_CLUSTERED, SYNTH_LOW_RANK = range(6)
SYNTH_NAMES = ("uniform01", "uniform_pm1", "normal", "normal_outliers", "clustered", "low_rank")


def tree_hash(forest, tree: int, header_size: int, vector_size: int) -> str:
    """Numbering-independent content hash of one tree — equal iff `canonical()` is equal, without the nested tuples (a tree
    over 10M items has 10M ids and ~26 000 nodes).  `forest`: an arroy_amd.Forest or `Tree.as_forest(data)` (the same
    attributes).  Leaf: H("D" | ids); split: H("S" | normal record [header][vector] or "N" | left | right), bottom-up
    with an explicit stack."""
    import hashlib
    nodes, desc, normals = forest.nodes, forest.descendants, forest.normals
    stride, ho, vo = int(forest.normal_stride), int(forest._hdr_off), int(forest._vec_off)
    done = {}
    stack = [int(forest.roots[tree])]
    while stack:
        i = stack[-1]
        nd = nodes[i]
        if nd["kind"] == 1:
            off, cnt = int(nd["offset"]), int(nd["count"])
            done[i] = hashlib.blake2b(b"D" + np.ascontiguousarray(desc[off:off + cnt], dtype="<u4").tobytes(), digest_size=16).digest()
            stack.pop()
            continue
        left, right = int(nd["left"]), int(nd["right"])
        if left in done and right in done:
            if nd["has_normal"]:
                rec = normals[int(nd["offset"]): int(nd["offset"]) + stride]
                nb = b"V" + rec[ho:ho + header_size].tobytes() + rec[vo:vo + vector_size].tobytes()
            else:
                nb = b"N"
            done[i] = hashlib.blake2b(b"S" + nb + done.pop(left) + done.pop(right), digest_size=16).digest()
            stack.pop()
        else:
            stack += [x for x in (left, right) if x not in done]
    return done[int(forest.roots[tree])].hex()


def synth(seed: int, distribution: int, n: int, dims: int, first_item: int = 0) -> np.ndarray:
    out = np.zeros((n, dims), dtype=np.float32)
    lib().ao_synth_fill(seed, distribution, first_item, n, dims, _p(out))
    return out


class ChaCha12:
    """`StdRng` of rand 0.8 (restated in arroy_oracle.c): only what the tests need."""

    def __init__(self, seed: bytes):
        self._st = C.create_string_buffer(8 * 4 + 8 + 64 * 4 + 8)
        lib().ao_rng_from_seed(self._st, C.c_char_p(bytes(seed)))

    def next_u32(self) -> int:
        return int(lib().ao_rng_next_u32(self._st))

    def gen_f32(self) -> np.float32:
        """`rng.gen::<f32>()`: 24 random bits scaled into [0, 1)."""
        return np.float32((self.next_u32() >> 8) * (1.0 / (1 << 24)))

    def gen_seed(self) -> bytes:
        """`rng.gen::<[u8; 32]>()` (the argument of `StdRng::from_seed(rng.gen())`, src/writer.rs:575,795)."""
        out = C.create_string_buffer(32)
        lib().ao_rng_gen_seed(self._st, out)
        return out.raw

    def gen_bool(self) -> bool:
        return bool(lib().ao_rng_gen_bool(self._st))

    def gen_range_inclusive(self, low: int, high: int) -> int:
        return int(lib().ao_rng_gen_range_inclusive_u32(self._st, low, high))

    def index_sample2(self, length: int):
        out = (C.c_uint32 * 2)()
        lib().ao_rng_index_sample2(self._st, length, out)
        return int(out[0]), int(out[1])


def build_tree_reference_order(data: "Data", split_after: int, seed: bytes = bytes([42] * 32), n_trees: int = 1,
                               skip_u32: int = 0):
    """`Writer::build` of a fresh index with the reference's own randomness (StdRng seeded like the reference's
    tests), depth-first, one rayon thread.  Returns {node_id: node} with node = ("D", [ids]) or
    ("S", left_id, right_id, header f32[], vector bytes or None)."""
    L = lib()
    h = L.ao_build_forest_reference_order(data.c(), split_after, n_trees, C.c_char_p(bytes(seed)), skip_u32)
    nodes_p, normals_p, desc_p = C.POINTER(AoRefNode)(), C.POINTER(C.c_uint8)(), C.POINTER(C.c_uint32)()
    n = L.ao_ref_tree_nodes(h, C.byref(nodes_p), C.byref(normals_p), C.byref(desc_p))
    hs, vs = 4 * header_floats(data.metric), vector_bytes(data.metric, data.dims)
    out = {}
    for i in range(n):
        nd = nodes_p[i]
        if nd.kind == 1:
            out[int(nd.id)] = ("D", [int(desc_p[nd.offset + j]) for j in range(nd.count)])
        else:
            raw = bytes(C.string_at(C.addressof(normals_p.contents) + nd.offset, hs + vs))
            hdr = np.frombuffer(raw[:hs], dtype=np.float32).copy()
            out[int(nd.id)] = ("S", int(nd.left), int(nd.right), hdr, raw[hs:] if nd.has_normal else None)
    L.ao_ref_tree_free(h)
    return out


def forest_view(forest) -> "AhForestView":
    """An ah_forest_view over the numpy arrays of an arroy_amd.Forest (the arrays must stay alive)."""
    v = AhForestView()
    v.n_trees = forest.n_trees
    v.n_nodes = len(forest.nodes)
    v.roots = forest.roots.ctypes.data_as(C.POINTER(C.c_uint32))
    v.nodes = C.cast(forest.nodes.ctypes.data, C.POINTER(AhNode))
    v.normals = forest.normals.ctypes.data_as(C.POINTER(C.c_uint8)) if forest.normals.size else None
    v.normals_len = forest.normals.size
    v.normal_stride = forest.normal_stride
    v.normal_vector_offset = forest._vec_off
    v.normal_header_offset = forest._hdr_off
    v.descendants = forest.descendants.ctypes.data_as(C.POINTER(C.c_uint32)) if forest.descendants.size else None
    v.descendants_len = forest.descendants.size
    return v


def search(data: "Data", forest, qv, qh, count: int, search_k: int = 0, oversampling: int = 0, candidates=None,
           candidates_sorted: bool = False, want_candidates: bool = True, view=None):
    """`Reader::nns_by_leaf` (src/reader.rs:317-401) over `forest`; returns ([(id, dist)...], candidate ids).
    candidates_sorted: `candidates` is already an ascending uint32 array of distinct ids; want_candidates=False skips the
    copy of the candidate list (its buffer is sized by the forest's Descendants blob: gigabytes at 10M x 100 trees);
    view: a forest_view(forest) the caller already holds."""
    view = forest_view(forest) if view is None else view
    if candidates is None:
        filt = None
    elif candidates_sorted:
        filt = np.ascontiguousarray(candidates, dtype=np.uint32)
    else:
        filt = np.ascontiguousarray(sorted(set(int(c) for c in candidates)), dtype=np.uint32)
    oi = np.zeros(max(count, 1), dtype=np.uint32)
    od = np.zeros(max(count, 1), dtype=np.float32)
    cap = int(forest.descendants.size) + 1 if want_candidates else 0
    cand = np.zeros(max(cap, 1), dtype=np.uint32)
    nc = C.c_size_t(0)
    qh2 = np.zeros(2, dtype=np.float32)
    qh2[: len(qh)] = qh
    m = lib().ao_search(data.c(), C.byref(view), _p(np.ascontiguousarray(qv)), _p(qh2), count, min(search_k, 2**62),
                        oversampling, None if filt is None else _p(filt), 0 if filt is None else filt.size,
                        0 if filt is None else 1, _p(oi), _p(od), _p(cand) if want_candidates else None, cap, C.byref(nc))
    return [(int(oi[i]), float(od[i])) for i in range(m)], cand[: nc.value if want_candidates else 0].copy()


def route_items(data: "Data", forest, rows, tree_seeds) -> np.ndarray:
    """Incremental routing (src/writer.rs:1398-1459): [n_trees, n] node index reached by every item."""
    view = forest_view(forest)
    r = np.ascontiguousarray(rows, dtype=np.uint32)
    seeds = np.ascontiguousarray(tree_seeds, dtype=np.uint64)
    out = np.zeros((forest.n_trees, r.size), dtype=np.uint32)
    lib().ao_route_items(data.c(), C.byref(view), _p(r), r.size, _p(seeds), _p(out))
    return out
