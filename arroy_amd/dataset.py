"""`Dataset` / `Forest`: numpy-facing wrappers of the C ABI handles (ah_dataset / ah_forest).

`Dataset` is the HBM-resident image of what arroy calls `ImmutableLeafs` (src/parallel.rs:262-312);
every method is one C-ABI call, all arithmetic happens in HIP kernels.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib
from .distances import BY_METRIC, Distance


def _u32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint32)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Dataset:
    def __init__(self, distance: type[Distance], dimensions: int, capacity: int, device: int = 0, _handle=None,
                 _finalized: bool = False):
        self.distance = distance
        self.metric = distance.metric
        self.dimensions = int(dimensions)
        self.device = int(device)
        self._h = C.c_void_p()
        if _handle is not None:
            self._h = _handle
        else:
            _lib.check(_lib.lib().ah_dataset_create(self.metric, self.dimensions, int(capacity), device, C.byref(self._h)))
        self.finalized = _finalized

    def replicate(self, device: int) -> "Dataset":
        """A replica on another GPU of the node, copied device to device (ah_dataset_replicate): the multi-GPU build
        shards trees over replicas of the read-only dataset."""
        h = C.c_void_p()
        _lib.check(_lib.lib().ah_dataset_replicate(self._h, int(device), C.byref(h)))
        return Dataset(self.distance, self.dimensions, 0, device=device, _handle=h, _finalized=self.finalized)

    # -- staging ---------------------------------------------------------------------------------
    def upload_vectors(self, item_ids: Sequence[int], vectors) -> None:
        """`Writer::add_item` for a batch (src/writer.rs:380-394)."""
        ids = _u32(item_ids)
        v = _f32(vectors)
        if v.ndim != 2 or v.shape[1] != self.dimensions:
            got = v.shape[1] if v.ndim == 2 else v.size
            raise _lib.InvalidVecDimension(1, f"invalid vector dimensions, provided {got} but expected "
                                              f"{self.dimensions}")  # src/error.rs:17-23
        if v.shape[0] != ids.size:
            raise ValueError("ids and vectors disagree on the number of items")
        _lib.check(_lib.lib().ah_dataset_upload_vectors(self._h, _ptr(ids), _ptr(v), ids.size))

    def upload_records(self, item_ids: Sequence[int], records: Sequence[bytes], preprocessed: Optional[bool] = None) -> None:
        """Stored item records `[0u8][header][vector]` as they sit in LMDB pages (src/node.rs:224-228).
        `preprocessed` (DotProduct only): True = the headers come from a built database, i.e. `DotProduct::preprocess`
        already ran over them (ah_dataset_set_preprocessed); False = freshly added items ({0, 0} headers); None (default)
        leaves the dataset's flag alone, so a build over records nobody vouched for fails with AH_ERR_NEED_PREPROCESS
        instead of silently using extra_dim = 0."""
        ids = _u32(item_ids)
        n = ids.size
        if self.metric == 3 and preprocessed is not None:
            _lib.check(_lib.lib().ah_dataset_set_preprocessed(self._h, 1 if preprocessed else 0))
        if n == 0:
            return
        rec_len = len(records[0])
        # deliberately misaligned copies: LMDB hands out pointers at page+16+... (SURVEY.md §7 hard part 4)
        keep = [C.create_string_buffer(b"\0" + bytes(r), rec_len + 1) for r in records]
        ptrs = (C.c_void_p * n)(*[C.addressof(b) + 1 for b in keep])
        _lib.check(_lib.lib().ah_dataset_upload_records(self._h, _ptr(ids), ptrs, rec_len, n))

    def reserve_build(self, n_trees: int, split_after: int = 0) -> None:
        """ah_dataset_reserve_build: while the records are still being staged, obtain the device memory the first build will
        ask for (fresh HBM can cost the driver tens of ms per GB) and park it in the library's device cache."""
        _lib.check(_lib.lib().ah_dataset_reserve_build(self._h, int(n_trees), int(split_after)))

    def fill_synthetic(self, seed: int, distribution: int, n_items: int) -> None:
        _lib.check(_lib.lib().ah_dataset_fill_synthetic(self._h, seed, distribution, n_items))

    def finalize(self) -> "Dataset":
        _lib.check(_lib.lib().ah_dataset_finalize(self._h))
        self.finalized = True
        return self

    def __len__(self) -> int:
        n = C.c_uint64(0)
        _lib.check(_lib.lib().ah_dataset_len(self._h, C.byref(n)))
        return int(n.value)

    def item_vector(self, item_id: int) -> np.ndarray:
        out = np.zeros(self.dimensions, dtype=np.float32)
        _lib.check(_lib.lib().ah_dataset_item_vector(self._h, item_id, _ptr(out)))
        return out

    def read_headers(self, first_row: int = 0, n: Optional[int] = None) -> np.ndarray:
        n = len(self) - first_row if n is None else n
        hf = self.distance.header_size() // 4
        out = np.zeros((n, hf), dtype=np.float32)
        _lib.check(_lib.lib().ah_dataset_read_headers(self._h, first_row, n, _ptr(out)))
        return out

    def preprocess_dot(self) -> np.float32:
        m = C.c_float(0)
        _lib.check(_lib.lib().ah_preprocess_dot(self._h, C.byref(m)))
        return np.float32(m.value)

    # -- search side -------------------------------------------------------------------------------
    def distances(self, query=None, item: Optional[int] = None, ids=None, n: Optional[int] = None) -> np.ndarray:
        ids_a = None if ids is None else _u32(ids)
        n = (len(self) if n is None else n) if ids_a is None else ids_a.size
        out = np.zeros(n, dtype=np.float32)
        if query is not None:
            q = self._check_query(query)
            _lib.check(_lib.lib().ah_distances_by_vector(self._h, _ptr(q), _ptr(ids_a), n, _ptr(out)))
        else:
            _lib.check(_lib.lib().ah_distances_by_item(self._h, int(item), _ptr(ids_a), n, _ptr(out)))
        return out

    def rerank(self, k: int, query=None, item: Optional[int] = None, sorted_ids=None):
        ids_a = None if sorted_ids is None else _u32(sorted_ids)
        n = len(self) if ids_a is None else ids_a.size
        kk = max(1, min(int(k), n)) if n else 1
        oi = np.zeros(kk, dtype=np.uint32)
        od = np.zeros(kk, dtype=np.float32)
        on = C.c_size_t(0)
        if query is not None:
            q = self._check_query(query)
            _lib.check(_lib.lib().ah_rerank_by_vector(self._h, _ptr(q), _ptr(ids_a), n, int(k), _ptr(oi), _ptr(od),
                                                      C.byref(on)))
        else:
            _lib.check(_lib.lib().ah_rerank_by_item(self._h, int(item), _ptr(ids_a), n, int(k), _ptr(oi), _ptr(od),
                                                    C.byref(on)))
        return oi[: on.value].copy(), od[: on.value].copy()

    def rerank_batch(self, queries, id_lists, k: int):
        """`id_lists`: one ascending id list per query, or the C ABI's own shape `(ids, offsets)` (concatenated
        u32 ids + nq+1 u64 offsets) when the caller already holds it."""
        q = _f32(queries)
        if q.ndim != 2 or q.shape[1] != self.dimensions:
            raise _lib.InvalidVecDimension(1, "invalid query dimensions")
        nq = q.shape[0]
        if isinstance(id_lists, tuple):
            ids, offsets = _u32(id_lists[0]), np.ascontiguousarray(id_lists[1], dtype=np.uint64)
            assert offsets.size == nq + 1 and int(offsets[-1]) == ids.size
        else:
            offsets = np.zeros(nq + 1, dtype=np.uint64)
            offsets[1:] = np.cumsum([len(l) for l in id_lists])
            ids = _u32(np.concatenate([_u32(l) for l in id_lists])) if nq else np.zeros(0, np.uint32)
        oi = np.zeros((nq, k), dtype=np.uint32)
        od = np.zeros((nq, k), dtype=np.float32)
        oc = np.zeros(nq, dtype=np.uint32)
        _lib.check(_lib.lib().ah_rerank_batch(self._h, _ptr(q), nq, _ptr(ids), _ptr(offsets), k, _ptr(oi), _ptr(od),
                                              _ptr(oc)))
        return oi, od, oc

    def rerank_stats(self, reset: bool = False) -> dict:
        """ah_dataset_rerank_stats: where ah_rerank_batch's wall time went (kept while the tunable AH_RERANK_TIMING is 1)."""
        st = _lib.AhRerankStats()
        _lib.check(_lib.lib().ah_dataset_rerank_stats(self._h, C.byref(st), 1 if reset else 0))
        return {f: getattr(st, f) for f, _ in _lib.AhRerankStats._fields_}

    # -- build side --------------------------------------------------------------------------------
    def split_sides(self, normal_vector: np.ndarray, normal_header, sorted_ids=None, want_margins: bool = True):
        """The margin loop (src/writer.rs:1201-1207). Returns (sides u8 per item, n_left, margins)."""
        ids_a = None if sorted_ids is None else _u32(sorted_ids)
        n = len(self) if ids_a is None else ids_a.size
        nv = np.ascontiguousarray(normal_vector).view(np.uint8)
        assert nv.size == self.distance.vector_size(self.dimensions)
        nh = np.zeros(2, dtype=np.float32)
        h = _f32(normal_header).ravel()
        nh[: h.size] = h
        bits = np.zeros((n + 7) // 8, dtype=np.uint8)
        margins = np.zeros(n, dtype=np.float32) if want_margins else None
        nl = C.c_uint64(0)
        _lib.check(_lib.lib().ah_split_sides(self._h, _ptr(nv), _ptr(nh), _ptr(ids_a), n, _ptr(bits), C.byref(nl),
                                             _ptr(margins)))
        sides = np.unpackbits(bits, bitorder="little")[:n]
        return sides, int(nl.value), margins

    def create_split(self, sample_ids: Sequence[int]):
        """`D::create_split` with host-supplied samples (choose_two + 10 x choose)."""
        s = _u32(sample_ids)
        assert s.size == _lib.AH_SPLIT_SAMPLES
        nv = np.zeros(self.distance.vector_size(self.dimensions), dtype=np.uint8)
        nh = np.zeros(2, dtype=np.float32)
        _lib.check(_lib.lib().ah_create_split(self._h, _ptr(s), _ptr(nv), _ptr(nh)))
        return nv, nh[: self.distance.header_size() // 4].copy()

    def upload_record_pointers(self, item_ids, addresses, record_len: int) -> None:
        """ah_dataset_upload_records with raw addresses (e.g. into an mmap of an LMDB data file: the real, arbitrarily
        misaligned pointers `ImmutableLeafs::new` collects, src/parallel.rs:271-293)."""
        ids = _u32(item_ids)
        ptrs = (C.c_void_p * ids.size)(*[int(a) for a in addresses])
        _lib.check(_lib.lib().ah_dataset_upload_records(self._h, _ptr(ids), ptrs, int(record_len), ids.size))

    def build_forest(self, tree_seeds: Sequence[int], split_after: int = 0, cancel=None, progress=None,
                     max_trees_in_flight: int = 0, margin_mode: int = 0, max_host_threads: int = 0) -> "Forest":
        seeds = np.ascontiguousarray(tree_seeds, dtype=np.uint64)
        opt = _lib.AhBuildOptions()
        opt.n_trees = seeds.size
        opt.split_after = int(split_after)
        opt.tree_seeds = seeds.ctypes.data_as(C.POINTER(C.c_uint64))
        cflag = C.c_int(0)
        opt.cancel = C.pointer(cflag)
        keep = None
        watcher_stop = None
        if progress is not None:
            def _cb(_user, level, nodes_done, items_routed):
                progress(level, nodes_done, items_routed)
            keep = _lib.PROGRESS_FN(_cb)
            opt.progress = keep
        if cancel is not None:
            # `cancel` is arroy's `Fn() -> bool` (src/writer.rs:100,117-123).  The library polls a flag while the level's
            # kernels run; a watcher thread evaluates the closure meanwhile (ctypes releases the GIL during the call).
            import threading
            if cancel():  # polled before the first level too (src/writer.rs:1178)
                cflag.value = 1
            watcher_stop = threading.Event()

            def _watch():
                while not watcher_stop.wait(0.0005):
                    if cancel():
                        cflag.value = 1
                        return
            threading.Thread(target=_watch, daemon=True).start()
        opt.max_trees_in_flight = int(max_trees_in_flight)
        opt.margin_mode = int(margin_mode)
        opt.max_host_threads = int(max_host_threads)
        h = C.c_void_p()
        try:
            _lib.check(_lib.lib().ah_build_forest(self._h, C.byref(opt), C.byref(h)))
        finally:
            if watcher_stop is not None:
                watcher_stop.set()
        return Forest(h, self.distance, self.dimensions)

    def build_forest_stream(self, tree_seeds: Sequence[int], sink=None, split_after: int = 0, margin_mode: int = 0,
                            max_trees_in_flight: int = 0, max_host_threads: int = 0):
        """ah_build_forest_stream: the node sink DURING the build (`TmpNodes::put`, src/parallel.rs:130-147).  `sink(batch)`
        receives every `_lib.AhNodeBatch` (valid only during the call; return non-zero to stop the build); with sink=None the
        batches are collected into a `StreamedForest`.  Returns (roots, stats, collected or None)."""
        seeds = np.ascontiguousarray(tree_seeds, dtype=np.uint64)
        opt = _lib.AhBuildOptions()
        opt.n_trees = seeds.size
        opt.split_after = int(split_after)
        opt.tree_seeds = seeds.ctypes.data_as(C.POINTER(C.c_uint64))
        opt.margin_mode = int(margin_mode)
        opt.max_trees_in_flight = int(max_trees_in_flight)
        opt.max_host_threads = int(max_host_threads)
        collected = StreamedForest(self.distance, self.dimensions) if sink is None else None
        fn = collected.take if sink is None else sink
        failure = []

        def _cb(_user, batch_p):
            try:
                return int(fn(batch_p.contents) or 0)
            except BaseException as e:  # noqa: BLE001 — nothing may unwind through the C frames
                failure.append(e)
                return -1
        cb = _lib.NODE_BATCH_FN(_cb)
        roots = np.zeros(seeds.size, dtype=np.uint32)
        st = _lib.AhBuildStats()
        code = _lib.lib().ah_build_forest_stream(self._h, C.byref(opt), cb, None, _ptr(roots), C.byref(st))
        if failure:
            raise failure[0]
        _lib.check(code)
        stats = {f: getattr(st, f) for f, _ in _lib.AhBuildStats._fields_}
        stats["margin_mode_launches"] = list(st.margin_mode_launches)
        if collected is not None:
            collected.roots = roots
        return roots, stats, collected

    def build_subtrees(self, id_lists: Sequence[Sequence[int]], tree_seeds: Sequence[int], split_after: int = 0) -> "Forest":
        """`incremental_index_large_descendant` for many item subsets at once (ah_build_subtrees)."""
        seeds = np.ascontiguousarray(tree_seeds, dtype=np.uint64)
        assert seeds.size == len(id_lists)
        offsets = np.zeros(len(id_lists) + 1, dtype=np.uint64)
        offsets[1:] = np.cumsum([len(l) for l in id_lists])
        ids = _u32(np.concatenate([_u32(l) for l in id_lists])) if len(id_lists) else np.zeros(1, np.uint32)
        opt = _lib.AhBuildOptions()
        opt.n_trees = seeds.size
        opt.split_after = int(split_after)
        opt.tree_seeds = seeds.ctypes.data_as(C.POINTER(C.c_uint64))
        h = C.c_void_p()
        _lib.check(_lib.lib().ah_build_subtrees(self._h, C.byref(opt), _ptr(ids), _ptr(offsets), C.byref(h)))
        return Forest(h, self.distance, self.dimensions)

    def create_index(self, forest: "Forest") -> "Index":
        """Mirror `forest` in HBM next to this dataset (ah_index_create)."""
        return Index(self, forest)

    # -- measurement -------------------------------------------------------------------------------
    def bench_scan(self, query_item: int, n: int, iterations: int, want_out: bool = False):
        out = np.zeros(n, dtype=np.float32) if want_out else None
        ms = C.c_double(0)
        _lib.check(_lib.lib().ah_bench_scan(self._h, query_item, n, iterations, _ptr(out), C.byref(ms)))
        return ms.value, out

    # -- plumbing ----------------------------------------------------------------------------------
    def _check_query(self, query) -> np.ndarray:
        q = _f32(query).ravel()
        if q.size != self.dimensions:  # src/reader.rs:64-69
            raise _lib.InvalidVecDimension(1, f"invalid vector dimensions, provided {q.size} but expected "
                                              f"{self.dimensions}")
        return q

    def close(self) -> None:
        if self._h:
            _lib.lib().ah_dataset_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Index:
    """Dataset + forest resident in HBM: the whole `Reader::nns_by_leaf` runs on device (ah_search_batch)."""

    def __init__(self, dataset: Dataset, forest: Optional["Forest"], view=None):
        self.dataset = dataset
        self._h = C.c_void_p()
        if view is not None:  # caller-owned arrays in the ah_forest_view shape (ah_index_create_from_view)
            _lib.check(_lib.lib().ah_index_create_from_view(dataset._h, C.cast(C.byref(view), C.POINTER(_lib.AhForestView)),
                                                            C.byref(self._h)))
        else:
            _lib.check(_lib.lib().ah_index_create(dataset._h, forest._h, C.byref(self._h)))

    def search(self, count: int, queries=None, items=None, search_k: int = 0, oversampling: int = 0, candidates=None,
               raw: bool = False, candidates_sorted: bool = False):
        """Batch of `QueryBuilder::by_vector` (queries: nq x dims) or `by_item` (items: nq ids).
        Returns a list (one entry per query) of [(id, distance), ...]; with raw=True the (ids, distances, counts)
        arrays of the C ABI (no per-result Python objects).  `candidates` = `QueryBuilder::candidates`."""
        ds = self.dataset
        if queries is not None:
            q = _f32(queries)
            if q.ndim == 1:
                q = q[None, :]
            if q.shape[1] != ds.dimensions:
                raise _lib.InvalidVecDimension(1, f"Invalid vector dimensions. Got {q.shape[1]} but expected "
                                                  f"{ds.dimensions}")
            nq, it = q.shape[0], None
        else:
            it = _u32(items).ravel()
            nq, q = it.size, None
        if candidates is None:
            filt = None
        elif candidates_sorted:  # already an ascending array of distinct ids (what a RoaringBitmap iterates)
            filt = _u32(candidates)
        else:
            filt = _u32(sorted(set(int(c) for c in candidates)))
        oi = np.zeros((nq, count), dtype=np.uint32)
        od = np.zeros((nq, count), dtype=np.float32)
        oc = np.zeros(nq, dtype=np.uint32)
        _lib.check(_lib.lib().ah_search_batch(self._h, _ptr(q), _ptr(it), nq, int(count), int(min(search_k, 2**62)),
                                              int(oversampling), _ptr(filt), 0 if filt is None else filt.size,
                                              0 if filt is None else 1, _ptr(oi), _ptr(od), _ptr(oc)))
        if raw:
            return oi, od, oc
        return [[(int(oi[i, j]), float(od[i, j])) for j in range(int(oc[i]))] for i in range(nq)]

    def stats(self, reset: bool = False) -> dict:
        """ah_index_search_stats: which descent tier / dedup path / re-rank path served the searches so far."""
        st = _lib.AhSearchStats()
        _lib.check(_lib.lib().ah_index_search_stats(self._h, C.byref(st), 1 if reset else 0))
        return {f: int(getattr(st, f)) for f, _ in _lib.AhSearchStats._fields_ if f != "reserved"}

    def route_items(self, item_ids: Sequence[int], tree_seeds: Sequence[int]) -> np.ndarray:
        """Incremental routing (src/writer.rs:1398-1459): [n_trees, n] forest-local Descendants node per item."""
        ids = _u32(item_ids)
        seeds = np.ascontiguousarray(tree_seeds, dtype=np.uint64)
        out = np.zeros((seeds.size, ids.size), dtype=np.uint32)
        _lib.check(_lib.lib().ah_route_items(self._h, _ptr(ids), ids.size, _ptr(seeds), _ptr(out)))
        return out

    def close(self) -> None:
        if self._h:
            _lib.lib().ah_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class StreamedForest:
    """What a sink of ah_build_forest_stream has seen, kept as dictionaries (test aid: small forests)."""

    def __init__(self, distance, dimensions):
        self.distance, self.dimensions = distance, dimensions
        self.splits, self.leaves, self.roots = {}, {}, None
        self.batches = []  # (kind, level, n_nodes, payload_len) in arrival order

    def take(self, b) -> int:
        n = int(b.n_nodes)
        self.batches.append((int(b.kind), int(b.level), n, int(b.payload_len)))
        payload = np.ctypeslib.as_array(b.payload, shape=(int(b.payload_len),)) if b.payload_len else np.zeros(0, np.uint8)
        hs, vs = self.distance.header_size(), self.distance.vector_size(self.dimensions)
        for i in range(n):
            nd = b.nodes[i]
            assert nd.kind == b.kind, f"node {nd.id} of kind {nd.kind} in a batch of kind {b.kind}"
            assert nd.id not in self.splits and nd.id not in self.leaves, \
                f"node {nd.id} (kind {nd.kind}, tree {nd.tree}, depth {nd.depth}, count {nd.count}) arrives twice; batches so far {self.batches[-6:]}"
            off = int(nd.payload_offset)
            if nd.kind == 2:
                nb = None
                if nd.has_normal:
                    rec = payload[off: off + int(b.normal_stride)]
                    h = rec[int(b.normal_header_offset): int(b.normal_header_offset) + hs]
                    v = rec[int(b.normal_vector_offset): int(b.normal_vector_offset) + vs]
                    nb = h.tobytes() + v.tobytes()  # the canonical form of Forest.canonical / the oracle: [header][vector]
                self.splits[int(nd.id)] = (nb, int(nd.left), int(nd.right), int(nd.tree), int(nd.depth), int(nd.count))
            else:
                ids = payload[off: off + 4 * int(nd.count)].view(np.uint32)
                self.leaves[int(nd.id)] = (tuple(int(x) for x in ids), int(nd.tree), int(nd.depth))
        return 0

    def canonical(self, tree: int):
        import sys
        sys.setrecursionlimit(100000)

        def rec(i):
            if i in self.leaves:
                return ("D", self.leaves[i][0])
            nb, left, right = self.splits[i][:3]
            return ("S", nb, rec(left), rec(right))
        return rec(int(self.roots[tree]))


class Forest:
    """Host-side result of one forest build (ah_forest): flat nodes, normals blob, descendant ids."""

    def __init__(self, handle: C.c_void_p, distance: type[Distance], dimensions: int):
        self._h = handle
        self.distance = distance
        self.dimensions = dimensions
        v = _lib.AhForestView()
        _lib.check(_lib.lib().ah_forest_view_get(self._h, C.byref(v)))
        self.n_trees = int(v.n_trees)
        n = int(v.n_nodes)
        node_dt = np.dtype([("kind", "u1"), ("has_normal", "u1"), ("reserved", "<u2"), ("tree", "<u4"), ("left", "<u4"),
                            ("right", "<u4"), ("offset", "<u8"), ("count", "<u4"), ("depth", "<u4")], align=True)
        assert node_dt.itemsize == C.sizeof(_lib.AhNode)

        def view(ptr, count, dtype):
            # zero-copy numpy view of a buffer owned by the ah_forest handle (kept alive by self)
            if not count:
                return np.zeros(0, dtype)
            nbytes = count * np.dtype(dtype).itemsize
            buf = (C.c_uint8 * nbytes).from_address(C.cast(ptr, C.c_void_p).value)
            return np.frombuffer(buf, dtype=dtype, count=count)

        self.roots = view(v.roots, self.n_trees, np.uint32)
        self.nodes = view(v.nodes, n, node_dt)
        self.normal_stride = int(v.normal_stride)
        self._vec_off, self._hdr_off = int(v.normal_vector_offset), int(v.normal_header_offset)
        self.normals = view(v.normals, int(v.normals_len), np.uint8)
        self.descendants = view(v.descendants, int(v.descendants_len), np.uint32)
        st = _lib.AhBuildStats()
        _lib.check(_lib.lib().ah_forest_stats(self._h, C.byref(st)))
        self.stats = {f: getattr(st, f) for f, _ in _lib.AhBuildStats._fields_}
        self.stats["margin_mode_launches"] = list(st.margin_mode_launches)

    def view_struct(self) -> "_lib.AhForestView":
        """The ah_forest_view of this forest (pointers into the handle's own buffers: valid while the forest lives) — what
        `Index(ds, None, view=...)` / ah_index_create_from_view takes."""
        v = _lib.AhForestView()
        _lib.check(_lib.lib().ah_forest_view_get(self._h, C.byref(v)))
        return v

    def digest(self):
        """(total, per-tree array) 64-bit content digests (ah_forest_digest): equal for equal forests whatever the
        margin mode, tuning or batching that built them."""
        per = np.zeros(self.n_trees, dtype=np.uint64)
        total = C.c_uint64(0)
        _lib.check(_lib.lib().ah_forest_digest(self._h, _ptr(per), C.byref(total)))
        return int(total.value), per

    def digest_keyed(self, tree_keys: Sequence[int]) -> np.ndarray:
        """Per-tree digests with `tree_keys[t]` in place of the tree's index inside this forest (ah_forest_digest_keyed): the
        digest of a tree is then the same whichever share / device built it."""
        keys = np.ascontiguousarray(tree_keys, dtype=np.uint64)
        assert keys.size == self.n_trees
        per = np.zeros(self.n_trees, dtype=np.uint64)
        _lib.check(_lib.lib().ah_forest_digest_keyed(self._h, _ptr(keys), _ptr(per)))
        return per

    def close(self) -> None:
        if self._h:
            self.roots = self.nodes = self.normals = self.descendants = None
            _lib.lib().ah_forest_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def normal_of(self, node: int):
        """(header f32[], vector codec bytes) of a split node, or None for `normal: None`."""
        nd = self.nodes[node]
        if nd["kind"] != 2 or not nd["has_normal"]:
            return None
        hs, vs = self.distance.header_size(), self.distance.vector_size(self.dimensions)
        raw = self.normals[int(nd["offset"]): int(nd["offset"]) + self.normal_stride]
        return (raw[self._hdr_off: self._hdr_off + hs].view(np.float32).copy(),
                raw[self._vec_off: self._vec_off + vs].copy())

    def descendants_of(self, node: int) -> np.ndarray:
        nd = self.nodes[node]
        return self.descendants[int(nd["offset"]): int(nd["offset"]) + int(nd["count"])]

    def canonical(self, tree: int):
        """Numbering-independent nested tuples; comparable with oracle.Tree.canonical()."""
        import sys
        sys.setrecursionlimit(100000)

        def rec(i):
            nd = self.nodes[i]
            if nd["kind"] == 1:
                return ("D", tuple(int(x) for x in self.descendants_of(i)))
            nb = None
            if nd["has_normal"]:
                h, v = self.normal_of(i)  # canonical form = [header][vector], the oracle's record layout
                nb = h.tobytes() + v.tobytes()
            return ("S", nb, rec(int(nd["left"])), rec(int(nd["right"])))

        return rec(int(self.roots[tree]))

    def tree_stats(self, tree: int):
        """`Reader::stats` per tree (src/reader.rs:210-252): depth, split nodes, dummy normals, descendants."""
        depth = splits = dummies = descs = 0
        stack = [(int(self.roots[tree]), 1)]
        while stack:
            i, d = stack.pop()
            nd = self.nodes[i]
            depth = max(depth, d)
            if nd["kind"] == 1:
                descs += 1
            else:
                splits += 1
                dummies += 0 if nd["has_normal"] else 1
                stack.append((int(nd["left"]), d + 1))
                stack.append((int(nd["right"]), d + 1))
        return {"depth": depth, "split_nodes": splits, "dummy_normals": dummies, "descendants": descs}
