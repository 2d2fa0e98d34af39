"""Host-side mirror of arroy's public surface for the hot path: `Writer`, `ArroyBuilder`, `Reader`,
`QueryBuilder` (src/writer.rs:37-485, src/reader.rs:26-298) over an in-memory item store instead of LMDB.

Scope: this mirror exists so that the parity tests read like the reference's own tests and so that the
C ABI is exercised the way arroy's Rust host code would drive it.  It implements `Writer::build` — the full
build, and the incremental one (updated items leave the trees, new ones are routed by `ah_route_items`,
overgrown descendants are re-split by `ah_build_subtrees`, trees are added / dropped per `target_n_trees`) —
and the search; LMDB, upgrades and `available_memory` batching are out of scope.  All distance / margin / split
arithmetic is done by libarroy_hip.so; the host only keeps dictionaries, a priority queue and id lists —
exactly the split of work of the Rust integration.
"""
from __future__ import annotations

import math
import random
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .dataset import Dataset, Forest
from .distances import Distance

ItemId = int


class MissingMetadata(RuntimeError):
    """arroy::Error::MissingMetadata(index) (src/error.rs:48-49)."""

    def __init__(self, index: int):
        super().__init__(f"Metadata are missing on index {index}, You must build your database before attempting "
                         "to read it")
        self.index = index


class NeedBuild(RuntimeError):
    """arroy::Error::NeedBuild(index) (src/error.rs:51-52)."""

    def __init__(self, index: int):
        super().__init__(f"The trees have not been built after an update on index {index}")
        self.index = index


class UnmatchingDistance(RuntimeError):
    """arroy::Error::UnmatchingDistance { expected, received } (src/error.rs:34-41)."""

    def __init__(self, expected: str, received: str):
        super().__init__(f"Invalid distance provided. Got {received} but expected {expected}")
        self.expected, self.received = expected, received


class InvalidVecDimension(_lib.InvalidVecDimension):
    """arroy::Error::InvalidVecDimension { expected, received } (src/error.rs:17-23)."""

    def __init__(self, expected: int, received: int):
        _lib.ArroyHipError.__init__(self, 1, f"Invalid vector dimensions. Got {received} but expected {expected}")
        self.expected, self.received = expected, received

    def __str__(self) -> str:
        return self.message


def target_n_trees(n_trees: Optional[int], dimensions: int, n_items: int, n_roots: int) -> int:
    """`target_n_trees` (src/writer.rs:1358-1394): explicit option, else the fitted formula, with the
    "do not shrink by less than 20 %" rule."""
    if n_trees is not None:
        return int(n_trees)
    nb_vec = float(n_items)
    if nb_vec < 10_000.0:
        nb = 2.0 ** (math.log2(nb_vec) - 6.0) if nb_vec > 0 else 0.0
    else:
        nb = 2.0 ** (math.log10(nb_vec) + math.log10(float(dimensions)) + (768.0 / float(dimensions)) ** 4.0)
    nb_trees = int(math.ceil(nb))
    if n_roots > nb_trees:
        to_remove = n_roots - nb_trees
        if nb_trees == 0 or (to_remove / nb_trees) < 0.20:
            nb_trees = n_roots
    return nb_trees


class TreeStore:
    """Host-side tree nodes — what arroy keeps in LMDB under `Key::tree` (src/node.rs:216-241):
    id -> ("D", ascending ids) | ("S", left, right, header f32[], vector bytes or None)."""

    def __init__(self):
        self.nodes: Dict[int, tuple] = {}
        self.roots: List[int] = []
        self._next = 0
        self._available: List[int] = []

    def begin_build(self) -> None:
        """`ConcurrentNodeIds::new(used_tree_node)` (src/parallel.rs:222-237, src/writer.rs:516-518): taken BEFORE the
        build deletes anything — ids freed by earlier builds are handed out first (ascending), ids freed by this
        build only become available to the next one."""
        self._next = max(self.nodes) + 1 if self.nodes else 0
        self._available = [i for i in range(self._next) if i not in self.nodes]

    def next_id(self) -> int:  # ConcurrentNodeIds::next (src/parallel.rs:239-254)
        if self._available:
            return self._available.pop(0)
        self._next += 1
        return self._next - 1

    def delete_tree(self, node: int) -> None:  # src/writer.rs:1263-1277
        for nid in self.subtree_ids(node):
            del self.nodes[nid]

    def delete_items(self, node: int, to_delete: set, split_after: int):
        """`delete_items_in_file` (src/writer.rs:1021-1114): remove `to_delete` below `node`; a split whose child
        became empty is replaced by the other child, two descendants that fit together are merged into their parent.
        Returns (new node id, ids of the branch if it is a single Descendants node else None)."""
        nd = self.nodes[node]
        if nd[0] == "D":
            kept = np.array([i for i in nd[1] if int(i) not in to_delete], dtype=np.uint32)
            if len(kept) != len(nd[1]):
                self.nodes[node] = ("D", kept)
            return node, kept
        _, left, right, hdr, vec = nd
        new_left, left_items = self.delete_items(left, to_delete, split_after)
        new_right, right_items = self.delete_items(right, to_delete, split_after)
        if left_items is not None and len(left_items) == 0:
            self.nodes.pop(new_left, None)
            self.nodes.pop(node, None)
            return new_right, right_items
        if right_items is not None and len(right_items) == 0:
            self.nodes.pop(new_right, None)
            self.nodes.pop(node, None)
            return new_left, left_items
        if left_items is not None and right_items is not None and len(left_items) + len(right_items) <= split_after:
            total = np.union1d(left_items, right_items).astype(np.uint32)
            self.nodes.pop(new_left, None)
            self.nodes.pop(new_right, None)
            self.nodes[node] = ("D", total)
            return node, total
        if new_left != left or new_right != right:
            self.nodes[node] = ("S", new_left, new_right, hdr, vec)
        return node, None

    def import_tree(self, forest: Forest, tree: int, root_id: Optional[int] = None) -> int:
        """Copy tree `tree` of a freshly built ah_forest; children get fresh ids before their parent (the order
        `make_tree_in_file` allocates them, src/writer.rs:1235-1258), the root takes `root_id` when given
        (`Some(descendant_id)`, src/writer.rs:693-702)."""
        order, stack = [], [int(forest.roots[tree])]
        while stack:  # reverse post-order, then reversed: children before parents
            i = stack.pop()
            order.append(i)
            nd = forest.nodes[i]
            if nd["kind"] == 2:
                stack += [int(nd["left"]), int(nd["right"])]
        ids: Dict[int, int] = {}
        for i in reversed(order):
            nd = forest.nodes[i]
            is_root = i == int(forest.roots[tree])
            ids[i] = root_id if (is_root and root_id is not None) else self.next_id()
            if nd["kind"] == 1:
                self.nodes[ids[i]] = ("D", forest.descendants_of(i).copy())
            else:
                normal = forest.normal_of(i)
                hdr, vec = normal if normal is not None else (np.zeros(forest.distance.header_size() // 4, np.float32), None)
                self.nodes[ids[i]] = ("S", ids[int(nd["left"])], ids[int(nd["right"])], hdr,
                                      None if vec is None else vec.tobytes())
        return ids[int(forest.roots[tree])]

    def subtree_ids(self, root: int) -> List[int]:
        out, stack = [], [root]
        while stack:
            i = stack.pop()
            out.append(i)
            nd = self.nodes[i]
            if nd[0] == "S":
                stack += [nd[1], nd[2]]
        return out

    def to_view(self, distance: type[Distance], dimensions: int):
        """Dense arrays in the ah_forest_view shape (records [header][vector]); returns (view, keepalive)."""
        import ctypes as C
        order = sorted(self.nodes)
        dense = {nid: i for i, nid in enumerate(order)}
        hs, vs = distance.header_size(), distance.vector_size(dimensions)
        node_dt = np.dtype([("kind", "u1"), ("has_normal", "u1"), ("reserved", "<u2"), ("tree", "<u4"), ("left", "<u4"),
                            ("right", "<u4"), ("offset", "<u8"), ("count", "<u4"), ("depth", "<u4")], align=True)
        nodes = np.zeros(len(order), dtype=node_dt)
        normals, desc = bytearray(), []
        n_desc = 0
        for nid in order:
            nd, i = self.nodes[nid], dense[nid]
            if nd[0] == "D":
                nodes[i] = (1, 0, 0, 0, 0, 0, n_desc, len(nd[1]), 0)
                desc.append(np.asarray(nd[1], dtype=np.uint32))
                n_desc += len(nd[1])
            else:
                has = nd[4] is not None
                nodes[i] = (2, 1 if has else 0, 0, 0, dense[nd[1]], dense[nd[2]], len(normals), 0, 0)
                if has:
                    normals += np.asarray(nd[3], dtype=np.float32).tobytes().ljust(hs, b"\0")[:hs] + nd[4]
        normals_a = np.frombuffer(bytes(normals), dtype=np.uint8).copy() if normals else np.zeros(1, np.uint8)
        desc_a = np.concatenate(desc).astype(np.uint32) if desc and n_desc else np.zeros(1, np.uint32)
        roots_a = np.array([dense[r] for r in self.roots], dtype=np.uint32) if self.roots else np.zeros(1, np.uint32)
        v = _lib.AhForestView()
        v.n_trees, v.n_nodes = len(self.roots), len(order)
        v.roots = roots_a.ctypes.data_as(C.POINTER(C.c_uint32))
        v.nodes = C.cast(nodes.ctypes.data, C.POINTER(_lib.AhNode))
        v.normals = normals_a.ctypes.data_as(C.POINTER(C.c_uint8))
        v.normals_len = len(normals)
        v.normal_stride, v.normal_vector_offset, v.normal_header_offset = hs + vs, hs, 0
        v.descendants = desc_a.ctypes.data_as(C.POINTER(C.c_uint32))
        v.descendants_len = n_desc
        return v, (nodes, normals_a, desc_a, roots_a, dense)

    def stats(self, root: int) -> dict:  # `Reader::stats` per tree (src/reader.rs:210-252)
        depth = splits = dummies = descs = 0
        stack = [(root, 1)]
        while stack:
            i, d = stack.pop()
            depth = max(depth, d)
            nd = self.nodes[i]
            if nd[0] == "D":
                descs += 1
            else:
                splits += 1
                dummies += 0 if nd[4] is not None else 1
                stack += [(nd[1], d + 1), (nd[2], d + 1)]
        return {"depth": depth, "split_nodes": splits, "dummy_normals": dummies, "descendants": descs}


class _IndexState:
    def __init__(self):
        self.items: Dict[ItemId, np.ndarray] = {}
        self.updated: set = set()          # the `Updated` key set (src/writer.rs:391)
        self.metadata: Optional[dict] = None
        self.dataset: Optional[Dataset] = None
        self.trees: Optional[TreeStore] = None
        self.index = None    # arroy_amd.Index: dataset + tree nodes resident in HBM
        self._keep = None    # arrays the index view was built from


class Database:
    """Stand-in for `Database<D>` (src/lib.rs:156) + its LMDB environment: one item store per index."""

    def __init__(self, distance: type[Distance]):
        self.distance = distance
        self._indexes: Dict[int, _IndexState] = {}

    def remap_data_type(self, distance: type[Distance]) -> "Database":
        """`database.remap_data_type::<NodeCodec<D2>>()`: the same store seen through another distance type."""
        other = Database(distance)
        other._indexes = self._indexes
        return other

    def _state(self, index: int) -> _IndexState:
        return self._indexes.setdefault(index, _IndexState())


class Writer:
    """`Writer<D>` (src/writer.rs:271-485)."""

    def __init__(self, database: Database, index: int, dimensions: int):
        self.database, self.index, self.dimensions = database, index, int(dimensions)
        self._st = database._state(index)

    def add_item(self, item: ItemId, vector: Sequence[float]) -> None:  # src/writer.rs:380-394
        v = np.ascontiguousarray(vector, dtype=np.float32).ravel()
        if v.size != self.dimensions:
            raise InvalidVecDimension(self.dimensions, v.size)
        self._st.items[int(item)] = v.copy()
        self._st.updated.add(int(item))

    append_item = add_item  # same observable effect without an LMDB cursor (src/writer.rs:396-432)

    def del_item(self, item: ItemId) -> bool:  # src/writer.rs:434-443
        if int(item) in self._st.items:
            del self._st.items[int(item)]
            self._st.updated.add(int(item))
            return True
        return False

    def clear(self) -> None:  # src/writer.rs:445-470
        self._st.__init__()

    def is_empty(self) -> bool:
        return not self._st.items

    def contains_item(self, item: ItemId) -> bool:
        return int(item) in self._st.items

    def need_build(self) -> bool:  # src/writer.rs:355-363
        return bool(self._st.updated) or self._st.metadata is None

    def builder(self, rng: Optional[random.Random] = None) -> "ArroyBuilder":
        return ArroyBuilder(self, rng if rng is not None else random.Random())


class ArroyBuilder:
    """`ArroyBuilder` (src/writer.rs:37-265): n_trees / split_after / cancel / progress / build."""

    def __init__(self, writer: Writer, rng: random.Random):
        self._w, self._rng = writer, rng
        self._n_trees: Optional[int] = None
        self._split_after: Optional[int] = None
        self._cancel: Optional[Callable[[], bool]] = None
        self._progress: Optional[Callable] = None

    def n_trees(self, n: int) -> "ArroyBuilder":
        self._n_trees = int(n)
        return self

    def split_after(self, n: int) -> "ArroyBuilder":
        self._split_after = int(n)
        return self

    def available_memory(self, _bytes: int) -> "ArroyBuilder":
        return self  # HBM-resident build: the page-budgeted batching of src/writer.rs:685-723 does not apply

    def cancel(self, fn: Callable[[], bool]) -> "ArroyBuilder":
        self._cancel = fn
        return self

    def progress(self, fn: Callable) -> "ArroyBuilder":
        self._progress = fn
        return self

    def build(self) -> None:
        """`Writer::build` (src/writer.rs:487-629): a full build when the index has no trees yet, otherwise the
        incremental path — remove updated items from the trees (:525), route the new / changed ones through the
        existing planes (:541-542, `ah_route_items`), re-split the descendants that outgrew `split_after` (:660-739,
        `ah_build_subtrees`), then add or drop whole trees to reach `target_n_trees` (:521-524, 556-561)."""
        w, st = self._w, self._w._st
        if self._cancel is not None and self._cancel():
            raise _lib.BuildCancelled(2, "build cancelled")
        dist = w.database.distance
        ids = np.array(sorted(st.items), dtype=np.uint32)
        n = ids.size
        split_after = self._split_after or w.dimensions
        st.index = None
        ds = None
        if n:
            vecs = np.stack([st.items[int(i)] for i in ids])
            ds = Dataset(dist, w.dimensions, n)
            ds.upload_vectors(ids, vecs)
            if dist.metric == 3:
                ds.preprocess_dot()  # pre_process_items, src/writer.rs:964-976
            ds.finalize()
        st.dataset = ds
        if n <= split_after:
            # clear_db_and_create_a_single_leaf (src/writer.rs:916-962): ONE Descendants root, whatever n_trees says
            st.trees = TreeStore()
            if n:
                root = st.trees.next_id()
                st.trees.nodes[root] = ("D", ids.copy())
                st.trees.roots = [root]
        elif st.trees is None or not st.trees.roots or st.metadata is None:
            st.trees = TreeStore()
            n_trees = target_n_trees(self._n_trees, w.dimensions, n, 0)
            self._add_trees(ds, st.trees, n_trees, split_after)
        else:
            self._incremental(ds, st, ids, split_after)
        if n:
            view, keep = st.trees.to_view(dist, w.dimensions)
            from .dataset import Index
            st.index, st._keep = Index(ds, None, view=view), keep
        st.metadata = {"dimensions": w.dimensions, "items": [int(i) for i in ids], "roots": list(st.trees.roots),
                       "distance": dist.name}  # src/writer.rs:611-626
        st.updated.clear()

    def _seeds(self, count: int) -> List[int]:
        return [self._rng.getrandbits(64) for _ in range(count)]  # one RNG per task (src/writer.rs:575,795)

    def _add_trees(self, ds: Dataset, trees: TreeStore, count: int, split_after: int) -> None:
        if count <= 0:
            return
        # arroy's formula (src/writer.rs:1371-1380) explodes for >= 10 000 items of fewer than 768 dimensions
        # ((768 / dims)^4 in the exponent); the reference then builds that many trees, and so does this mirror
        # (ah_node.tree is 32 bits since ABI v2) — in slices, so that the host-side forest stays bounded.
        seeds = self._seeds(count)
        for lo in range(0, count, 4096):
            forest = ds.build_forest(seeds[lo:lo + 4096], split_after=split_after, cancel=self._cancel,
                                     progress=self._progress)
            for t in range(forest.n_trees):
                root = trees.next_id()  # roots are allocated before their subtree (src/writer.rs:556-561)
                trees.import_tree(forest, t, root_id=root)
                trees.roots.append(root)
            forest.close()

    def _incremental(self, ds: Dataset, st: "_IndexState", ids: np.ndarray, split_after: int) -> None:
        from .dataset import Index
        w, trees, dist = self._w, st.trees, self._w.database.distance
        present = set(int(i) for i in ids)
        to_delete = np.array(sorted(st.updated), dtype=np.uint32)                 # :504
        to_insert = np.array(sorted(i for i in st.updated if i in present), dtype=np.uint32)  # :505
        import sys
        sys.setrecursionlimit(max(sys.getrecursionlimit(), 100000))
        trees.begin_build()  # node ids in use are snapshotted before anything is deleted (:516-518)
        # target_n_trees / delete_extra_trees (:521-522, 631-655): the oldest tree first, `roots.swap_remove(0)`
        want = target_n_trees(self._n_trees, w.dimensions, len(ids), len(trees.roots))
        while len(trees.roots) > want:
            root = trees.roots[0]
            trees.roots[0] = trees.roots[-1]
            trees.roots.pop()
            trees.delete_tree(root)
        # delete_items_from_trees (:525, 979-1114): updated ids leave the trees, emptied / shrunken branches collapse
        if to_delete.size:
            gone = set(int(i) for i in to_delete)
            trees.roots = [trees.delete_items(root, gone, split_after)[0] for root in trees.roots]
        trees.roots.sort()
        # insert_items_in_current_trees (:541-542): one ah_route_items call for all trees
        grown: Dict[int, List[int]] = {}
        if to_insert.size and trees.roots:
            view, keep = trees.to_view(dist, w.dimensions)
            dense = keep[4]
            back = {i: nid for nid, i in dense.items()}
            old = Index(ds, None, view=view)
            leaf_of = old.route_items(to_insert, self._seeds(len(trees.roots)))
            old.close()
            for t in range(leaf_of.shape[0]):
                for i, leaf in enumerate(leaf_of[t]):
                    grown.setdefault(back[int(leaf)], []).append(int(to_insert[i]))
            for nid, extra in grown.items():
                trees.nodes[nid] = ("D", np.union1d(trees.nodes[nid][1], np.array(extra, dtype=np.uint32)).astype(np.uint32))
        # the descendants the routing touched and that no longer fit (`fit_in_descendant`, :474-477, 787-795) are
        # re-split (incremental_index_large_descendant, :660-739); untouched ones are left alone, whatever their size
        large = [nid for nid in sorted(grown) if len(trees.nodes[nid][1]) > split_after]
        if large:
            forest = ds.build_subtrees([trees.nodes[nid][1] for nid in large], self._seeds(len(large)), split_after)
            for t, nid in enumerate(large):
                trees.import_tree(forest, t, root_id=nid)  # the sub-tree's root keeps the descendant's id (:693-702)
        # missing trees (:556-561)
        self._add_trees(ds, trees, want - len(trees.roots), split_after)


class Reader:
    """`Reader<D>` (src/reader.rs:128-298)."""

    def __init__(self, database: Database, index: int, st: _IndexState):
        self.database, self.index, self._st = database, index, st
        self.distance = database.distance

    @classmethod
    def open(cls, database: Database, index: int) -> "Reader":  # src/reader.rs:138-200
        st = database._indexes.get(index)
        if st is None or st.metadata is None:
            raise MissingMetadata(index)
        if database.distance.name != st.metadata["distance"]:  # src/reader.rs:153-158
            raise UnmatchingDistance(st.metadata["distance"], database.distance.name)
        if st.updated:
            raise NeedBuild(index)
        return cls(database, index, st)

    def dimensions(self) -> int:
        return self._st.metadata["dimensions"]

    def n_trees(self) -> int:
        return len(self._st.metadata["roots"])

    def n_items(self) -> int:
        return len(self._st.metadata["items"])

    def item_ids(self) -> List[int]:
        return list(self._st.metadata["items"])

    def is_empty(self) -> bool:
        return self.n_items() == 0

    def contains_item(self, item: ItemId) -> bool:
        return int(item) in self._st.items

    def item_vector(self, item: ItemId) -> Optional[np.ndarray]:  # src/reader.rs:266-276
        if int(item) not in self._st.items:
            return None
        return self._st.dataset.item_vector(int(item))

    def stats(self) -> dict:  # src/reader.rs:210-252
        tr = self._st.trees
        return {"leaf": self.n_items(), "tree_stats": [tr.stats(r) for r in tr.roots] if tr else []}

    def nns(self, count: int) -> "QueryBuilder":  # src/reader.rs:296-298
        return QueryBuilder(self, int(count))


class QueryBuilder:
    """`QueryBuilder` (src/reader.rs:26-124)."""

    def __init__(self, reader: Reader, count: int):
        self._r, self._count = reader, count
        self._search_k: Optional[int] = None
        self._oversampling: Optional[int] = None
        self._candidates: Optional[set] = None

    def search_k(self, n: int) -> "QueryBuilder":
        self._search_k = int(n)
        return self

    def oversampling(self, n: int) -> "QueryBuilder":
        self._oversampling = int(n)
        return self

    def candidates(self, ids: Iterable[int]) -> "QueryBuilder":
        self._candidates = set(int(i) for i in ids)
        return self

    def by_item(self, item: ItemId) -> Optional[List[Tuple[int, float]]]:  # src/reader.rs:46-51
        if int(item) not in self._r._st.items:
            return None
        return self._nns(item=int(item))

    def by_vector(self, vector: Sequence[float]) -> List[Tuple[int, float]]:  # src/reader.rs:64-75
        v = np.ascontiguousarray(vector, dtype=np.float32).ravel()
        if v.size != self._r.dimensions():
            raise InvalidVecDimension(self._r.dimensions(), v.size)
        return self._nns(vector=v)

    # nns_by_leaf, src/reader.rs:317-401 — the whole thing runs on device (ah_search_batch)
    def _nns(self, vector: Optional[np.ndarray] = None, item: Optional[int] = None):
        r, st = self._r, self._r._st
        if r.is_empty():
            return []
        res = st.index.search(self._count, queries=None if vector is None else vector[None, :],
                              items=None if item is None else [item],
                              search_k=0 if self._search_k is None else self._search_k,
                              oversampling=0 if self._oversampling is None else self._oversampling,
                              candidates=self._candidates)
        return res[0]
