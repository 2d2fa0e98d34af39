"""Host-side mirror of arroy's public surface for the hot path: `Writer`, `ArroyBuilder`, `Reader`,
`QueryBuilder` (src/writer.rs:37-485, src/reader.rs:26-298) over an in-memory item store instead of LMDB.

Scope: this mirror exists so that the parity tests read like the reference's own tests and so that the
C ABI is exercised the way arroy's Rust host code would drive it.  It implements the *full rebuild* path
(`Writer::build` with every tree missing) and the search; LMDB, the incremental insert/delete machinery,
upgrades and `available_memory` batching are out of scope.  All distance / margin / split
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


class _IndexState:
    def __init__(self):
        self.items: Dict[ItemId, np.ndarray] = {}
        self.updated: set = set()          # the `Updated` key set (src/writer.rs:391)
        self.metadata: Optional[dict] = None
        self.dataset: Optional[Dataset] = None
        self.forest: Optional[Forest] = None
        self.index = None  # arroy_amd.Index: dataset + forest resident in HBM


class Database:
    """Stand-in for `Database<D>` (src/lib.rs:156) + its LMDB environment: one item store per index."""

    def __init__(self, distance: type[Distance]):
        self.distance = distance
        self._indexes: Dict[int, _IndexState] = {}

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

    def build(self) -> None:  # Writer::build, src/writer.rs:487-629 (full rebuild)
        w, st = self._w, self._w._st
        if self._cancel is not None and self._cancel():
            raise _lib.BuildCancelled(2, "build cancelled")
        dist = w.database.distance
        ids = np.array(sorted(st.items), dtype=np.uint32)
        n = ids.size
        st.dataset = st.forest = st.index = None
        if n:
            vecs = np.stack([st.items[int(i)] for i in ids])
            ds = Dataset(dist, w.dimensions, n)
            ds.upload_vectors(ids, vecs)
            if dist.metric == 3:
                ds.preprocess_dot()  # pre_process_items, src/writer.rs:964-976
            ds.finalize()
            n_trees = target_n_trees(self._n_trees, w.dimensions, n, 0)
            seeds = [self._rng.getrandbits(64) for _ in range(n_trees)]  # one RNG per root task (:575)
            forest = ds.build_forest(seeds, split_after=self._split_after or 0, cancel=self._cancel,
                                     progress=self._progress)
            st.dataset, st.forest = ds, forest
            st.index = ds.create_index(forest)  # forest mirrored in HBM for the on-device search
            roots = [int(r) for r in forest.roots]
        else:
            roots = []
        st.metadata = {"dimensions": w.dimensions, "items": [int(i) for i in ids], "roots": roots,
                       "distance": dist.name}  # src/writer.rs:611-626
        st.updated.clear()


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
        f = self._st.forest
        return {"leaf": self.n_items(), "tree_stats": [f.tree_stats(t) for t in range(f.n_trees)] if f else []}

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
