"""Replay of arroy's `Writer::build` — full AND incremental — in the reference's own order and with the reference's own
randomness, on top of the CPU oracle (test infrastructure, no GPU).  It exists to replay the reference's inline insta
snapshots (src/tests/writer.rs) step by step: what is pinned is the oracle's arithmetic (create_split, side), the
restatement of rand 0.8 and — new here — the host-side bookkeeping of incremental builds that arroy_amd/index.py
mirrors: removal of updated items with node collapsing (`delete_items_in_file`, src/writer.rs:1021-1114), routing of
new items through the existing planes (:1398-1459), re-splitting of the descendants that outgrew `split_after` with
the sub-tree root keeping its node id (:660-739), node-id reuse (`ConcurrentNodeIds`, src/parallel.rs:206-254) and the
add / drop of whole trees (:521-524, 556-561, 631-655).

Nodes: {id: ("D", [ids]) | ("S", left, right, header f32[], vector bytes | None)}.
"""
import ctypes as C
import math

import numpy as np

from oracle import oracle as O


def target_n_trees(n_trees, dimensions, n_items, n_roots):  # src/writer.rs:1358-1394
    if n_trees is not None:
        return int(n_trees)
    nb_vec = float(n_items)
    if nb_vec < 10_000.0:
        nb = 2.0 ** (math.log2(nb_vec) - 6.0) if nb_vec > 0 else 0.0
    else:
        nb = 2.0 ** (math.log10(nb_vec) + math.log10(float(dimensions)) + (768.0 / float(dimensions)) ** 4.0)
    nb = int(math.ceil(nb))
    if n_roots > nb and (nb == 0 or (n_roots - nb) / nb < 0.20):
        nb = n_roots
    return nb


class IntMapOrder:
    """Iteration order of `nohash::IntMap<u32, _>` = std `HashMap` (hashbrown, SSE2 groups of 16) with the identity
    hash, as far as the reference's build depends on it (the order in which `descendants` is walked decides which task
    gets which seed, src/writer.rs:778-796).  Restated from hashbrown 0.14's `RawTable`: buckets are a power of two
    (4, 8, then next_power_of_two(cap * 8 / 7)), a new key goes to the first empty control byte of the 16-wide window
    starting at `hash & mask` (windows advance triangularly), growth re-inserts the old buckets in index order, and
    iteration walks the buckets in index order.  No deletions are needed here."""

    def __init__(self):
        self.buckets = []      # key or None per bucket
        self.items = 0
        self.growth_left = 0

    @staticmethod
    def _cap(mask):
        return mask if mask < 8 else ((mask + 1) // 8) * 7

    def _slot(self, buckets, key):
        n = len(buckets)
        mask = n - 1
        pos, stride = key & mask, 0
        while True:
            for bit in range(16):
                idx = pos + bit
                if n < 16 and idx >= n:
                    # control bytes n..15 of a small table are always EMPTY: the reference then falls back to the
                    # first empty bucket of the aligned group at 0
                    if idx < 16:
                        for j in range(n):
                            if buckets[j] is None:
                                return j
                    idx -= 16  # mirrored tail
                    if idx >= n:
                        continue
                if buckets[idx & mask] is None:
                    return idx & mask
            stride += 16
            pos = (pos + stride) & mask

    def _resize(self, capacity):
        if capacity < 8:
            n = 4 if capacity < 4 else 8
        else:
            adj = capacity * 8 // 7
            n = 1 << (adj - 1).bit_length()
        new = [None] * n
        for key in self.buckets:
            if key is not None:
                new[self._slot(new, key)] = key
        self.buckets = new
        self.growth_left = self._cap(n - 1) - self.items

    def insert(self, key):
        if key in self.buckets:
            return
        if self.growth_left == 0:
            full = self._cap(len(self.buckets) - 1) if self.buckets else 0
            self._resize(max(self.items + 1, full + 1))
        self.buckets[self._slot(self.buckets, key)] = key
        self.items += 1
        self.growth_left -= 1

    def order(self):
        return [k for k in self.buckets if k is not None]


class OracleBackend:
    """create_split / side by the CPU oracle (item ids in, item ids out)."""

    def __init__(self, metric, item_ids, vecs):
        ids = None if item_ids == list(range(len(item_ids))) else np.array(item_ids, dtype=np.uint32)
        self.data = O.Data(metric, vecs, ids=ids)
        self.row_of = {item: r for r, item in enumerate(item_ids)}

    def create_split(self, sample_ids):
        nv, nh = self.data.create_split(np.array([self.row_of[int(i)] for i in sample_ids], dtype=np.uint32))
        return nv.tobytes(), np.asarray(nh, dtype=np.float32)

    def split_sides(self, vec: bytes, hdr, ids):
        nh = np.zeros(2, dtype=np.float32)
        nh[: len(hdr)] = hdr
        rows = np.array([self.row_of[int(i)] for i in ids], dtype=np.uint32)
        return self.data.split_sides(np.frombuffer(vec, dtype=np.uint8), nh, rows)[0]


class GpuBackend:
    """The same two operations through the C ABI of libarroy_hip.so (`ah_create_split`, `ah_split_sides`)."""

    def __init__(self, distance_cls, dims, item_ids, vecs):
        from arroy_amd import Dataset
        self.ds = Dataset(distance_cls, dims, len(item_ids))
        self.ds.upload_vectors(np.array(item_ids, dtype=np.uint32), vecs)
        if distance_cls.metric == 3:
            self.ds.preprocess_dot()
        self.ds.finalize()

    def create_split(self, sample_ids):
        nv, nh = self.ds.create_split(np.array(sample_ids, dtype=np.uint32))
        return nv.tobytes(), np.asarray(nh, dtype=np.float32)

    def split_sides(self, vec: bytes, hdr, ids):
        nh = np.zeros(2, dtype=np.float32)
        nh[: len(hdr)] = hdr
        sides, _, _ = self.ds.split_sides(np.frombuffer(vec, dtype=np.uint8), nh[: max(1, len(hdr))],
                                          sorted_ids=np.array(ids, dtype=np.uint32), want_margins=False)
        return sides


class RefWriter:
    def __init__(self, metric: int, dimensions: int, backend_factory=None):
        """backend_factory(item_ids, vecs) -> object with create_split / split_sides; None = the oracle, with the
        sub-trees built by its C restatement of make_tree_in_file (fast path of the CPU tests)."""
        self.backend_factory = backend_factory
        self.metric, self.dims = metric, dimensions
        self.items = {}
        self.updated = set()
        self.nodes = {}
        self.roots = None  # None = no metadata yet

    # ---- Writer::add_item / del_item (src/writer.rs:271-377): both mark the id as updated ----
    def add_item(self, item: int, vector):
        self.items[int(item)] = np.asarray(vector, dtype=np.float32)
        self.updated.add(int(item))

    def del_item(self, item: int) -> bool:
        if int(item) in self.items:
            del self.items[int(item)]
            self.updated.add(int(item))
            return True
        return False

    # ---- src/writer.rs:1263-1277 ----
    def _delete_tree(self, node):
        nd = self.nodes.pop(node)
        if nd[0] == "S":
            self._delete_tree(nd[1])
            self._delete_tree(nd[2])

    # ---- delete_items_in_file, src/writer.rs:1021-1114: (new node id, items of the branch if it is one descendant) ----
    def _delete_items(self, node, to_delete, split_after):
        nd = self.nodes[node]
        if nd[0] == "D":
            kept = [i for i in nd[1] if i not in to_delete]
            if len(kept) != len(nd[1]):
                self.nodes[node] = ("D", kept)
            return node, kept
        _, left, right, hdr, vec = nd
        new_left, left_items = self._delete_items(left, to_delete, split_after)
        new_right, right_items = self._delete_items(right, to_delete, split_after)
        if left_items is not None and len(left_items) == 0:
            self.nodes.pop(new_left, None)
            self.nodes.pop(node, None)
            return new_right, right_items
        if right_items is not None and len(right_items) == 0:
            self.nodes.pop(new_right, None)
            self.nodes.pop(node, None)
            return new_left, left_items
        if left_items is not None and right_items is not None:
            if len(left_items) + len(right_items) <= split_after:
                total = sorted(set(left_items) | set(right_items))
                self.nodes.pop(new_left, None)
                self.nodes.pop(new_right, None)
                self.nodes[node] = ("D", total)
                return node, total
        if new_left != left or new_right != right:
            self.nodes[node] = ("S", new_left, new_right, hdr, vec)
        return node, None

    # ---- insert_items_in_descendants_from_frozen_reader, src/writer.rs:1398-1459 ----
    def _route(self, be, node, to_insert, out):
        nd = self.nodes[node]
        if nd[0] == "D":
            out[node] = sorted(set(nd[1]) | set(to_insert))
            return
        _, left, right, hdr, vec = nd
        assert vec is not None, "normal: None needs the per-tree rng (not exercised by the replayed snapshots)"
        sides = be.split_sides(vec, hdr, to_insert)
        left_ids = [i for i, s in zip(to_insert, sides) if s == 0]
        right_ids = [i for i, s in zip(to_insert, sides) if s == 1]
        if left_ids:
            self._route(be, left, left_ids, out)
        if right_ids:
            self._route(be, right, right_ids, out)

    # ---- fit_in_memory, src/writer.rs:1536-1584 ----
    def _fit_in_memory(self, memory, to_insert, rng):
        """`to_insert` (ascending list = the RoaringBitmap) is consumed; returns the next batch or None."""
        if not to_insert:
            return None
        if len(to_insert) <= self.dims:
            out, to_insert[:] = list(to_insert), []
            return out
        page = 4096
        nb_page_allowed = int(memory // page)
        largest = 4 * O.header_floats(self.metric) + (self.dims // 64 if O.is_bq(self.metric) else 4 * self.dims)
        per_page, page_per_item = page // largest, -(-largest // page)
        if per_page > 1:
            nb_items = nb_page_allowed * per_page
        elif page_per_item > 1:
            nb_items = nb_page_allowed // page_per_item
        else:
            nb_items = nb_page_allowed
        if nb_items <= self.dims:
            nb_items = self.dims + 1
        if nb_items >= len(to_insert):
            out, to_insert[:] = list(to_insert), []
            return out
        items = []
        for _ in range(nb_items):  # `rng.gen_range(0..to_insert.len())` on u64 (rand 0.8 UniformInt::sample_single)
            rng_range = len(to_insert)
            zone = ((rng_range << (64 - rng_range.bit_length())) - 1) & 0xFFFFFFFFFFFFFFFF
            while True:
                lo32 = rng.next_u32()
                v = lo32 | (rng.next_u32() << 32)
                prod = v * rng_range
                if (prod & 0xFFFFFFFFFFFFFFFF) <= zone:
                    idx = prod >> 64
                    break
            items.append(to_insert.pop(idx))  # RoaringBitmap::select(idx), then remove
        return sorted(items)

    # ---- insert_descendants_in_file_and_spawn_tasks, src/writer.rs:744-844 ----
    def _walk(self, rng, order, dmap, stack, split_after):
        for node in order:
            items = dmap[node]
            if len(items) <= split_after:
                self.nodes[node] = ("D", sorted(items))
            else:
                stack.append((rng.gen_seed(), node, sorted(items)))

    # ---- insert_items_in_descendants_from_tmpfile, src/writer.rs:1463-1531 ----
    def _route_tmp(self, be, node, to_insert, local):
        if node in local:
            local[node] = sorted(set(local[node]) | set(to_insert))
            return
        _, left, right, hdr, vec = self.nodes[node]
        assert vec is not None, "normal: None needs the task rng (not exercised by the replayed snapshots)"
        sides = be.split_sides(vec, hdr, to_insert)
        left_ids = [i for i, s in zip(to_insert, sides) if s == 0]
        right_ids = [i for i, s in zip(to_insert, sides) if s == 1]
        if left_ids:
            self._route_tmp(be, left, left_ids, local)
        if right_ids:
            self._route_tmp(be, right, right_ids, local)

    # ---- make_tree_in_file, src/writer.rs:1167-1261, on any backend: [(id, node)] in creation order ----
    def _subtree(self, be, ids, rng, root_id, alloc, split_after):
        out = []

        def next_id():
            if alloc["pos"] < len(alloc["avail"]):
                alloc["pos"] += 1
                return alloc["avail"][alloc["pos"] - 1]
            alloc["current"] += 1
            return alloc["current"] - 1

        def rec(ids, node_id):
            if len(ids) <= split_after:
                nid = next_id() if node_id is None else node_id
                out.append((nid, ("D", [int(i) for i in ids])))
                return nid
            remaining = 3
            while True:
                a, b = rng.index_sample2(len(ids))  # choose_two, src/parallel.rs:342-355
                sample = [ids[a], ids[b]] + [ids[rng.gen_range_inclusive(0, len(ids) - 1)] for _ in range(10)]  # :358-367
                vec, hdr = be.create_split(sample)
                sides = np.asarray(be.split_sides(vec, hdr, ids))
                n_left = int((sides == 0).sum())
                imb = O.lib().ao_split_imbalance(n_left, len(ids) - n_left)
                if imb < 0.95 or remaining == 0:
                    break
                remaining -= 1
            if imb > 0.99:  # randomly_split_children: true -> Left (src/lib.rs:135-141)
                sides = np.array([0 if rng.gen_bool() else 1 for _ in ids], dtype=np.uint8)
                vec = None
            left = rec([i for i, s in zip(ids, sides) if s == 0], None)
            right = rec([i for i, s in zip(ids, sides) if s == 1], None)
            nid = next_id() if node_id is None else node_id  # allocated AFTER the children (:1257)
            out.append((nid, ("S", left, right, np.asarray(hdr, dtype=np.float32).copy(), vec)))
            return nid

        rec(list(ids), root_id)
        return out

    def _subtree_oracle_c(self, be, ids, rng, root_id, alloc, split_after):
        """The same through the oracle's C restatement (ao_ref_subtree): identical output, much faster."""
        L = O.lib()
        rows = np.array([be.row_of[i] for i in ids], dtype=np.uint32)
        avail_a = np.array(alloc["avail"] if alloc["avail"] else [0], dtype=np.uint32)
        pos, cur = C.c_uint32(alloc["pos"]), C.c_uint32(alloc["current"])
        L.ao_ref_subtree.restype = C.c_void_p
        L.ao_ref_subtree.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p,
                                     C.c_uint32, C.c_void_p, C.c_void_p]
        h = C.c_void_p(L.ao_ref_subtree(be.data.c(), split_after, rows.ctypes.data, len(rows), C.cast(rng._st, C.c_void_p),
                                        root_id, avail_a.ctypes.data, len(alloc["avail"]), C.byref(pos), C.byref(cur)))
        alloc["pos"], alloc["current"] = pos.value, cur.value
        nodes_p, normals_p, desc_p = C.POINTER(O.AoRefNode)(), C.POINTER(C.c_uint8)(), C.POINTER(C.c_uint32)()
        n = L.ao_ref_tree_nodes(h, C.byref(nodes_p), C.byref(normals_p), C.byref(desc_p))
        hs, vs = 4 * O.header_floats(self.metric), O.vector_bytes(self.metric, self.dims)
        out = []
        for i in range(n):
            nd = nodes_p[i]
            if nd.kind == 1:
                out.append((int(nd.id), ("D", [int(desc_p[nd.offset + j]) for j in range(nd.count)])))
            else:
                raw = bytes(C.string_at(C.addressof(normals_p.contents) + nd.offset, hs + vs))
                out.append((int(nd.id), ("S", int(nd.left), int(nd.right), np.frombuffer(raw[:hs], dtype=np.float32).copy(),
                                         raw[hs:] if nd.has_normal else None)))
        L.ao_ref_tree_free(h)
        return out

    # ---- incremental_index_large_descendant, src/writer.rs:660-739 ----
    def _run_task(self, be, seed, node, items, memory, split_after, alloc, stack):
        rng = O.ChaCha12(seed)
        to_insert = list(items)
        first = self._fit_in_memory(memory, to_insert, rng)
        build = self._subtree_oracle_c if self.backend_factory is None else self._subtree
        local, order = {}, IntMapOrder()  # the task's `descendants`: leaves in the order make_tree_in_file meets them
        for nid, nd in build(be, first, rng, node, alloc, split_after):
            if nd[0] == "D":
                local[nid] = nd[1]
                order.insert(nid)
                self.nodes.pop(nid, None)
            else:
                self.nodes[nid] = nd
        while True:
            batch = self._fit_in_memory(memory, to_insert, rng)
            if batch is None:
                break
            self._route_tmp(be, node, batch, local)
        self._walk(rng, order.order(), local, stack, split_after)

    # ---- Writer::build, src/writer.rs:487-629 ----
    def build(self, rng: "O.ChaCha12", n_trees=None, split_after=None, available_memory=None):
        split_after = split_after or self.dims
        memory = (1 << 64) - 1 if available_memory is None else available_memory
        item_ids = sorted(self.items)
        updated, self.updated = set(self.updated), set()
        if len(item_ids) <= split_after:  # clear_db_and_create_a_single_leaf, :916-962
            self.nodes = {0: ("D", item_ids)} if item_ids else {}
            self.roots = [0] if item_ids else []
            return
        to_delete = updated
        to_insert = [i for i in item_ids if i in updated]
        roots = list(self.roots) if self.roots is not None else []
        # ConcurrentNodeIds::new(used_tree_node): computed BEFORE anything is deleted (:516-518)
        used = sorted(self.nodes)
        last_id = used[-1] + 1 if used else 0
        alloc = {"avail": [i for i in range(last_id) if i not in self.nodes], "pos": 0, "current": last_id}
        want = target_n_trees(n_trees, self.dims, len(item_ids), len(roots))
        for _ in range(max(0, len(roots) - want)):  # delete_extra_trees, :631-655: the oldest first, swap_remove(0)
            if not roots:
                break
            root = roots[0]
            roots[0] = roots[-1]
            roots.pop()
            self._delete_tree(root)
        for i, root in enumerate(roots):  # delete_items_from_trees, :979-1017
            roots[i], _ = self._delete_items(root, to_delete, split_after)
        roots.sort()
        vecs = np.stack([self.items[i] for i in item_ids])
        be = OracleBackend(self.metric, item_ids, vecs) if self.backend_factory is None else self.backend_factory(item_ids, vecs)
        descendants, walk = {}, IntMapOrder()
        pending = list(to_insert)
        while roots:  # insert_items_in_current_trees, :846-889 (returns early without trees)
            batch = self._fit_in_memory(memory, pending, rng)
            if batch is None:
                break
            rng.next_u32()  # insert_items_in_tree, :1118-1160: `rng.next_u64()` seeds the per-tree rngs (only used
            rng.next_u32()  # below `normal: None` nodes)
            per_root = []
            for root in roots:
                touched = {}  # python dicts keep insertion order = the depth-first order of the routing
                self._route(be, root, batch, touched)
                m = IntMapOrder()
                for k in touched:
                    m.insert(k)
                per_root.append((m, touched))

            def fold(parts):  # rayon `reduce` on one thread: sequential fold of a half into a fresh map
                acc = IntMapOrder()
                for m, _ in parts:
                    for k in m.order():
                        acc.insert(k)
                return acc

            if len(per_root) >= 2:  # one split at len / 2 (Splitter with one thread), then `op(left, right)`
                left, right = fold(per_root[: len(per_root) // 2]), fold(per_root[len(per_root) // 2:])
                for k in right.order():
                    left.insert(k)
                acc = left
            else:
                acc = fold(per_root)
            merged = {}
            for _, touched in per_root:
                for k, v in touched.items():
                    merged[k] = sorted(set(merged.get(k, [])) | set(v))
            for k in acc.order():  # `descendants.entry(item_id).or_default().extend(desc)`, :880-882
                walk.insert(k)
                descendants[k] = sorted(set(descendants.get(k, [])) | set(merged[k]))

        def next_id():
            if alloc["pos"] < len(alloc["avail"]):
                alloc["pos"] += 1
                return alloc["avail"][alloc["pos"] - 1]
            alloc["current"] += 1
            return alloc["current"] - 1

        for _ in range(max(0, want - len(roots))):  # :556-561
            new_id = next_id()
            roots.append(new_id)
            descendants[new_id] = list(item_ids)
            walk.insert(new_id)
        order = walk.order()  # `for (item_id, item_indices) in descendants.into_iter()`, :778
        assert sorted(order) == sorted(descendants)
        # the walking task gets `StdRng::from_seed(rng.gen())` (:575); large entries become tasks with their own seed
        # (:795); on one rayon thread tasks run last-in-first-out, nested spawns included
        stack = []
        self._walk(O.ChaCha12(rng.gen_seed()), order, descendants, stack, split_after)
        task_memory = memory  # available_memory / current_num_threads() with one thread (:685-686)
        while stack:
            seed, node, items = stack.pop()
            self._run_task(be, seed, node, items, task_memory, split_after, alloc, stack)
        self.roots = roots

    def dump(self):
        """The database dump of the reference's test handle, in the shape of the parsed golden snapshots."""
        trees = {}
        for k, nd in self.nodes.items():
            if nd[0] == "D":
                trees[str(k)] = {"kind": "D", "descendants": list(nd[1])}
            else:
                vec = None if nd[4] is None else ["%.4f" % x for x in np.frombuffer(nd[4], dtype=np.float32)]
                trees[str(k)] = {"kind": "S", "left": nd[1], "right": nd[2],
                                 "bias": None if nd[4] is None else "%.4f" % nd[3][0], "vector": vec}
        return {"roots": list(self.roots or []), "item_ids": sorted(self.items), "trees": trees}
