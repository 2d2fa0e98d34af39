"""GPU tests of the staging side of the boundary (`ImmutableLeafs::new`, src/parallel.rs:271-293): records staged from the
raw, arbitrarily misaligned pointers a reader gets from LMDB; asynchronous uploads; device-to-device replicas; the
typed error details; caller-owned forest views."""
import ctypes as C
import mmap
import os
import struct

import numpy as np
import pytest

import test_gpu_parity as P
from conftest import ROOT, hex_f32
from oracle import oracle as O
from test_gpu_parity import assert_bit_equal, make_data

pytestmark = pytest.mark.gpu

from arroy_amd import _lib  # noqa: E402
from arroy_amd import distances as D  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def _imports():
    import arroy_amd
    assert arroy_amd.device_count() >= 1, "no GPU visible: these tests must run on an MI355X"
    P.D, P.O = D, O


def lmdb_item_offsets(data):
    """(item id, file offset of the value, value length) of every arroy item record of an LMDB data file, following the
    page layout of SURVEY.md Appendix B (the same walk tests/golden/make_golden.py does; offsets instead of copies)."""
    def meta(pg_off):
        off = pg_off + 16
        magic, _version = struct.unpack_from("<II", data, off)
        assert magic == 0xBEEFC0DE
        off += 8 + 8 + 8
        dbs = []
        for _ in range(2):
            pad, _flags, _depth, _b, _l, _o, entries, root = struct.unpack_from("<IHHQQQQQ", data, off)
            dbs.append((pad, entries, root))
            off += 48
        _last, txnid = struct.unpack_from("<QQ", data, off)
        return dbs, txnid
    psize = meta(0)[0][0][0]
    dbs, _ = max([meta(0), meta(psize)], key=lambda m: m[1])
    out = []

    def walk(pgno):
        base = pgno * psize
        _pg, _pad, flags, lower, _upper = struct.unpack_from("<QHHHH", data, base)
        for o in struct.unpack_from("<%dH" % ((lower - 16) // 2), data, base + 16):
            lo, hi, nflags, ksize = struct.unpack_from("<HHHH", data, base + o)
            key = data[base + o + 8: base + o + 8 + ksize]
            if flags & 0x01:
                walk(lo | (hi << 16) | (nflags << 32))
                continue
            dsize = lo | (hi << 16)
            if nflags & 0x01:  # F_BIGDATA: the value lives on an overflow page, 16 bytes in
                (ov,) = struct.unpack_from("<Q", data, base + o + 8 + ksize)
                voff = ov * psize + 16
            else:
                voff = base + o + 8 + ksize
            _index, mode, item = struct.unpack(">HBI", key[:7])
            if mode == 3:  # Key::item (src/key.rs:56-71)
                out.append((item, voff, dsize))
    walk(dbs[1][2])
    return sorted(out)


def test_records_staged_from_pointers_into_a_mapped_lmdb_file(golden):
    """The reference's own database file (src/tests/assets/v0_6/large.mdb, copied to tests/golden/) mapped into memory:
    ah_dataset_upload_records receives the addresses of the item values inside the pages — what `ImmutableLeafs::new`
    collects (src/parallel.rs:271-293) — at whatever alignment LMDB left them.  Items and the golden nearest neighbours
    of src/tests/upgrade.rs:116-128 must come out bit for bit."""
    from arroy_amd import Dataset
    g = golden["large_v0_6"]
    path = os.path.join(ROOT, "tests", "golden", "large_v0_6.mdb")
    with open(path, "rb") as f:
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_COPY)  # private mapping: from_buffer needs a writable view
    data = bytes(mm)
    items = lmdb_item_offsets(data)
    assert [i for i, _, _ in items] == g["ids"] and len(items) == 100
    rec_len = 1 + 4 + 4 * g["dims"]
    assert all(l == rec_len for _, _, l in items)
    base = C.addressof(C.c_char.from_buffer(mm))
    addrs = [base + off for _, off, _ in items]
    assert any((a + 5) % 4 != 0 for a in addrs), "the fixture is expected to hold misaligned vectors"
    ds = Dataset(D.Euclidean, g["dims"], len(items))
    ds.upload_record_pointers(g["ids"], addrs, rec_len)
    ds.finalize()
    for k in (0, 17, 99):
        assert_bit_equal(ds.item_vector(g["ids"][k]), hex_f32(g["vectors_hex"][k]))
    assert_bit_equal(ds.read_headers().ravel(), np.concatenate([hex_f32(h) for h in g["headers_hex"]]))
    ids, dists = ds.rerank(g["count"], query=np.array(g["query"], dtype=np.float32))
    from test_oracle_golden import rust_display_f32
    assert [[int(i), rust_display_f32(d)] for i, d in zip(ids, dists)] == g["expected"]
    ds.close()
    del base, addrs
    mm.close()


def test_shim_roundtrip_in_plain_c_on_the_reference_database(tmp_path):
    """examples/shim_roundtrip.c end to end: records staged from pointers into the mapped large.mdb, forest built, every
    node passed through the sink and ENCODED in the NodeCodec v0.7 layout (src/node.rs:224-241: `[2u8][left BE][right BE]
    [header][vector]`, `[1u8][roaring]`), decoded again (src/node.rs:252-273), mirrored on the device from the decoded
    arrays and searched: the golden neighbours of src/tests/upgrade.rs:116-128."""
    import subprocess

    from test_abi import build_c_example
    exe = build_c_example("shim_roundtrip", tmp_path)
    out = subprocess.run([str(exe), os.path.join(ROOT, "tests", "golden", "large_v0_6.mdb")], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "id(92): distance(2.4881108)" in out.stdout and out.stdout.strip().endswith("ok")


def test_shim_incremental_in_plain_c_equals_the_oracles_replay(tmp_path, golden):
    """examples/shim_incremental.c: the INCREMENTAL half of the Rust shim (integration/arroy-hip: `route_into_current_trees`,
    `build_large_descendants`) through the C ABI in the patch's call order — index over items 0..89 of the reference's
    large.mdb, ten items added, routed down the existing trees on a dataset that holds ONLY the new items and a tree mirror
    WITHOUT item lists, the outgrown descendants re-split in one ah_build_subtrees call over ONLY their members, ids handed out
    like `make_tree_in_file` does.  The updated forest must equal, node for node, what the CPU oracle makes of the same steps
    (src/writer.rs:846-889,1398-1459 `ao_route_items`; :660-739,1167-1261 `ao_build_tree_on`)."""
    import subprocess

    from oracle import oracle as O
    from test_abi import build_c_example
    exe = build_c_example("shim_incremental", tmp_path)
    dump = tmp_path / "forest.bin"
    mdb = os.path.join(ROOT, "tests", "golden", "large_v0_6.mdb")
    out = subprocess.run([str(exe), mdb, str(dump)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert out.stdout.strip().endswith("ok") and "outgrew split_after" in out.stdout
    # ---- the C side's updated forest
    raw = dump.read_bytes()
    n_trees, n_nodes, normals_len, desc_len, stride, vec_off, hdr_off = np.frombuffer(raw, "<u8", 7)
    node_dt = np.dtype([("kind", "u1"), ("has_normal", "u1"), ("reserved", "<u2"), ("tree", "<u4"), ("left", "<u4"), ("right", "<u4"),
                        ("offset", "<u8"), ("count", "<u4"), ("depth", "<u4")], align=True)
    at = 56
    roots = np.frombuffer(raw, "<u4", int(n_trees), at); at += 4 * int(n_trees)
    nodes = np.frombuffer(raw, node_dt, int(n_nodes), at); at += node_dt.itemsize * int(n_nodes)
    normals = np.frombuffer(raw, "u1", int(normals_len), at); at += int(normals_len)
    desc = np.frombuffer(raw, "<u4", int(desc_len), at)
    dims, hs = 30, 4

    def canonical_c(i):
        nd = nodes[i]
        if nd["kind"] == 1:
            return ("D", tuple(int(x) for x in desc[int(nd["offset"]): int(nd["offset"]) + int(nd["count"])]))
        rec = normals[int(nd["offset"]): int(nd["offset"]) + int(stride)]
        nb = rec[int(hdr_off): int(hdr_off) + hs].tobytes() + rec[int(vec_off): int(vec_off) + 4 * dims].tobytes() if nd["has_normal"] else None
        return ("S", nb, canonical_c(int(nd["left"])), canonical_c(int(nd["right"])))
    # ---- the oracle's replay of the same steps
    g = golden["large_v0_6"]
    assert g["ids"] == list(range(100))
    vecs = np.stack([hex_f32(h) for h in g["vectors_hex"]])
    od = O.Data(O.EUCLIDEAN, vecs)
    new = np.arange(90, 100, dtype=np.uint32)
    resplit = 0
    for t in range(int(n_trees)):
        tree = od.build_tree(8, 42 + t, rows=np.arange(90, dtype=np.uint32))
        assert tree.dummy_normals == 0
        f = tree.as_forest(od)
        leaf_of = O.route_items(od, f, new, [0x5EED + int(roots[t])])[0]
        grown = {}
        for item, leaf in zip(new, leaf_of):
            grown.setdefault(int(leaf), []).append(int(item))

        def canonical_o(i):
            kind, has_normal, left, right, offset, count, _depth = tree.nodes[i]
            if kind == 1:
                stored = [int(x) for x in tree.descendants[offset:offset + count]]
                if i not in grown:
                    return ("D", tuple(stored))
                merged = sorted(stored + grown[i])
                if len(merged) <= 8:
                    return ("D", tuple(merged))
                nonlocal_resplit.append(1)
                return od.build_tree(8, 1000 + 1000 * t + merged[0], rows=np.array(merged, dtype=np.uint32)).canonical()
            nb = tree.normals[offset:offset + tree.stride] if has_normal else None
            return ("S", nb, canonical_o(left), canonical_o(right))
        nonlocal_resplit = []
        assert canonical_c(int(roots[t])) == canonical_o(tree.root), f"tree {t} of the updated forest differs from the oracle's replay"
        resplit += len(nonlocal_resplit)
    assert resplit >= 1


def test_c_abi_demo_runs_to_the_end(tmp_path):
    """examples/c_abi_demo.c from plain C99: build, search (the item itself first), the node sink after the build and the
    batch sink DURING the build — the streamed forest has the node and item counts of the materialised one."""
    import subprocess

    from test_abi import build_c_example
    exe = build_c_example("c_abi_demo", tmp_path)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "streamed:" in out.stdout and "holding 80000 items" in out.stdout and out.stdout.strip().endswith("ok")


def test_uploads_are_asynchronous_but_never_read_the_callers_memory_after_return():
    """The staging contract (include/arroy_hip.h): the pointers are not used after return, although the DMA of the last
    chunks may still be in flight.  Overwrite the source right after every call; the dataset must hold the originals."""
    from arroy_amd import Dataset
    n, dims = 60_000, 768  # 184 MB: several 32 MiB chunks per call
    rng = np.random.default_rng(3)
    vecs = rng.standard_normal((n, dims)).astype(np.float32)
    keep = vecs.copy()
    ds = Dataset(D.Cosine, dims, n)
    half = n // 2
    a = vecs[:half].copy()
    ds.upload_vectors(np.arange(half, dtype=np.uint32), a)
    a[:] = np.nan
    b = vecs[half:].copy()
    ds.upload_vectors(np.arange(half, n, dtype=np.uint32), b)
    b[:] = np.nan
    ds.finalize()
    for i in (0, half - 1, half, n - 1, 12345):
        assert_bit_equal(ds.item_vector(i), keep[i])
    od = O.Data(O.COSINE, keep)
    q, qh = od.item_leaf(7)
    assert_bit_equal(ds.distances(item=7), od.distances(q, qh))
    ds.close()


def test_replica_is_identical_and_independent():
    """ah_dataset_replicate: a device-to-device copy of a finalized dataset (the one-tree-batch-per-GPU build stages once
    and replicates).  On a one-GPU box the replica lives on the same device; it must answer and build exactly like the
    source, and outlive it."""
    ids = np.sort(np.random.default_rng(2).choice(50_000, 4000, replace=False)).astype(np.uint32)
    ds, oracle, vecs, ids = make_data(D.DotProduct, 4000, 96, seed=21, ids=ids)
    rep = ds.replicate(0)
    assert len(rep) == len(ds)
    q = vecs[11] * np.float32(1.01)
    assert_bit_equal(rep.distances(query=q), ds.distances(query=q))
    f_src = ds.build_forest([7, 8, 9], split_after=40)
    ds.close()  # the replica owns its memory
    f_rep = rep.build_forest([7, 8, 9], split_after=40)
    for t in range(3):
        assert f_rep.canonical(t) == f_src.canonical(t) == oracle.build_tree(40, [7, 8, 9][t]).canonical()
    rep.close()


def test_error_details_carry_the_typed_fields_of_arroy_errors():
    """ah_last_error_detail: Error::InvalidVecDimension { expected, received } (src/error.rs:17-23) and
    Error::MissingKey { item } (:58-67) can be rebuilt by the caller instead of a string."""
    from arroy_amd import Dataset
    ds = Dataset(D.Euclidean, 32, 4)
    with pytest.raises(_lib.InvalidVecDimension) as e:
        ds.upload_records([0], [b"\x00" + b"\x00" * 4 + b"\x00" * 4 * 31])  # 31-dim record in a 32-dim index
    assert (e.value.expected, e.value.received) == (1 + 4 + 128, 1 + 4 + 124)
    ds.upload_vectors([3, 9], np.ones((2, 32), dtype=np.float32))
    ds.finalize()
    with pytest.raises(_lib.MissingKey) as e:
        ds.item_vector(5)
    assert e.value.item == 5
    ds.close()


@pytest.mark.parametrize("metric,dims", [(4, 64), (5, 64), (6, 64), (6, 960), (4, 130), (2, 40), (3, 70)])
def test_index_from_a_compact_caller_view_equals_the_forest_handle(metric, dims):
    """ah_index_create_from_view on records of stride header + vector (what a caller decoding LMDB tree nodes holds): the
    device pitch beyond the stored vector is zero-filled, never copied from the next record — for 1-bit metrics with an
    odd word count (64, 960 dims) a stray word would change every margin of the descent."""
    from arroy_amd import Index
    from arroy_amd.index import TreeStore
    cls = D.BY_METRIC[metric]
    n = 3000
    ds, oracle, vecs, ids = make_data(cls, n, dims, seed=metric * 100 + dims)
    seeds = [3, 4, 5]
    forest = ds.build_forest(seeds, split_after=24)
    by_handle = ds.create_index(forest)
    store = TreeStore()
    for t in range(forest.n_trees):
        store.roots.append(store.import_tree(forest, t))
    view, keep = store.to_view(cls, dims)
    assert view.normal_stride == cls.header_size() + cls.vector_size(dims)  # compact records
    by_view = Index(ds, None, view=view)
    queries = vecs[:40] + np.float32(0.001)
    for count, search_k in [(5, 0), (20, 300)]:
        a = by_handle.search(count, queries=queries, search_k=search_k)
        b = by_view.search(count, queries=queries, search_k=search_k)
        assert [[i for i, _ in r] for r in a] == [[i for i, _ in r] for r in b]
        for ra, rb in zip(a, b):
            assert_bit_equal([d for _, d in ra], [d for _, d in rb])
        qv, qh = oracle.query_leaf(queries[0])
        want, _ = O.search(oracle, forest, qv, qh, count, search_k, 0, None)
        assert [i for i, _ in a[0]] == [i for i, _ in want]
    new_ids = ids[:50]
    assert np.array_equal(by_view.route_items(new_ids, seeds)[0] >= 0, np.ones(50, bool))
    del keep


def test_caller_views_must_be_forests():
    """A view with a cycle or a shared sub-tree is rejected on the host (it would hang the descent / overflow the
    candidate buffer on the device), as is a record geometry that does not fit its stride."""
    from arroy_amd import Index
    from arroy_amd.index import TreeStore
    ds, oracle, vecs, ids = make_data(D.Euclidean, 500, 32, seed=4)
    forest = ds.build_forest([1], split_after=16)
    store = TreeStore()
    store.roots.append(store.import_tree(forest, 0))
    view, keep = store.to_view(D.Euclidean, 32)
    nodes = np.ctypeslib.as_array(C.cast(view.nodes, C.POINTER(C.c_uint8)), shape=(view.n_nodes * 32,)).view(
        np.dtype([("kind", "u1"), ("has_normal", "u1"), ("reserved", "<u2"), ("tree", "<u4"), ("left", "<u4"), ("right", "<u4"),
                  ("offset", "<u8"), ("count", "<u4"), ("depth", "<u4")]))
    root = int(np.ctypeslib.as_array(view.roots, shape=(1,))[0])
    split = next(i for i in range(len(nodes)) if nodes[i]["kind"] == 2 and i != root)
    saved = nodes[split].copy()
    nodes[split]["left"] = root  # a cycle
    with pytest.raises(_lib.ArroyHipError, match="reachable twice"):
        Index(ds, None, view=view)
    nodes[split] = saved
    nodes[split]["right"] = nodes[split]["left"]  # a shared sub-tree
    with pytest.raises(_lib.ArroyHipError, match="reachable twice"):
        Index(ds, None, view=view)
    nodes[split] = saved
    stride = view.normal_stride
    view.normal_stride = stride - 4  # the vector no longer fits the record
    with pytest.raises(_lib.ArroyHipError, match="do not fit"):
        Index(ds, None, view=view)
    view.normal_stride = stride
    Index(ds, None, view=view).close()  # restored: accepted
    del keep


def test_null_arguments_are_errors_not_crashes():
    from arroy_amd import Dataset
    L = _lib.lib()
    ds = Dataset(D.Euclidean, 8, 4)
    assert L.ah_dataset_upload_vectors(ds._h, None, None, 2) == 5  # AH_ERR_INVALID_ARGUMENT, not a segfault
    assert L.ah_dataset_upload_records(ds._h, None, None, 37, 2) == 5
    ds.close()


def test_build_is_cancelled_within_a_level():
    """The cancel closure is evaluated while the level's kernels run (the reference polls per node / per item,
    src/writer.rs:1178,1196): a closure that turns true in the middle of a build stops it within about one level's run
    time, whenever it fires — three trials at different moments of a 13-level build."""
    import threading
    import time

    from arroy_amd import BuildCancelled, Dataset
    n, dims = 4_000_000, 768
    ds = Dataset(D.Cosine, dims, n)
    ds.fill_synthetic(42, 1, n)
    ds.finalize()
    seeds = list(range(1, 49))
    mode = _lib.MARGIN_NODE_MAJOR | _lib.MARGIN_EXACT_ONLY  # every level costs the same: 48 x 4M rows of 3 KB, ~90 ms
    ds.build_forest(seeds[:2], margin_mode=mode).close()
    t0 = time.perf_counter()
    f = ds.build_forest(seeds, margin_mode=mode)
    full = time.perf_counter() - t0
    level_s = full / f.stats["levels"]
    f.close()
    for frac in (0.21, 0.37, 0.52):
        fired = {}
        go = threading.Event()

        def fire():
            fired["t"] = time.perf_counter()
            go.set()
        timer = threading.Timer(full * frac, fire)
        timer.start()
        with pytest.raises(BuildCancelled):
            ds.build_forest(seeds, margin_mode=mode, cancel=go.is_set)
        latency = time.perf_counter() - fired["t"]
        assert latency < 1.5 * level_s, f"cancel took {latency * 1e3:.1f} ms, a level runs {level_s * 1e3:.1f} ms"
    ds.close()


def test_search_accepts_counts_beyond_the_batched_top_k():
    """`Reader::nns(count)` takes any count (src/reader.rs:296-315); above 2048 the device search leaves the batched
    tournament top-k for the single-query kernels.  Same results as the oracle's `nns_by_leaf`, short lists padded."""
    ds, oracle, vecs, ids = make_data(D.Euclidean, 9000, 48, seed=8)
    forest = ds.build_forest([1, 2, 3], split_after=64)
    index = ds.create_index(forest)
    queries = vecs[:3] + np.float32(0.01)
    for count, search_k in [(3000, 5000), (2049, 2**62), (12000, 2**62)]:
        got = index.search(count, queries=queries, search_k=search_k)
        for qi in range(len(queries)):
            qv, qh = oracle.query_leaf(queries[qi])
            want, _ = O.search(oracle, forest, qv, qh, count, search_k, 0, None)
            assert [i for i, _ in got[qi]] == [i for i, _ in want]
            assert_bit_equal([d for _, d in got[qi]], [d for _, d in want])
        ids_raw, d_raw, counts = index.search(count, queries=queries, search_k=search_k, raw=True)
        for qi in range(len(queries)):
            assert np.all(ids_raw[qi, counts[qi]:] == 0xFFFFFFFF) and np.all(np.isnan(d_raw[qi, counts[qi]:]))
