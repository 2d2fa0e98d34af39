"""One process, several devices / several concurrent builds: arroy's own shape (`Writer::build` is one process whose tasks
share ONE ImmutableLeafs, src/writer.rs:530,556-591).  Device d holds a replica and builds the trees t = d (mod N) from its
own host thread; the forests must be the ones a single device builds, and the caller's current device is never changed.
The two-device test skips on a one-GPU box (the driver's 8-GPU node runs it); the concurrency test runs everywhere."""
import ctypes as C
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from arroy_amd import _lib  # noqa: E402
from arroy_amd import distances as D  # noqa: E402


def current_device() -> int:
    hip = C.CDLL("libamdhip64.so")
    d = C.c_int(-1)
    assert hip.hipGetDevice(C.byref(d)) == 0
    return d.value


def make(n=200_000, dims=128, device=0):
    from arroy_amd import Dataset
    ds = Dataset(D.Euclidean, dims, n, device=device)
    ds.fill_synthetic(7, 2, n)
    ds.finalize()
    return ds


def build_sharded(replicas, seeds, max_host_threads):
    """Tree t on replica t mod N, one host thread per replica; returns {tree index: canonical tuple}."""
    from arroy_amd import shard
    world = len(replicas)
    out, errors = {}, []

    def work(rank):
        try:
            mine = shard.trees_for_rank(len(seeds), rank, world)
            f = replicas[rank].build_forest([seeds[t] for t in mine], max_host_threads=max_host_threads)
            for local, t in enumerate(mine):
                out[t] = f.canonical(local)
            f.close()
        except BaseException as e:  # noqa: BLE001
            errors.append(e)
    ths = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errors, errors
    return out


def test_concurrent_builds_in_one_process_equal_the_sequential_forest():
    """Four host threads, four replicas on ONE device, two host threads of budget each: the forest of the single call."""
    from arroy_amd import shard
    ds = make()
    seeds = shard.tree_seeds(42, range(10))
    whole = ds.build_forest(seeds)
    want = {t: whole.canonical(t) for t in range(len(seeds))}
    whole.close()
    before = current_device()
    replicas = [ds] + [ds.replicate(0) for _ in range(3)]
    assert current_device() == before
    got = build_sharded(replicas, seeds, max_host_threads=2)
    assert got == want
    assert current_device() == before
    # the same with every build's tail in groups of trees (each shard has 2-3 trees): the pooled node records, the ring of node
    # tables and the groups' hand-overs of four builds at once
    with _lib.tuning(AH_BUILD_TAIL_GROUPS=2, AH_BUILD_TAIL_MIN_MB=0, AH_ROWMAJOR=0):
        for _rep in range(3):
            assert build_sharded(replicas, seeds, max_host_threads=2) == want
    for r in replicas[1:]:
        r.close()
    ds.close()


@pytest.mark.parametrize("sparse", [False, True])
def test_replicate_through_pinned_host_memory_when_there_is_no_peer_access(sparse):
    """ABI v7: ah_dataset_replicate checks every array it copied (head and tail read back from both devices) and falls back to
    a copy through pinned host memory when the destination cannot address the source (or the peer copy fails / delivers other
    bytes).  AH_REPLICATE_HOST_BOUNCE=1 forces that path where peer access works: the replica must be the dataset — ids,
    headers, rows, the id lookup table of a sparse id set — bit for bit, and build the same forest."""
    from arroy_amd import Dataset, shard
    n, dims = 150_000, 100
    ds = Dataset(D.Cosine, dims, n)
    rng = np.random.default_rng(3)
    vecs = rng.standard_normal((n, dims)).astype(np.float32)
    ids = np.sort(rng.choice(8 * n, n, replace=False)).astype(np.uint32) if sparse else np.arange(n, dtype=np.uint32)
    ds.upload_vectors(ids, vecs)
    ds.finalize()
    seeds = shard.tree_seeds(5, range(3))
    want = ds.build_forest(seeds)
    with _lib.tuning(AH_REPLICATE_HOST_BOUNCE=1):
        rep = ds.replicate(0)
    assert len(rep) == n and rep.read_headers().tobytes() == ds.read_headers().tobytes()
    for i in (0, 1, n // 2, n - 1):
        assert rep.item_vector(int(ids[i])).tobytes() == vecs[i].tobytes()
    got = rep.build_forest(seeds)
    assert got.digest()[0] == want.digest()[0]
    q = vecs[17]
    a, b = ds.rerank(20, query=q), rep.rerank(20, query=q)
    assert list(a[0]) == list(b[0]) and a[1].tobytes() == b[1].tobytes()
    for x in (got, want):
        x.close()
    rep.close()
    ds.close()


def test_two_devices_build_disjoint_tree_sets_equal_to_the_one_device_forest():
    import arroy_amd
    if arroy_amd.device_count() < 2:
        pytest.skip("needs two visible devices")
    from arroy_amd import shard
    n_dev = min(arroy_amd.device_count(), 8)
    ds = make()
    seeds = shard.tree_seeds(42, range(2 * n_dev + 1))
    whole = ds.build_forest(seeds)
    want = {t: whole.canonical(t) for t in range(len(seeds))}
    whole.close()
    before = current_device()
    replicas = [ds] + [ds.replicate(d) for d in range(1, n_dev)]
    assert current_device() == before, "ah_dataset_replicate must put the caller's device back"
    # the replica answers queries with the source's bits
    q = np.linspace(-1, 1, 128, dtype=np.float32)
    assert replicas[-1].distances(query=q).tobytes() == ds.distances(query=q).tobytes()
    got = build_sharded(replicas, seeds, max_host_threads=2)
    assert got == want
    assert current_device() == before
    # a search on the last device's replica, through an index mirrored there
    f = replicas[-1].build_forest(seeds[:3])
    ix = replicas[-1].create_index(f)
    f0 = ds.build_forest(seeds[:3])
    ix0 = ds.create_index(f0)
    qs = np.stack([q, -q])
    a, b = ix.search(10, queries=qs, raw=True), ix0.search(10, queries=qs, raw=True)
    assert all(np.array_equal(x.view(np.uint32), y.view(np.uint32)) for x, y in zip(a, b))
    for h in (ix, ix0, f, f0):
        h.close()
    for r in replicas[1:]:
        r.close()
    ds.close()


def test_reserved_and_recycled_memory_never_changes_a_forest():
    """The memory side of a build — ah_dataset_reserve_build (device memory obtained on a helper thread while records are
    staged), the caching device allocator, the recycled host blobs — must be invisible in the result: same digest with fresh
    memory, with recycled memory, after the caches were trimmed, and with the caches switched off."""
    from arroy_amd import Dataset, shard
    n, dims = 400_000, 96  # (12 trees x 400 000 ids = 19 MB: both output blobs are above the pool's 16 MB floor)
    seeds = shard.tree_seeds(7, range(12))

    def fresh(reserve):
        ds = Dataset(D.Cosine, dims, n)
        if reserve:
            ds.reserve_build(len(seeds))
            ds.reserve_build(len(seeds))  # a second call joins the first helper
        ds.fill_synthetic(3, 2, n)
        ds.finalize()
        return ds
    ds = fresh(False)
    f = ds.build_forest(seeds)
    want = f.digest()[0]
    assert f.stats["host_blob_recycled"] in (0, 1, 2)
    f.close()
    g = ds.build_forest(seeds)  # blobs and device scratch recycled from the build before
    assert g.digest()[0] == want and g.stats["host_blob_recycled"] == 2, g.stats
    g.close()
    assert _lib.host_cache_trim() > 0 and _lib.device_cache_trim() > 0
    h = ds.build_forest(seeds)  # ... and fresh again
    assert h.digest()[0] == want and h.stats["host_blob_recycled"] == 0
    h.close()
    with _lib.tuning(AH_DEVICE_CACHE_MB=0, AH_HOST_CACHE_MB=0):
        k = ds.build_forest(seeds)
        assert k.digest()[0] == want
        k.close()
    ds.close()
    dr = fresh(True)
    r = dr.build_forest(seeds)
    assert r.digest()[0] == want
    r.close()
    dr.close()
    unused = Dataset(D.Cosine, dims, n)
    unused.reserve_build(1000)  # destroyed while (or right after) the helper runs
    unused.close()


# ---- bench.py --gpus N on ONE GPU (--virtual): the un-dry-run N > 1 path end to end -------------------------------------------

def _bench(argv, launcher=None, timeout=900):
    import os
    import subprocess
    import sys

    from conftest import ROOT
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    cmd = ([sys.executable] + launcher if launcher else [sys.executable]) + [os.path.join(ROOT, "bench.py")] + argv
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def _check_virtual_line(out, world):
    import json
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    j = json.loads(lines[0])
    assert j["n_gpus"] == world and "virtual_devices" in j["config"]
    u = j["build_10m_union"]
    # every tree of the 100 was built by exactly one rank, and tree by tree the shares equal rank 0's own 100-tree build
    assert u["trees"] == 100 and u["complete"] and u["checked_against_one_device_build"] and u["identical"], u
    per = j["build_10m_per_device"]
    assert sorted(int(k) for k in per) == list(range(world))
    for r, d in per.items():
        assert d["trees"] in (100 // world, 100 // world + 1) and d["seconds"] > 0 and d["seconds_device"] > 0, (r, d)
        assert d["seconds_setup"] is not None and d["seconds_after_device"] is not None and d["max_host_threads"] >= 1
    same = j["build_10m_identical_per_device"]
    assert all(same.values()) and len([k for k in same if k != "normal"]) == world  # screened == f32-only on every rank
    assert j["build"]["trees_this_rank"] == len(range(0, 50, world))
    return j


def test_bench_four_virtual_devices_on_one_gpu_threads():
    """`python bench.py --gpus 4 --virtual`: one process, four device threads, the 1M x 768 dataset staged once and replicated
    (ah_dataset_replicate onto the same device), every thread builds its trees t = rank (mod 4); the union of the shares'
    per-tree digests must equal a one-device build of all 100 trees (bench.py exits 6 otherwise)."""
    out = _bench(["--gpus", "4", "--virtual", "--steps", "3", "--warmup", "1", "--no-cpu", "--no-extra", "--no-live-pmc"])
    j = _check_virtual_line(out, 4)
    assert j["replicate_10m"]["replicas"] == 3 and j["replicate_10m"]["gb_per_s_total"] > 0
    assert "4 host threads" in j["config"]["launch"]


def test_bench_two_virtual_devices_one_process_per_rank():
    """The launch the driver uses for N > 1 (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`), two ranks
    on the one GPU: each rank fills its own dataset, builds its share, rank 0 gathers the per-device figures and checks the
    union (gloo for the control path: RCCL wants one rank per device)."""
    out = _bench(["--gpus", "2", "--virtual-devices", "--steps", "3", "--warmup", "1", "--no-cpu", "--no-extra", "--no-live-pmc"],
                 launcher=["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                           "--master-port", "29577"])
    j = _check_virtual_line(out, 2)
    assert "one process per GPU" in j["config"]["launch"]
