"""The N>1 path on CPU: world_size 2 over gloo (the GPU run uses the same code over RCCL).

Trees shard round-robin over ranks with no data-path collective; bench.py's barrier and
max-over-ranks timing are exercised end to end through torch.distributed.run with --dry-run."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def test_union_digest_is_independent_of_the_sharding():
    """bench.py folds {tree index: keyed digest} into one value in tree order: whoever built which tree, the value is the same."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    per_tree = {t: (t * 0x9E3779B97F4A7C15 + 12345) & 0xFFFFFFFFFFFFFFFF for t in range(100)}
    whole = bench.union_digest(per_tree)
    for world in (2, 4, 8):
        merged = {}
        for r in reversed(range(world)):  # any arrival order
            merged.update({t: per_tree[t] for t in range(r, 100, world)})
        assert bench.union_digest(merged) == whole
    per_tree[37] ^= 1
    assert bench.union_digest(per_tree) != whole


def test_tree_sharding_is_a_partition_and_seeds_do_not_depend_on_world_size():
    from arroy_amd import shard
    for n_trees in (0, 1, 7, 50, 100):
        for world in (1, 2, 4, 8):
            parts = [shard.trees_for_rank(n_trees, r, world) for r in range(world)]
            flat = sorted(t for p in parts for t in p)
            assert flat == list(range(n_trees))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
            # the seed of tree t is the same whoever builds it => same forest at 1 and 8 GPUs
            all_seeds = shard.tree_seeds(42, range(n_trees))
            for p in parts:
                assert shard.tree_seeds(42, p) == [all_seeds[t] for t in p]
    seeds = shard.tree_seeds(42, range(100))
    assert len(set(seeds)) == 100 and all(0 <= s < 2**64 for s in seeds)
    # 100 trees on 8 GPUs: 13/12 split -> upper bound of the speed-up is 100/13 = 7.7x
    assert max(len(shard.trees_for_rank(100, r, 8)) for r in range(8)) == 13


def test_bench_two_ranks_over_gloo_dry_run():
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29571", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1", "--dry-run", "--trees", "5"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak"
    assert j["unit"] == "distances/s" and j["higher_is_better"] is True and j["vs_baseline"] is None
    # max over ranks: rank 1 sleeps 20 ms, rank 0 10 ms
    assert j["ms_per_step"] * 3 >= 19.0
    assert j["build"]["trees"] == 5 and j["build"]["trees_this_rank"] == 3
    # the gather that carries the per-device figures of an N > 1 line (build_10m_per_device, the union of the shares' digests)
    assert j["build"]["per_rank"] == [{"rank": 0, "trees": 3}, {"rank": 1, "trees": 2}]
    assert set(j["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}


def _run_bench(argv, env_extra=None):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, timeout=300,
                          env=env, cwd=ROOT)


def test_bench_gpus_n_started_directly_drives_n_devices_from_one_process_dry_run():
    """`python bench.py --gpus 2` without a launcher: one process, one host thread per device (the shape of arroy's own
    single-process build, src/writer.rs:556-591).  --dry-run exercises the control path on CPU."""
    out = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run", "--trees", "5"])
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and "2 host threads" in j["config"]["launch"]
    assert j["ms_per_step"] * 3 >= 19.0  # max over the device threads: thread 1 sleeps 20 ms
    assert j["build"]["trees"] == 5 and j["build"]["trees_this_rank"] == 3
    assert j["build"]["per_rank"] == [{"rank": 0, "trees": 3}, {"rank": 1, "trees": 2}]


def test_bench_refuses_to_run_on_fewer_devices_than_asked():
    """--gpus N must never silently become a 1-GPU run: fewer visible devices is exit code 3, a launcher world that
    disagrees with --gpus is exit code 2."""
    import arroy_amd
    if arroy_amd.device_count() < 8:
        out = _run_bench(["--gpus", "8", "--steps", "1", "--warmup", "0", "--no-cpu"])
        assert out.returncode == 3 and "device(s) visible" in out.stderr
    out = _run_bench(["--gpus", "2", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode == 2 and "WORLD_SIZE" in out.stderr
