#!/usr/bin/env python3
"""bench.py — the headline measurement of arroy's distance-kernel hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1]): 1M x 768-dim cosine, synthetic i.i.d. uniform[-1,1) vectors generated
in HBM by the counter-based generator of include/arroy_hip_policy.h (seed 42), ids 0..N-1.
  * A "step" = one Q=1 batched cosine-distance scan over all 1M resident items (one kernel launch):
    `D::built_distance(query, item)` for every item, src/reader.rs:381-391 -> src/distance/cosine.rs:43-59.
    Inputs are resident in HBM when the timed region starts; outputs stay in HBM.
  * value = total distances/s over all ranks (every rank scans its own replica: weak scaling).
  * roofline: algorithmic bytes per launch = 1M x (4*768 + 4 header + 4 out) = 3080 B/distance
     / the average kernel time measured with HIP events on the launch stream.
  * build: the n_trees=50 forest of the same config, trees sharded round-robin over ranks, no collective;
    build_10m: the 10M x 768, n_trees=100 forest of configs[2] ("tree-build seconds at 10M vectors"), same sharding.
  * cpu_baseline (rank 0, N=1 only): the C oracle (a restatement of arroy's AVX2+FMA path, NOT arroy) on a
    bounded sample of the same workload, all host cores.
One JSON line on stdout (rank 0).  `--dry-run` exercises the multi-process control path on CPU (gloo).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_ITEMS = 1_000_000
DIMS = 768
N_TREES = 50
SEED = 42
BYTES_PER_DISTANCE = 4 * DIMS + 4 + 4  # vector + stored norm + written distance
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--items", type=int, default=N_ITEMS)
    ap.add_argument("--trees", type=int, default=N_TREES)
    ap.add_argument("--no-build", action="store_true", help="skip the forest-build measurement")
    ap.add_argument("--no-build-10m", action="store_true",
                    help="skip the 10M x 768 x 100-tree build (BASELINE configs[2], the 'tree-build seconds at 10M' of the metric)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: test the N>1 control path (gloo)")
    ap.add_argument("--extra", default="", help="comma list of extra measurements: c3 (10M build), c4 (re-rank), "
                                               "c5 (1-bit scan), metrics (every f32 metric), search, staging; reported "
                                               "under the `extra` key")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per launch from a separate rocprofv3 --pmc pass (profiles/), if known")
    return ap.parse_args()


def cpu_baseline(args, n_items):
    """Oracle timed on the host cores: bounded sample of the same workload (scan + a slice of the build)."""
    import numpy as np

    from oracle import oracle as O
    L = O.lib()
    cores = L.ao_num_threads()
    n = min(n_items, 200_000)
    vecs = O.synth(SEED, 1, n, DIMS)
    big = O.Data(O.COSINE, vecs)  # headers = exact norms, computed by the oracle itself (parallel)
    import ctypes as C
    q, qh = big.item_leaf(0)
    out = np.zeros(n, dtype=np.float32)
    # warm-up + timed repetitions for ~cpu_seconds/2
    big.distances(q, qh)
    reps, t0 = 0, time.perf_counter()
    while True:
        L.ao_distances(big.c(), q.ctypes.data_as(C.c_void_p), qh.ctypes.data_as(C.c_void_p), None, n,
                       out.ctypes.data_as(C.c_void_p))
        reps += 1
        el = time.perf_counter() - t0
        if el > args.cpu_seconds * 0.5 or reps >= 200:
            break
    scan_rate = reps * n / el
    res = {"value": scan_rate, "unit": "distances/s", "cores": int(cores), "kind": "port",
           "sample": f"Q=1 cosine scan over {n}x{DIMS} resident rows x {reps} reps, OpenMP on {cores} threads; "
                     "C restatement of arroy's AVX2+FMA path (not arroy), data in RAM",
           "scan_gb_per_s": scan_rate * BYTES_PER_DISTANCE / 1e9}
    if not args.no_build:
        nb = min(n, 100_000)
        small = O.Data(O.COSINE, vecs[:nb], headers=big.headers[:nb])
        seeds = np.arange(1, cores + 1, dtype=np.uint64)
        t0 = time.perf_counter()
        evals = L.ao_build_forest_count(small.c(), 0, seeds.ctypes.data_as(C.c_void_p), len(seeds))
        el = time.perf_counter() - t0
        res["build_margins_per_s"] = evals / el
        res["build_sample"] = f"{len(seeds)} trees over {nb}x{DIMS} (one tree per thread), {el:.2f}s"
    return res


def measured_traffic(n_items):
    """HBM bytes per scan launch from the committed rocprofv3 --pmc passes (profiles/rNN_pmc_*_size.csv; separate
    FETCH_SIZE / WRITE_SIZE runs).  gfx950 correction per MI355X_MICROARCH.md: FETCH_SIZE counts half of a
    16 B/lane coalesced read stream, so reads = FETCH_SIZE x 1024 x 2; writes = WRITE_SIZE x 1024."""
    import csv
    import glob
    if n_items != N_ITEMS:
        return None, None
    fetch = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_fetch_size.csv")))
    write = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_write_size.csv")))
    if not fetch or not write:
        return None, None

    def mean(path, counter):
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
                if "k_distances_f32<2, false>" in r["Kernel_Name"] and r["Counter_Name"] == counter]
        return sum(vals) / len(vals) if vals else None

    f, w = mean(fetch[-1], "FETCH_SIZE"), mean(write[-1], "WRITE_SIZE")
    if f is None or w is None:
        return None, None
    return f * 1024 * 2 + w * 1024, f"{os.path.basename(fetch[-1])} + {os.path.basename(write[-1])}"


def extra_c5(device):
    """BASELINE configs[4]: 5M x 768 binary-quantized vectors, Q=1 popcount scan (src/spaces/simple.rs:119-131)."""
    from arroy_amd import Dataset, distances
    out = {}
    n, iters = 5_000_000, 50
    for dist in (distances.BinaryQuantizedCosine, distances.BinaryQuantizedEuclidean, distances.BinaryQuantizedManhattan):
        ds = Dataset(dist, DIMS, n, device=device)
        ds.fill_synthetic(SEED, 1, n)
        ds.finalize()
        ds.bench_scan(7, n, 3)
        ms, _ = ds.bench_scan(7, n, iters)
        per = 96 + 4 + (4 if dist is distances.BinaryQuantizedCosine else 0)  # 12 words + out (+ stored norm)
        rate = n * iters / (ms * 1e-3)
        out[dist.name] = {"distances_per_s": rate, "bytes_per_distance": per, "gb_per_s": rate * per / 1e9,
                          "frac_of_hbm_peak": rate * per / 1e9 / HBM_PEAK_GBS, "kernel_ms": ms / iters}
        ds.close()
    return {"workload": "5M x 768 1-bit vectors, Q=1 scan", "metrics": out}


def extra_metrics(device):
    """Q=1 scan of 1M x 768 for every f32 metric (the north star names cosine / dot / Euclid / Manhattan)."""
    from arroy_amd import Dataset, distances
    out = {}
    n, iters = N_ITEMS, 50
    for dist in (distances.Euclidean, distances.Manhattan, distances.Cosine, distances.DotProduct):
        ds = Dataset(dist, DIMS, n, device=device)
        ds.fill_synthetic(SEED, 1, n)
        if dist is distances.DotProduct:
            ds.preprocess_dot()
        ds.finalize()
        ds.bench_scan(7, n, 3)
        ms, _ = ds.bench_scan(7, n, iters)
        per = 4 * DIMS + 4 + (4 if dist is distances.Cosine else 0)
        rate = n * iters / (ms * 1e-3)
        out[dist.name] = {"distances_per_s": rate, "bytes_per_distance": per, "gb_per_s": rate * per / 1e9,
                          "frac_of_hbm_peak": rate * per / 1e9 / HBM_PEAK_GBS, "kernel_ms": ms / iters}
        ds.close()
    return {"workload": f"{n}x{DIMS} f32 vectors, Q=1 scan", "metrics": out}


def _timed_callers(fn, batches, threads):
    """Seconds to push `batches` through `fn` from `threads` concurrent callers (two untimed passes first so every
    caller's stream, pinned staging and scratch exist at their final size)."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(threads) as pool:
        for _ in range(2):
            list(pool.map(fn, batches))
        t0 = time.perf_counter()
        list(pool.map(fn, batches))
        return time.perf_counter() - t0


def extra_c4(device):
    """BASELINE configs[3]: 1M x 1536 dot product, search_k=10000 candidate re-rank + top-100, 1000 queries
    (src/reader.rs:376-400).  Candidate lists are 10 000..11 535 sorted random ids (search_k <= |nns| < search_k + K)."""
    import numpy as np

    from arroy_amd import Dataset, distances
    n, dims, nq, k = 1_000_000, 1536, 1000, 100
    ds = Dataset(distances.DotProduct, dims, n, device=device)
    ds.fill_synthetic(SEED, 1, n)
    ds.preprocess_dot()
    ds.finalize()
    rng = np.random.default_rng(SEED)
    queries = np.stack([ds.item_vector(int(i)) for i in rng.choice(n, 64, replace=False)])
    queries = np.tile(queries, (nq // 64 + 1, 1))[:nq]
    lists = [np.sort(rng.choice(n, int(rng.integers(10_000, 11_536)), replace=False)).astype(np.uint32) for _ in range(nq)]
    total = sum(len(l) for l in lists)
    per = 4 * dims + 4 + 4  # vector + id + written distance
    out = {"workload": f"{n}x{dims} dot product, {nq} queries x ~10.8k candidates, top-{k} (host in/out included)",
           "bytes_per_candidate": per}
    # the reference's readers are concurrent (`Reader: Sync`, one RoTxn per thread): 1 caller, then 4 callers, each
    # with its own stream + scratch inside the library (ctypes drops the GIL during the call)
    def flat(b, e):  # the C ABI's own shape: concatenated ids + offsets (what a Rust caller holds)
        off = np.zeros(e - b + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(l) for l in lists[b:e]])
        return queries[b:e], (np.concatenate(lists[b:e]), off)
    batches = [flat(b, min(nq, b + 125)) for b in range(0, nq, 125)]
    for threads in (1, 4):
        el = _timed_callers(lambda a: ds.rerank_batch(a[0], a[1], k), batches, threads)
        out[f"callers_{threads}"] = {"queries_per_s": nq / el, "candidates_per_s": total / el,
                                     "gb_per_s": total / el * per / 1e9,
                                     "frac_of_hbm_peak": total / el * per / 1e9 / HBM_PEAK_GBS, "seconds": el}
    # one submission with all queries: >= 2 candidates per stored row, so the library re-ranks row-major
    # (each row leaves HBM once per submission instead of once per candidate)
    el = _timed_callers(lambda a: ds.rerank_batch(a[0], a[1], k), [flat(0, nq)], 1)
    out["one_submission"] = {"queries_per_s": nq / el, "candidates_per_s": total / el,
                             "effective_gb_per_s": total / el * per / 1e9, "seconds": el}
    ds.close()
    return out


def extra_staging(device):
    """PCIe-inclusive staging: 250k x 768 f32 vectors from pageable host memory through the pinned
    double buffer into HBM (`ah_dataset_upload_vectors`: rows re-pitched, norms computed on device)."""
    import numpy as np

    from arroy_amd import Dataset, distances
    from oracle import oracle as O
    n = 250_000
    vecs = O.synth(SEED, 1, n, DIMS)
    ids = np.arange(n, dtype=np.uint32)
    ds = Dataset(distances.Cosine, DIMS, n, device=device)
    t0 = time.perf_counter()
    ds.upload_vectors(ids, vecs)
    ds.finalize()
    el = time.perf_counter() - t0
    ds.close()
    return {"workload": f"{n}x{DIMS} f32 from pageable host memory", "seconds": el, "gb_per_s": n * DIMS * 4 / el / 1e9}


def extra_search(device):
    """End-to-end on-device search on the configs[3] shape: 1M x 1536 dot product, 20 trees, 1000 by-vector queries,
    count=100, search_k=10000: descent + candidate collection + sort/dedup + re-rank + top-k (src/reader.rs:317-401)."""
    import numpy as np

    from arroy_amd import Dataset, distances, shard
    n, dims, nq, k, n_trees = 1_000_000, 1536, int(os.environ.get("AH_BENCH_SEARCH_QUERIES", "1000")), 100, 20
    ds = Dataset(distances.DotProduct, dims, n, device=device)
    ds.fill_synthetic(SEED, 1, n)
    ds.preprocess_dot()
    ds.finalize()
    t0 = time.perf_counter()
    forest = ds.build_forest(shard.tree_seeds(SEED, range(n_trees)))
    build_s = time.perf_counter() - t0
    index = ds.create_index(forest)
    rng = np.random.default_rng(SEED)
    queries = np.stack([ds.item_vector(int(i)) for i in rng.choice(n, 64, replace=False)])
    queries = (np.tile(queries, (nq // 64 + 1, 1))[:nq] + rng.standard_normal((nq, dims)).astype(np.float32) * 0.05).astype(np.float32)
    _ids, _d, counts = index.search(k, queries=queries[:64], search_k=10_000, raw=True)
    out = {"workload": f"{n}x{dims} dot product, {n_trees} trees, {nq} queries, count={k}, search_k=10000 (host in/out included)",
           "forest_build_seconds": build_s, "results_per_query": float(counts.mean())}
    # best-first descent is sequential per query, so a call wants many queries: every caller submits all `nq`
    for threads in (1, 4):
        el = _timed_callers(lambda q: index.search(k, queries=q, search_k=10_000, raw=True), [queries] * threads, threads)
        out[f"callers_{threads}"] = {"queries_per_s": threads * nq / el, "queries": threads * nq, "seconds": el}
    index.close()
    forest.close()
    ds.close()
    return out


def extra_c3(device, my_seeds_fn):
    """BASELINE configs[2]: 10M x 768 cosine, n_trees=100; this rank's share of the trees (all 100 at N=1)."""
    from arroy_amd import Dataset, distances
    n = 10_000_000
    ds = Dataset(distances.Cosine, DIMS, n, device=device)
    ds.fill_synthetic(SEED, 1, n)
    ds.finalize()
    seeds = my_seeds_fn(100)
    t0 = time.perf_counter()
    forest = ds.build_forest(seeds)
    el = time.perf_counter() - t0
    st = forest.stats
    out = {"workload": f"{n}x{DIMS} cosine, n_trees=100, trees on this rank: {len(seeds)}", "seconds": el,
           "seconds_library": st["seconds_total"], "seconds_device": st["seconds_device"],
           "seconds_margin_kernel": st["seconds_margin"], "margin_evaluations": st["margin_evaluations"],
           "levels": st["levels"], "split_nodes": st["split_nodes"],
           "margin_effective_gb_per_s": st["margin_evaluations"] * 4 * DIMS / st["seconds_margin"] / 1e9,
           "margin_row_major_passes": st["margin_row_passes"]}
    forest.close()
    ds.close()
    return out


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if args.dry_run else "nccl", rank=rank, world_size=world)

    import torch

    def barrier_sync():
        if dist is not None:
            if args.dry_run:
                dist.barrier()
            else:
                dist.barrier(device_ids=[local_rank])
        if not args.dry_run:
            torch.cuda.synchronize()

    def max_over_ranks(x: float) -> float:
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if args.dry_run else f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    from arroy_amd import shard
    my_trees = shard.trees_for_rank(args.trees, rank, world)
    my_seeds = shard.tree_seeds(SEED, my_trees)

    n = args.items
    result = {}
    if args.dry_run:
        # control-path test: no device work, fixed fake durations
        barrier_sync()
        t0 = time.perf_counter()
        time.sleep(0.01 * (rank + 1))
        barrier_sync()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        scan_ms = elapsed * 1e3 / max(args.steps, 1)
        build = {"trees": args.trees, "trees_this_rank": len(my_trees), "seconds": max_over_ranks(0.001 * len(my_trees))}
        kernel_ms = scan_ms
        copy_gbs = read_gbs = None
        dev_name = "dry-run (cpu, gloo)"
        cpu = None
    else:
        torch.cuda.set_device(local_rank)
        import arroy_amd
        from arroy_amd import Dataset, distances
        dev_name = arroy_amd.device_name(local_rank)
        ds = Dataset(distances.Cosine, DIMS, n, device=local_rank)
        ds.fill_synthetic(SEED, 1, n)
        ds.finalize()
        query_item = 12345 % n
        if args.warmup > 0:
            ds.bench_scan(query_item, n, args.warmup)
        barrier_sync()
        t0 = time.perf_counter()
        kernel_ms_total, _ = ds.bench_scan(query_item, n, args.steps)  # K launches, HIP events on the launch stream
        barrier_sync()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        kernel_ms = max_over_ranks(kernel_ms_total / args.steps)
        # measured streaming ceiling of this device, next to the spec peak: a 2 GiB device-to-device copy
        from arroy_amd import _lib as ahlib
        cp_bytes, cp_iters = 2 << 30, 10
        cp_ms = ahlib.bench_memcpy(local_rank, cp_bytes, cp_iters)
        copy_gbs = 2 * cp_bytes * cp_iters / (cp_ms * 1e-3) / 1e9  # read + write
        rd_bytes, rd_iters = 3 << 30, 20
        read_gbs = rd_bytes * rd_iters / (ahlib.bench_read(local_rank, rd_bytes, rd_iters) * 1e-3) / 1e9
        build = None
        if not args.no_build and args.trees > 0:
            barrier_sync()
            t0 = time.perf_counter()
            forest = ds.build_forest(my_seeds) if my_seeds else None
            barrier_sync()
            b_elapsed = max_over_ranks(time.perf_counter() - t0)
            st = forest.stats if forest is not None else {}
            margin_s = st.get("seconds_margin", 0.0)
            evals = st.get("margin_evaluations", 0)
            build = {
                "workload": f"{n}x{DIMS} cosine, n_trees={args.trees} (split_after={DIMS}), trees t=rank mod {world}",
                "trees": args.trees, "trees_this_rank": len(my_trees), "seconds": b_elapsed,
                "seconds_library_rank0": st.get("seconds_total"),
                "seconds_device_rank0": st.get("seconds_device"), "seconds_margin_kernel_rank0": margin_s,
                "margin_evaluations_rank0": evals, "levels": st.get("levels"),
                "margins_per_s_rank0": evals / margin_s if margin_s else None,
                # algorithmic = 4*dims bytes per (item, node visit).  With row-major passes one HBM read of a row serves
                # several trees, so this is an EFFECTIVE rate (it may exceed the HBM peak), not a roofline fraction.
                "margin_effective_gb_per_s_rank0": evals * 4 * DIMS / margin_s / 1e9 if margin_s else None,
                "margin_row_major_passes_rank0": st.get("margin_row_passes"),
                "split_nodes_rank0": st.get("split_nodes"), "retries_rank0": st.get("retries"),
                "dummy_normals_rank0": st.get("dummy_normals"), "scaling": "strong",
            }
        cpu = None
        if rank == 0 and world == 1 and not args.no_cpu:
            cpu = cpu_baseline(args, n)
        extra = {}
        wanted = [x for x in args.extra.split(",") if x]
        ds.close()
        # "tree-build seconds at 10M vectors" (BASELINE.json metric, configs[2]): 100 trees sharded over the ranks
        build_10m = None
        if not args.no_build and not args.no_build_10m and n == N_ITEMS and "c3" not in wanted:
            wanted.append("c3")
        if "c5" in wanted and rank == 0:
            extra["c5"] = extra_c5(local_rank)
        if "metrics" in wanted and rank == 0:
            extra["metrics"] = extra_metrics(local_rank)
        if "c4" in wanted and rank == 0:
            extra["c4"] = extra_c4(local_rank)
        if "staging" in wanted and rank == 0:
            extra["staging"] = extra_staging(local_rank)
        if "search" in wanted and rank == 0:
            extra["search"] = extra_search(local_rank)
        if "c3" in wanted:
            barrier_sync()
            t0 = time.perf_counter()
            c3 = extra_c3(local_rank, lambda t: shard.tree_seeds(SEED, shard.trees_for_rank(t, rank, world)))
            barrier_sync()
            c3["seconds_max_over_ranks"] = max_over_ranks(time.perf_counter() - t0)
            c3["seconds"] = max_over_ranks(c3["seconds"])
            c3["scaling"] = "strong"
            build_10m = c3
        result["extra"] = extra
        result["build_10m"] = build_10m

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = world * n * args.steps / elapsed
        achieved = n * BYTES_PER_DISTANCE / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_src = (args.traffic_bytes, "--traffic-bytes") if args.traffic_bytes else measured_traffic(n)
        line = {
            "metric": "distances/sec, Q=1 batched 768-dim cosine scan (GB/s vs HBM roofline in `roofline`)",
            "value": value, "unit": "distances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic uniform[-1,1) (counter-based generator, seed 42), generated in HBM",
            "config": {"workload": f"{n}x{DIMS} cosine Q=1 distance scan, one replica per GPU (BASELINE configs[1])",
                       "items": n, "dims": DIMS, "metric": "cosine", "device": dev_name},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "ah::k_distances_f32<COSINE,false>", "kernel_ms": kernel_ms,
                         "measured_d2d_copy_gb_per_s": copy_gbs, "measured_read_only_gb_per_s": read_gbs,
                         "frac_of_measured_read_ceiling": achieved / read_gbs if read_gbs else None,
                         "algorithmic_bytes_per_launch": n * BYTES_PER_DISTANCE},
            "cpu_baseline": cpu,
            "build": build,
            "build_10m": result.get("build_10m"),
        }
        if result.get("extra"):
            line["extra"] = result["extra"]
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
