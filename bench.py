#!/usr/bin/env python3
"""bench.py — the headline measurement of arroy's distance-kernel hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 runs either way: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (one
rank per GPU, RCCL only for the barrier and the max of the elapsed time; WORLD_SIZE must equal --gpus), or started
directly, in which case THIS process drives the N devices from N host threads — the shape of arroy's own build, which is
one process (`Writer::build`, src/writer.rs:556-591): the dataset is staged once on device 0 and replicated device to
device (`ah_dataset_replicate`, xGMI), every thread owns one replica and builds the trees t = i (mod N).  Fewer than N
visible devices is an error (exit code 3), never a silent N = 1 run.

Workload (BASELINE.json configs[1]): 1M x 768-dim cosine, synthetic i.i.d. uniform[-1,1) vectors generated
in HBM by the counter-based generator of include/arroy_hip_policy.h (seed 42), ids 0..N-1.
  * A "step" = one Q=1 batched cosine-distance scan over all 1M resident items (one kernel launch):
    `D::built_distance(query, item)` for every item, src/reader.rs:381-391 -> src/distance/cosine.rs:43-59.
    Inputs are resident in HBM when the timed region starts; outputs stay in HBM.
  * value = total distances/s over all devices (every device scans its own replica: weak scaling).
  * roofline: algorithmic bytes per launch = 1M x (4*768 + 4 header + 4 out) = 3080 B/distance
     / the average kernel time measured with HIP events on the launch stream.
  * build: the n_trees=50 forest of the same config, trees sharded round-robin over devices, no collective;
    build_10m: the 10M x 768, n_trees=100 forest of configs[2] ("tree-build seconds at 10M vectors"), same sharding;
    both with the certified binary16 screen (default) and, for build_10m, also in f32 arithmetic only.
  * rerank / bq_scan / search: BASELINE configs[3], configs[4] and the on-device search, on device 0.
  * cpu_baseline (N=1 only): the C oracle (a restatement of arroy's AVX2+FMA path, NOT arroy) on the host cores: the
    full 1M-row scan and the full configs[1] build (50 trees over 1M x 768), unless --cpu-seconds bounds it; and for
    configs[2] 8 (or one per core) of the 100 trees over all 10M rows, scaled to 100.
  * build_10m also carries: `identical` (content digest of the screened forest == that of the f32-only forest, computed
    in this run; a mismatch makes bench.py exit 5), `cold` (staging from host memory + the first build of the dataset,
    which makes the binary16 / int8 copies), `normal` (the same build on ~N(0,1) rows), `share_13` (the 13-tree share one
    GPU of eight would build, on this GPU).
One JSON line on stdout.  `--dry-run` exercises both N>1 control paths on CPU (gloo ranks / host threads).
"""
import argparse
import hashlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues per priority (its own switch, default 4);
# the four-caller legs bring a stream per caller next to the datasets' own, and two callers on one queue take turns: 8 queues give
# `rerank.callers_4` 190-203 -> 221-223 k and `search.callers_4` 552 -> 598-608 k queries/s, every one-caller figure unchanged
# (profiles/r06_experiments.txt).  Set before anything touches the device, reported in the line (`env`); a caller's own value wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

N_ITEMS = 1_000_000
ORACLE_TREES = (0, 99)  # trees of the 10M x 768 x 100 build the oracle also builds, whole, on the host cores (build_10m.oracle_tree)
DIMS = 768
N_TREES = 50
SEED = 42
BYTES_PER_DISTANCE = 4 * DIMS + 4 + 4  # vector + stored norm + written distance
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
SCAN_KERNEL_SOURCES = ["arroy_amd/csrc/distance.hip", "arroy_amd/csrc/device_math.h", "arroy_amd/csrc/common.h"]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--items", type=int, default=N_ITEMS)
    ap.add_argument("--trees", type=int, default=N_TREES)
    ap.add_argument("--no-build", action="store_true", help="skip the forest-build measurements")
    ap.add_argument("--no-build-10m", action="store_true",
                    help="skip the 10M x 768 x 100-tree build (BASELINE configs[2], the 'tree-build seconds at 10M' of the metric)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline")
    ap.add_argument("--no-extra", action="store_true", help="skip configs[3] / configs[4] / on-device search")
    ap.add_argument("--no-search", action="store_true", help="skip the on-device search leg (the PMC passes: its small "
                                                              "submissions share the re-rank kernel and would skew its per-launch mean)")
    ap.add_argument("--no-e2e", action="store_true",
                    help="build_10m from rows generated in HBM instead of 10M x 768 rows staged from host memory (skips the "
                         "cold end-to-end figures and the configs[2] CPU baseline, which share the host copy)")
    ap.add_argument("--cpu-seconds", type=float, default=45.0, help="budget of the CPU baseline (bounds the build sample)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: test the N>1 control paths on CPU")
    # (`--virtual-devices` under torch.distributed.run: its own parser takes a bare `--virtual` for an abbreviation of its
    # `--virtual-local-rank`, wherever on the command line it stands)
    ap.add_argument("--virtual", "--virtual-devices", dest="virtual", action="store_true",
                    help="N > 1 on ONE GPU: every rank / device thread uses device 0 (its own replica of the dataset, its own share "
                         "of the trees) — the real N > 1 code path end to end where only one GPU is at hand; the big build then "
                         "defaults to 1M items (N replicas of 10M x 768 and their screen copies do not fit one device)")
    ap.add_argument("--build-items", type=int, default=None,
                    help="items of the configs[2] build leg (default 10,000,000; 1,000,000 under --virtual)")
    ap.add_argument("--check-union", action="store_true",
                    help="N > 1: rank 0 also builds ALL trees and the union of the shares' per-tree digests must equal that build's "
                         "(exit 6 otherwise); always on under --virtual")
    ap.add_argument("--extra", default="", help="comma list of further measurements: metrics (every f32 metric), staging, "
                                               "e2e (10M x 768 staged from host memory + 100-tree build)")
    ap.add_argument("--cold-child", action="store_true",
                    help="child mode: the cold end-to-end build of configs[2] (staging from host memory + first build), alone in "
                         "this process; prints one JSON object")
    ap.add_argument("--scan-only", action="store_true",
                    help="child mode of the live PMC passes: fill the configs[1] dataset, launch the scan --steps times, exit")
    ap.add_argument("--pmc-child", default="scan", choices=["scan", "rerank", "bq_scan"],
                    help="with --scan-only: which roofline kernel the child launches (the Q=1 scan, the f32 re-rank gather of a "
                         "125-query submission, the 1-bit scan)")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not measure `roofline.traffic` in this run (two `rocprofv3 --pmc` child runs of the scan, ~15 s each); "
                         "quote the stored profiles/rNN_pmc_kernels.json figure instead")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per launch from a separate rocprofv3 --pmc pass (profiles/), if known")
    return ap.parse_args()


def scan_source_hash():
    """sha256 over the sources of the scan kernel: the PMC traffic figure is only quoted while it matches."""
    return source_hash("scan")


def usable_cpus():
    """CPUs this process may really use: the affinity mask, capped by a cgroup CPU quota if there is one (a container
    that shows 256 CPUs but is limited to a few dozen makes an all-threads OpenMP team thrash)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        txt = open("/sys/fs/cgroup/cpu.max").read().split()
        if txt and txt[0] != "max":
            quota = float(txt[0]) / float(txt[1])
    except OSError:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except OSError:
            pass
    info = {"cpu_count": os.cpu_count(), "affinity": n, "cgroup_quota_cpus": quota,
            "loadavg": list(os.getloadavg()) if hasattr(os, "getloadavg") else None}
    if quota:
        n = max(1, min(n, int(quota)))
    return n, info


def cpu_baseline(args, n_items):
    """The oracle timed on the host cores, same workload as the GPU numbers: the Q=1 cosine scan over ALL rows (3 GB:
    beyond the L3 of the host) and the configs[1] forest build over all rows — one tree per thread, the analogue of
    rayon's scope over the root tasks (src/writer.rs:568-591) — all 50 trees if a probe on a tenth of the rows says it
    fits the time budget, otherwise as many as fit, the rest extrapolated and said so."""
    import ctypes as C

    import numpy as np

    from oracle import oracle as O
    L = O.lib()
    cores, host = usable_cpus()
    L.ao_set_num_threads(cores)
    n = n_items
    vecs = O.synth(SEED, 1, n, DIMS)
    big = O.Data(O.COSINE, vecs)  # headers = exact norms, computed by the oracle itself (parallel)
    q, qh = big.item_leaf(0)
    out = np.zeros(n, dtype=np.float32)
    big.distances(q, qh)  # warm-up
    reps, t0 = 0, time.perf_counter()
    while True:
        L.ao_distances(big.c(), q.ctypes.data_as(C.c_void_p), qh.ctypes.data_as(C.c_void_p), None, n,
                       out.ctypes.data_as(C.c_void_p))
        reps += 1
        el = time.perf_counter() - t0
        if el > 3.0 or reps >= 100:
            break
    scan_rate = reps * n / el
    res = {"value": scan_rate, "unit": "distances/s", "cores": cores, "kind": "port",
           "sample": f"Q=1 cosine scan over all {n}x{DIMS} rows ({n * DIMS * 4 / 1e9:.1f} GB, data in RAM) x {reps} reps, "
                     f"OpenMP on {cores} threads; C restatement of arroy's AVX2+FMA path (not arroy)",
           "scan_gb_per_s": scan_rate * BYTES_PER_DISTANCE / 1e9, "host": host}
    if not args.no_build:
        team = max(1, min(cores, args.trees))
        L.ao_set_num_threads(team)  # n_trees >= team: the oracle runs one tree per thread
        seeds_all = np.arange(1, max(args.trees, team) + 1, dtype=np.uint64)
        # probe: `team` trees over a tenth of the rows (a tree over n rows costs ~ n log n row reads)
        n_probe = max(10_000, n // 10)
        small = O.Data(O.COSINE, vecs[:n_probe], headers=big.headers[:n_probe])
        t0 = time.perf_counter()
        L.ao_build_forest_count(small.c(), 0, seeds_all.ctypes.data_as(C.c_void_p), team)
        probe = time.perf_counter() - t0
        import math
        grow = (n / n_probe) * (math.log2(max(2.0, n / DIMS)) / math.log2(max(2.0, n_probe / DIMS)))
        rounds = math.ceil(args.trees / team)
        est_full = probe * grow * rounds
        k = args.trees
        if est_full > args.cpu_seconds:  # whole rounds of `team` concurrent trees that fit the budget, at least one
            k = min(args.trees, max(1, int(args.cpu_seconds / (probe * grow))) * team)
        t0 = time.perf_counter()
        evals = L.ao_build_forest_count(big.c(), 0, seeds_all.ctypes.data_as(C.c_void_p), k)
        total_s = time.perf_counter() - t0
        res["build_margins_per_s"] = evals / total_s
        res["build_seconds_measured"] = total_s
        res["build_trees_measured"] = k
        res["build_threads"] = team
        res["build_seconds_config_1"] = total_s * (args.trees / k)
        res["build_sample"] = (f"{k} of the {args.trees} trees of configs[1], each over all {n}x{DIMS} rows, one tree per thread on "
                               f"{team} threads: {total_s:.2f} s"
                               + ("" if k == args.trees else f"; build_seconds_config_1 scales it by {args.trees}/{k}"))
        L.ao_set_num_threads(cores)
    return res


KERNEL_SOURCES = {
    # kernel of a roofline entry -> the sources its PMC figure is stamped with
    "scan": SCAN_KERNEL_SOURCES,
    "rerank": ["arroy_amd/csrc/batch.hip", "arroy_amd/csrc/device_math.h", "arroy_amd/csrc/common.h"],
    "bq_scan": SCAN_KERNEL_SOURCES,
}


def source_hash(which):
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES[which]:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def measured_traffic(n_items, which="scan"):
    """HBM bytes per launch of a kernel from the committed rocprofv3 --pmc passes (profiles/rNN_pmc_kernels.json, written
    by scripts/collect_profiles.sh from separate FETCH_SIZE / WRITE_SIZE runs; gfx950 correction per MI355X_MICROARCH.md:
    FETCH_SIZE counts half of a 16 B/lane coalesced read stream).  Every entry carries the hash of the kernel sources it
    was measured on; a mismatch means the number is stale and is not quoted."""
    import glob
    if n_items != N_ITEMS:
        return None, "not measured for this size"
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_kernels.json")))
    if not files:
        return None, "no profiles/r*_pmc_kernels.json"
    j = json.load(open(files[-1])).get(which)
    if not j:
        return None, f"{os.path.basename(files[-1])} has no entry for {which}"
    if j.get("source_sha16") != source_hash(which):
        return None, f"{os.path.basename(files[-1])}:{which} was measured on other kernel sources (stale): re-run scripts/collect_profiles.sh"
    return float(j["hbm_bytes_per_launch"]), os.path.basename(files[-1])


PMC_KERNELS = {"scan": "k_distances_f32<2, false>", "rerank": "k_batch_distances_f32<3>", "bq_scan": "k_distances_bq<false>"}


def live_traffic(n_items, steps=6, which="scan"):
    """HBM bytes per launch of the scan kernel MEASURED IN THIS RUN: two child runs of `bench.py --scan-only` under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (each counter its own pass, no other trace domain — as the guide's
    HBM section prescribes), the dispatches of `k_distances_f32<2, false>` averaged, gfx950 correction applied (FETCH_SIZE
    tallies 128-byte requests at 64 bytes: x 2; WRITE_SIZE as is; both in KiB).  Returns (bytes or None, how)."""
    import csv
    import shutil
    import subprocess
    import tempfile
    tool = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if tool is None:
        return None, "rocprofv3 not found"
    means = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        tmp = tempfile.mkdtemp(prefix="ah_pmc_")
        try:
            cmd = [tool, "--pmc", counter, "--output-format", "csv", "-d", tmp, "-o", "pmc", "--", sys.executable,
                   os.path.abspath(__file__), "--scan-only", "--pmc-child", which, "--steps", str(steps), "--items", str(n_items)]
            env = dict(os.environ, TMPDIR="/tmp")
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120, check=True)
            vals = {}
            for root, _d, files in os.walk(tmp):
                for f in files:
                    if f.endswith("counter_collection.csv"):
                        for r in csv.DictReader(open(os.path.join(root, f))):
                            if PMC_KERNELS[which] in r["Kernel_Name"] and r["Counter_Name"] == counter:
                                vals[r["Dispatch_Id"]] = vals.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
            if not vals:
                return None, f"no {counter} rows for {PMC_KERNELS[which]} in the child run"
            means[counter] = sum(vals.values()) / len(vals)
        except (subprocess.SubprocessError, OSError, KeyError, ValueError) as e:
            return None, f"live {counter} pass failed: {type(e).__name__}"
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return means["FETCH_SIZE"] * 1024 * 2 + means["WRITE_SIZE"] * 1024, \
        f"measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child runs, {len(vals)} dispatches of {PMC_KERNELS[which]} each"


def rerank_lists(rng, n, nq):
    """nq candidate lists of configs[3]: 10 000 .. 11 535 sorted unique random ids each (what `nns.sort_unstable(); nns.dedup()`
    leaves, src/reader.rs:378-379)."""
    import numpy as np

    def one():
        m = int(rng.integers(10_000, 11_536))
        return np.unique(rng.integers(0, n, size=m + 400, dtype=np.uint32))[:m]
    return [one() for _ in range(nq)]


def scan_only(args):
    """Child of live_traffic: the dataset and the launches of ONE roofline kernel, nothing else (no torch, no timing)."""
    import numpy as np

    from arroy_amd import Dataset, distances
    from arroy_amd import _lib as ahlib
    if args.pmc_child == "rerank":  # the f32 gather of 125-query submissions (configs[3]), as extra_c4 times it
        n, dims = 1_000_000, 1536
        ds = Dataset(distances.DotProduct, dims, n, device=0)
        ds.fill_synthetic(SEED, 1, n)
        ds.preprocess_dot()
        ds.finalize()
        rng = np.random.default_rng(SEED)
        queries = np.stack([ds.item_vector(int(i)) for i in rng.choice(n, 64, replace=False)])
        queries = np.tile(queries, (2, 1))[:125]
        with ahlib.tuning(AH_RERANK_SCREEN=0):
            for _ in range(args.steps):
                ds.rerank_batch(queries, rerank_lists(rng, n, 125), 100)
        ds.close()
        return
    if args.pmc_child == "bq_scan":  # the three 1-bit metrics, as extra_c5 (its roofline entry is their mean)
        n = 5_000_000
        for dist in (distances.BinaryQuantizedCosine, distances.BinaryQuantizedEuclidean, distances.BinaryQuantizedManhattan):
            ds = Dataset(dist, DIMS, n, device=0)
            ds.fill_synthetic(SEED, 1, n)
            ds.finalize()
            ds.bench_scan(7, n, args.steps)
            ds.close()
        return
    ds = Dataset(distances.Cosine, DIMS, args.items, device=0)
    ds.fill_synthetic(SEED, 1, args.items)
    ds.finalize()
    ds.bench_scan(12345 % args.items, args.items, 1)
    ds.bench_scan(12345 % args.items, args.items, args.steps)
    ds.close()


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
    # roofline entry of the dominant kernel (BQ-cosine instantiation): algorithmic bytes per launch next to the PMC traffic
    # (one kernel serves the three 1-bit metrics: means over them, as the PMC summary averages its dispatches)
    traffic, src = measured_traffic(N_ITEMS, "bq_scan")
    gbs = sum(v["gb_per_s"] for v in out.values()) / len(out)
    roof = {"bound": "hbm", "kernel": "ah::k_distances_bq<false> (Q=1 scan of 5M rows, mean over the three 1-bit metrics)",
            "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "kernel_ms": sum(v["kernel_ms"] for v in out.values()) / len(out),
            "algorithmic_bytes_per_launch": n * sum(v["bytes_per_distance"] for v in out.values()) / len(out),
            "traffic": traffic, "traffic_source": src, "kernel_source_sha16": source_hash("bq_scan")}
    return {"workload": "5M x 768 1-bit vectors, Q=1 scan", "metrics": out, "roofline": roof}


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


def _timed_callers(fn, batches, threads, passes=9, spread=None):
    """Seconds to push `batches` through `fn` from `threads` concurrent callers: the median of nine timed passes (two
    untimed passes first so every caller's stream, pinned staging and scratch exist at their final size).  One caller runs
    the batches from this thread (no pool hand-off in the timed region).  spread: dict that receives min / max of the passes."""
    from concurrent.futures import ThreadPoolExecutor

    def one_pass(pool):
        if pool is None:
            for b in batches:
                fn(b)
        else:
            list(pool.map(fn, batches))
    pool = ThreadPoolExecutor(threads) if threads > 1 else None
    try:
        for _ in range(2):
            one_pass(pool)
        samples = []
        for _ in range(passes):  # a pass is 2-15 ms: one sample of it is at the mercy of one late thread
            t0 = time.perf_counter()
            one_pass(pool)
            samples.append(time.perf_counter() - t0)
    finally:
        if pool is not None:
            pool.shutdown()
    if spread is not None:
        spread.update(min=min(samples), max=max(samples), passes=passes)
    return sorted(samples)[passes // 2]


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
    lists = rerank_lists(rng, n, nq)
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
    from arroy_amd import _lib as ahlib
    # default: the certified top-k screen (candidates on the binary16 copy of the rows: 2 x dims bytes each; f32 rows only for
    # the ~1.5 % whose proven distance interval reaches the top k) — `gb_per_s` counts the bytes THAT path needs per candidate,
    # `f32_equivalent_gb_per_s` the 4 x dims + 8 of the reference's loop (an EFFECTIVE rate, not a roofline fraction)
    per_screen = 2 * dims + 4 + 4 + 0.015 * 4 * dims
    # round 6: the int8 copy of the rows in front of it (1 x dims bytes + the row's scale per candidate, a wider band of survivors —
    # measured below, `survivors_per_query`); a dataset whose candidates sit closer together than the int8 bound (clustered rows)
    # switches the stage off by itself and stays on the binary16 rows
    def per_screen8(survivors_per_candidate):
        return dims + 4 + 4 + 4 + 4 + survivors_per_candidate * 4 * dims
    def phases(threads):
        """Where one pass's wall time goes inside the library (ah_dataset_rerank_stats, AH_RERANK_TIMING=1; an extra pass, not
        one of the timed ones): seconds per pass of 1000 queries, summed over the calling threads."""
        with ahlib.tuning(AH_RERANK_TIMING=1):
            ds.rerank_stats(reset=True)
            _timed_callers(lambda a: ds.rerank_batch(a[0], a[1], k), batches, threads, passes=3)
            st = ds.rerank_stats(reset=True)
        n_pass = 5  # two warm-up passes + three timed
        return {key[8:]: st[key] / n_pass for key in st if key.startswith("seconds_")}
    def screened_leg(threads, int8):
        sp = {}
        ds.rerank_stats(reset=True)
        el = _timed_callers(lambda a: ds.rerank_batch(a[0], a[1], k), batches, threads, spread=sp)
        st = ds.rerank_stats(reset=True)
        on8 = int8 and st["chunks_int8"] > 0
        surv = st["survivors"] / max(1, st["queries_screened"])
        per_c = per_screen8(surv * nq / total) if on8 else per_screen
        return {"queries_per_s": nq / el, "candidates_per_s": total / el, "gb_per_s": total / el * per_c / 1e9,
                "frac_of_hbm_peak": total / el * per_c / 1e9 / HBM_PEAK_GBS,
                "f32_equivalent_gb_per_s": total / el * per / 1e9, "seconds": el, "seconds_passes": sp,
                "seconds_by_phase": phases(threads), "bytes_per_candidate": per_c, "survivors_per_query": surv,
                "int8_sub_batches": st["chunks_int8"], "int8_sub_batches_redone_on_binary16": st["chunks_int8_retried"],
                "path": "certified top-k screen (" + ("int8 rows first, " if on8 else "binary16 rows, ") + "f32 survivors)"}
    for threads in (1, 4):
        out[f"callers_{threads}"] = screened_leg(threads, True)
    with ahlib.tuning(AH_RERANK_SCREEN8=0):  # round 4's screen: binary16 rows for every candidate
        out["callers_1_binary16"] = screened_leg(1, False)
    with ahlib.tuning(AH_RERANK_SCREEN=0):  # the f32 gather for every candidate (rounds 1-3)
        for threads in (1, 4):
            sp = {}
            el = _timed_callers(lambda a: ds.rerank_batch(a[0], a[1], k), batches, threads, spread=sp)
            out[f"callers_{threads}_f32_only"] = {"queries_per_s": nq / el, "candidates_per_s": total / el,
                                                  "gb_per_s": total / el * per / 1e9,
                                                  "frac_of_hbm_peak": total / el * per / 1e9 / HBM_PEAK_GBS, "seconds": el,
                                                  "seconds_passes": sp, "seconds_by_phase": phases(threads)}
    # one submission with all queries: >= 2 candidates per stored row, so the library re-ranks row-major
    # (each row leaves HBM once per submission instead of once per candidate)
    out["phases_note"] = ("seconds_by_phase: one pass of the 1000 queries inside ah_rerank_batch, summed over the callers — prep "
                          "(tables, scratch, query copy), ids (host copies of the candidate ids into pinned memory), enqueue, "
                          "sync_wait (the host was done and the device was not: the device-bound share), device_span (HIP events, "
                          "first enqueue -> stream idle); wall - device_span = what the host alone costs the call")
    el = _timed_callers(lambda a: ds.rerank_batch(a[0], a[1], k), [flat(0, nq)], 1)
    out["one_submission"] = {"queries_per_s": nq / el, "candidates_per_s": total / el,
                             "effective_gb_per_s": total / el * per / 1e9, "seconds": el}
    # roofline entry of the gather kernel of the 125-query submissions: its algorithmic bytes per launch (mean over the
    # eight submissions) next to the PMC traffic; `achieved` is the END-TO-END rate of one caller (host in/out, top-k
    # rounds included), i.e. a lower bound of the kernel's own rate (rocprofv3 average in profiles/)
    traffic, src = measured_traffic(N_ITEMS, "rerank")
    one = out["callers_1_f32_only"]
    out["roofline"] = {"bound": "hbm", "kernel": "ah::k_batch_distances_f32<3> (DotProduct gather, 125-query submissions, AH_RERANK_SCREEN=0)",
                       "achieved": one["gb_per_s"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": one["frac_of_hbm_peak"],
                       "achieved_is": "end to end from one caller, host in/out and top-k included",
                       "algorithmic_bytes_per_launch": total / len(batches) * per, "traffic": traffic, "traffic_source": src,
                       "kernel_source_sha16": source_hash("rerank")}
    scr = out["callers_1"]
    out["roofline_screened"] = {"bound": "hbm", "kernel": "ah::k_pairs_screen8 + k_search_select_screened<3> (int8 gather, f32 survivors)"
                                if scr["int8_sub_batches"] else "ah::k_pairs_screen16 + k_search_select_screened<3> (binary16 gather, f32 survivors)",
                                "achieved": scr["gb_per_s"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": scr["frac_of_hbm_peak"],
                                "achieved_is": "end to end from one caller; bytes per candidate = dims + 16 + survivors x 4 x dims (int8 rows "
                                               "first) or 2 x dims + 8 + 1.5 % x 4 x dims (binary16 rows): fewer bytes per candidate, so a "
                                               "higher queries/s at a LOWER fraction — compare queries_per_s with callers_1_binary16",
                                "algorithmic_bytes_per_launch": total / len(batches) * scr["bytes_per_candidate"]}
    b16 = out["callers_1_binary16"]
    out["roofline_screened_binary16"] = {"bound": "hbm", "achieved": b16["gb_per_s"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": b16["frac_of_hbm_peak"], "kernel": "ah::k_pairs_screen16 + k_search_select_screened<3>"}
    ds.close()
    return out


def extra_staging(device):
    """PCIe-inclusive staging: 250k x 768 f32 vectors from pageable host memory through the pinned
    double buffer into HBM (`ah_dataset_upload_vectors`: rows re-pitched, norms computed on device)."""
    import numpy as np

    from arroy_amd import Dataset, distances
    from arroy_amd import _lib as ahlib
    n = 250_000
    vecs = ahlib.synth_rows_host(SEED, 1, n, DIMS)
    ids = np.arange(n, dtype=np.uint32)
    out = {"workload": f"{n}x{DIMS} f32 from pageable host memory, upload + finalize"}
    for run in ("first", "second"):  # the first dataset of a process also pays for the pinned staging ring
        tc = time.perf_counter()
        ds = Dataset(distances.Cosine, DIMS, n, device=device)
        t0 = time.perf_counter()
        ds.upload_vectors(ids, vecs)
        t1 = time.perf_counter()
        ds.finalize()
        el = time.perf_counter() - t0
        ds.close()
        out[run] = {"create_s": t0 - tc, "upload_call_s": t1 - t0, "finalize_s": el - (t1 - t0), "seconds": el,
                    "gb_per_s": n * DIMS * 4 / el / 1e9}
    return out


def extra_e2e(device, n=10_000_000, trees=100):
    """Cold build end to end on one GPU: 10M x 768 f32 vectors staged from pageable host memory (pinned ring, PCIe), then
    the 100-tree forest of configs[2].  Staging and build are timed separately (the build needs the finalized dataset)."""
    import numpy as np

    from arroy_amd import Dataset, distances, shard
    from arroy_amd import _lib as ahlib
    chunk = 1_000_000
    try:
        parts = [ahlib.synth_rows_host(SEED, 1, min(chunk, n - lo), DIMS, first_item=lo) for lo in range(0, n, chunk)]
    except MemoryError:
        return {"skipped": "host memory"}
    ds = Dataset(distances.Cosine, DIMS, n, device=device)
    t0 = time.perf_counter()
    for i, p in enumerate(parts):
        ds.upload_vectors(np.arange(i * chunk, i * chunk + len(p), dtype=np.uint32), p)
    t1 = time.perf_counter()
    ds.finalize()
    t2 = time.perf_counter()
    forest = ds.build_forest(shard.tree_seeds(SEED, range(trees)))
    t3 = time.perf_counter()
    out = {"workload": f"{n}x{DIMS} cosine from pageable host memory, n_trees={trees}, one GPU, cold (first build of the dataset)",
           "staging_s": t2 - t0, "staging_calls_s": t1 - t0, "staging_gb_per_s": n * DIMS * 4 / (t2 - t0) / 1e9,
           "build_s": t3 - t2, "total_s": t3 - t0}
    forest.close()
    ds.close()
    return out


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
    out = {"workload": f"{n}x{dims} dot product, {n_trees} trees, {nq} queries (64 items + noise), count={k}, search_k=10000 "
                       "(host in/out included)",
           "forest_build_seconds": build_s, "results_per_query": float(counts.mean())}
    # best-first descent is sequential per query, so a call wants many queries: every caller submits all `nq`
    for threads in (1, 4):
        el = _timed_callers(lambda q: index.search(k, queries=q, search_k=10_000, raw=True), [queries] * threads, threads)
        out[f"callers_{threads}"] = {"queries_per_s": threads * nq / el, "queries": threads * nq, "seconds": el}
    # the same call when no two queries start from the same item (they share leaves by chance only: more HBM rows per query)
    far = np.stack([ds.item_vector(int(i)) for i in rng.choice(n, nq, replace=False)])
    far = (far + rng.standard_normal((nq, dims)).astype(np.float32) * 0.05).astype(np.float32)
    index.search(k, queries=far[:64], search_k=10_000, raw=True)
    el = _timed_callers(lambda q: index.search(k, queries=q, search_k=10_000, raw=True), [far], 1)
    out["callers_1_distinct_items"] = {"queries_per_s": nq / el, "queries": nq, "seconds": el}
    # `QueryBuilder::candidates`: a filter that keeps half of the items (every search of a filtered index has one)
    half = np.sort(rng.choice(n, n // 2, replace=False)).astype(np.uint32)
    run = lambda q: index.search(k, queries=q, search_k=10_000, raw=True, candidates=half, candidates_sorted=True)  # noqa: E731
    run(queries[:64])
    el = _timed_callers(run, [queries], 1)
    out["callers_1_filter_half"] = {"queries_per_s": nq / el, "queries": nq, "seconds": el}
    out["stats"] = index.stats()  # which descent tier / dedup path / re-rank path served the timed calls (ah_index_search_stats)
    from oracle import oracle as O  # the checker (and the one-core CPU figure of the latency table): nothing timed above runs through it
    odata = O.Data(O.DOT_PRODUCT, O.synth(SEED, 1, n, dims))
    odata.preprocess_dot()
    out["latency"] = search_latency(ds, index, forest, far, n, dims, k, rng, odata)
    out.update(verify_search(ds, index, forest, n, dims, k, queries, far, half, od=odata))
    index.close()
    forest.close()
    ds.close()
    return out


def _percentiles(samples_s):
    import numpy as np
    a = np.sort(np.asarray(samples_s)) * 1e6
    return {"calls": int(a.size), "p50_us": float(a[a.size // 2]), "p90_us": float(a[int(a.size * 0.9)]),
            "p99_us": float(a[min(a.size - 1, int(a.size * 0.99))]), "mean_us": float(a.mean()), "min_us": float(a[0])}


def search_latency(ds, index, forest, far, n, dims, k, rng, odata, calls=300):
    """Per-call latency of the paths arroy's API takes one query at a time (`QueryBuilder::by_vector`, src/reader.rs:46-75, one
    `nns_by_leaf` per call, src/reader.rs:317-401): ah_search_batch at nq = 1 / 8 / 64 (distinct queries every call: nothing is
    answered from a warm leaf set) and ah_rerank_by_vector over one candidate list of configs[3] (10 000 - 11 535 sorted ids),
    straight through the C ABI with pre-built arguments (what a Rust caller pays; the ctypes call itself is ~2 us), host in/out
    and synchronisation included.  Beside them: the oracle's time for the same query on ONE host core."""
    import ctypes as C

    import numpy as np

    from arroy_amd import _lib as ahlib
    L = ahlib.lib()
    out = {"note": "wall time of one call through the C ABI (arguments pre-built, results written to caller memory); every call "
                   "takes queries it has not seen before",
           "search_k": 10_000, "count": k}
    for nq in (1, 8, 64):
        oi, od, oc = np.zeros((nq, k), np.uint32), np.zeros((nq, k), np.float32), np.zeros(nq, np.uint32)
        args_tail = (nq, k, 10_000, 0, None, 0, 0, oi.ctypes.data_as(C.c_void_p), od.ctypes.data_as(C.c_void_p), oc.ctypes.data_as(C.c_void_p))
        qs = [np.ascontiguousarray(far[(i * nq) % (len(far) - nq):][:nq]) for i in range(calls + 20)]
        ptrs = [q.ctypes.data_as(C.c_void_p) for q in qs]
        samples = []
        for i, qp in enumerate(ptrs):
            t0 = time.perf_counter()
            st = L.ah_search_batch(index._h, qp, None, *args_tail)
            el = time.perf_counter() - t0
            if st != 0:
                ahlib.check(st)
            if i >= 20:
                samples.append(el)
        e = _percentiles(samples)
        e["p50_us_per_query"] = e["p50_us"] / nq
        out[f"search_nq_{nq}"] = e
    # the re-rank alone (what integration/arroy-hip/src/hip.rs wires today: the descent stays in Rust)
    lists = rerank_lists(rng, n, 32)
    oi, od, on = np.zeros(k, np.uint32), np.zeros(k, np.float32), C.c_size_t(0)
    samples = []
    for i in range(calls + 20):
        ids = lists[i % len(lists)]
        q = far[i % len(far)]
        t0 = time.perf_counter()
        st = L.ah_rerank_by_vector(ds._h, q.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p), ids.size, k,
                                   oi.ctypes.data_as(C.c_void_p), od.ctypes.data_as(C.c_void_p), C.byref(on))
        el = time.perf_counter() - t0
        if st != 0:
            ahlib.check(st)
        if i >= 20:
            samples.append(el)
    out["rerank_by_vector"] = _percentiles(samples)
    out["rerank_by_vector"]["candidates"] = "10 000 - 11 535 sorted ids per call (configs[3])"
    # the oracle on one core, same queries / lists (the reference would add its LMDB page walks on top)
    from oracle import oracle as O
    cs, cr = [], []
    for i in range(6):
        qv, qh = odata.query_leaf(far[i])
        t0 = time.perf_counter()
        O.search(odata, forest, qv, qh, k, 10_000, 0, None, want_candidates=False)
        cs.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        odata.rerank(qv, qh, lists[i], k)
        cr.append(time.perf_counter() - t0)
    out["cpu_one_core"] = {"kind": "port", "search_us_per_query": float(np.median(cs)) * 1e6,
                           "rerank_us_per_query": float(np.median(cr)) * 1e6, "queries": len(cs),
                           "note": "oracle/arroy_oracle.c (ao_search / ao_rerank), rows in RAM, one thread"}
    return out


def verify_search(ds, index, forest, n, dims, k, queries, far, half, per_set=12, od=None):
    """`search.verified`: sampled queries of the timed sets — clustered, distinct items, under the half filter — answered
    again by the device and compared with the CPU oracle (`Reader::nns_by_leaf` restated, src/reader.rs:317-401) on the
    same forest: ids equal, distances bit-equal.  The oracle is the checker here, nothing timed runs through it."""
    import numpy as np

    from oracle import oracle as O
    if od is None:
        od = O.Data(O.DOT_PRODUCT, O.synth(SEED, 1, n, dims))
        od.preprocess_dot()
    checked, bad = 0, []
    for name, qs, cand in (("clustered", queries, None), ("distinct", far, None), ("filter_half", queries, half)):
        pick = np.linspace(0, len(qs) - 1, per_set).astype(int)
        sub = np.ascontiguousarray(qs[pick])
        index.stats(reset=True)
        ids, dist, counts = index.search(k, queries=sub, search_k=10_000, raw=True, candidates=cand, candidates_sorted=True)
        for i in range(len(sub)):
            qv, qh = od.query_leaf(sub[i])
            want, _ = O.search(od, forest, qv, qh, k, 10_000, 0, cand, candidates_sorted=True, want_candidates=False)
            wi = [a for a, _ in want]
            wd = np.array([b for _, b in want], dtype=np.float32)
            ok = int(counts[i]) == len(want) and list(ids[i, :counts[i]]) == wi and \
                dist[i, :counts[i]].view(np.uint32).tolist() == wd.view(np.uint32).tolist()
            checked += 1
            if not ok:
                bad.append(f"{name}[{int(pick[i])}]")
    return {"verified": not bad, "verified_queries": checked, "verified_against": "CPU oracle (oracle/arroy_oracle.c: ao_search) on the "
            "same forest: ids equal, distances bit-equal; clustered / distinct / half-filter queries of the timed sets",
            "mismatches": bad}


def build_stats(st, el, n, trees, my_trees, world):
    margin_s = st.get("seconds_margin", 0.0)
    evals = st.get("margin_evaluations", 0)
    return {
        "workload": f"{n}x{DIMS} cosine, n_trees={trees} (split_after={DIMS}), trees t = device (mod {world})",
        "trees": trees, "trees_this_rank": len(my_trees), "seconds": el,
        "seconds_library_rank0": st.get("seconds_total"), "seconds_device_rank0": st.get("seconds_device"),
        "seconds_margin_kernel_rank0": margin_s, "margin_evaluations_rank0": evals, "levels": st.get("levels"),
        "margins_per_s_rank0": evals / margin_s if margin_s else None,
        # algorithmic = 4*dims bytes per (item, node visit).  Row-major passes serve several trees per HBM read of a
        # row and the screen reads binary16 copies, so this is an EFFECTIVE rate, not a roofline fraction.
        "margin_effective_gb_per_s_rank0": evals * 4 * DIMS / margin_s / 1e9 if margin_s else None,
        "margin_row_major_passes_rank0": st.get("margin_row_passes"), "split_nodes_rank0": st.get("split_nodes"),
        "retries_rank0": st.get("retries"), "dummy_normals_rank0": st.get("dummy_normals"),
        "screened_launches_rank0": st.get("screened_launches"), "screen_fallbacks_rank0": st.get("screen_fallbacks"),
        "margin_mode_launches_rank0": st.get("margin_mode_launches"),
        # levels whose first attempt ran as one binary16 MFMA product (rows x normals^T) and the columns they covered
        "dense_mfma_levels_rank0": st.get("dense_launches"), "dense_mfma_columns_rank0": st.get("dense_columns"),
        "scaling": "strong",
    }


class RankSync:
    """Barrier + max-over-ranks for one-process-per-GPU runs (torch.distributed: RCCL, or gloo for --dry-run)."""

    def __init__(self, args, rank, world, local_rank):
        import torch
        self.torch, self.dist, self.args, self.local_rank = torch, None, args, local_rank
        # RCCL wants one rank per device: the ranks of a --virtual run share device 0 and talk over gloo (control path only —
        # there is no collective on the data path either way)
        self.cpu_group = bool(args.dry_run or args.virtual)
        if world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo" if self.cpu_group else "nccl", rank=rank, world_size=world)
            self.dist = dist

    def barrier(self, _i=0):
        if not self.args.dry_run:
            self.torch.cuda.synchronize()  # this rank's device work is done before it reports to the barrier
        if self.dist is not None:
            if self.cpu_group:
                self.dist.barrier()
            else:
                self.dist.barrier(device_ids=[self.local_rank])
        if not self.args.dry_run:
            self.torch.cuda.synchronize()

    def max(self, x, _i=0):
        if self.dist is None:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device="cpu" if self.cpu_group else f"cuda:{self.local_rank}")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather(self, obj, _i=0):
        """`obj` of every rank, in rank order, on every rank."""
        if self.dist is None:
            return [obj]
        out = [None] * self.dist.get_world_size()
        self.dist.all_gather_object(out, obj)
        return out

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


class ThreadSync:
    """Barrier + max over the N host threads of a single-process run (one thread per device)."""

    def __init__(self, args, world):
        self.args, self.world = args, world
        self.bar = threading.Barrier(world)
        self.vals = [0.0] * world
        self.objs = [None] * world
        self.torch = None
        if not args.dry_run:
            import torch
            self.torch = torch

    def barrier(self, i):
        self.bar.wait()
        if self.torch is not None:
            self.torch.cuda.synchronize(0 if self.args.virtual else i)

    def max(self, x, i):
        self.vals[i] = x
        self.bar.wait()
        m = max(self.vals)
        self.bar.wait()
        return m

    def gather(self, obj, i):
        self.objs[i] = obj
        self.bar.wait()
        out = list(self.objs)
        self.bar.wait()
        return out

    def close(self):
        pass


def device_work(args, rank, world, device, sync, ds_1m, result):
    """Everything one device measures.  `ds_1m`: this device's replica of the configs[1] dataset (threads mode) or None
    (the rank fills its own)."""
    import arroy_amd
    from arroy_amd import Dataset, distances, shard
    from arroy_amd import _lib as ahlib
    n = args.items
    my_trees = shard.trees_for_rank(args.trees, rank, world)
    my_seeds = shard.tree_seeds(SEED, my_trees)
    ds = ds_1m
    if ds is None:
        ds = Dataset(distances.Cosine, DIMS, n, device=device)
        ds.fill_synthetic(SEED, 1, n)
        ds.finalize()
    query_item = 12345 % n
    if args.warmup > 0:
        ds.bench_scan(query_item, n, args.warmup)
    sync.barrier(rank)
    t0 = time.perf_counter()
    kernel_ms_total, _ = ds.bench_scan(query_item, n, args.steps)  # K launches, HIP events on the launch stream
    sync.barrier(rank)
    result["elapsed"] = sync.max(time.perf_counter() - t0, rank)
    result["kernel_ms"] = sync.max(kernel_ms_total / args.steps, rank)
    if rank == 0:
        result["device"] = arroy_amd.device_name(device)
        # measured streaming ceilings of this device, next to the spec peak
        cp_bytes, cp_iters = 2 << 30, 10
        cp_ms = ahlib.bench_memcpy(device, cp_bytes, cp_iters)
        result["copy_gbs"] = 2 * cp_bytes * cp_iters / (cp_ms * 1e-3) / 1e9  # read + write
        rd_bytes, rd_iters = 3 << 30, 20
        result["read_gbs"] = rd_bytes * rd_iters / (ahlib.bench_read(device, rd_bytes, rd_iters) * 1e-3) / 1e9
    if not args.no_build and args.trees > 0:
        if my_seeds:
            ds.build_forest(my_seeds[:1]).close()  # warm-up: binary16 shadow of the rows, scratch, pinned buffers
        samples, owns, stats = [], [], {}
        for _rep in range(3):  # the host shares a 16-CPU container with other jobs: median of three builds
            sync.barrier(rank)
            t0 = time.perf_counter()
            forest = ds.build_forest(my_seeds) if my_seeds else None
            owns.append(time.perf_counter() - t0)
            sync.barrier(rank)
            samples.append(sync.max(time.perf_counter() - t0, rank))
            if forest is not None:
                stats = forest.stats
                forest.close()
        if rank == 0:
            result["build"] = build_stats(stats, sorted(samples)[1], n, args.trees, my_trees, world)
            result["build"]["seconds_samples"] = samples
        result.setdefault("build_seconds_per_device", {})[rank] = sorted(owns)[1]
    if rank == 0 and world == 1 and not args.no_cpu:
        result["cpu"] = cpu_baseline(args, n)
    ds.close()
    wanted = [x for x in args.extra.split(",") if x]
    if rank == 0:
        extra = {}
        if not args.no_extra and n == N_ITEMS:
            extra["bq_scan"] = extra_c5(device)
            extra["rerank"] = extra_c4(device)
            if not args.no_search:
                extra["search"] = extra_search(device)
        if "metrics" in wanted:
            extra["metrics"] = extra_metrics(device)
        if "staging" in wanted:
            extra["staging"] = extra_staging(device)
        if "e2e" in wanted:
            extra["e2e"] = extra_e2e(device)
        result["extra"] = extra


def host_rows_10m(n):
    """n x 768 uniform[-1,1) rows in host memory (the generator of arroy_hip_policy.h, run on the host cores by the
    library's own harness entry ah_synth_rows_host), or None when the host cannot hold them next to the CPU baseline's
    working set."""
    import numpy as np

    from arroy_amd import _lib as ahlib
    need = n * DIMS * 4
    try:
        avail = next(int(l.split()[1]) * 1024 for l in open("/proc/meminfo") if l.startswith("MemAvailable"))
    except (OSError, StopIteration):
        avail = 0
    if avail < need * 1.6:
        return None, f"host has {avail / 1e9:.0f} GB available, {need * 1.6 / 1e9:.0f} GB wanted"
    try:
        vecs = np.empty((n, DIMS), dtype=np.float32)
    except MemoryError:
        return None, "host allocation failed"
    chunk = 1_000_000
    for lo in range(0, n, chunk):
        part = vecs[lo:lo + chunk]
        ahlib.synth_rows_host(SEED, 1, len(part), DIMS, first_item=lo, out=part)
    return vecs, None


def cold_child(args):
    """`bench.py --cold-child`: the COLD end-to-end build of configs[2], alone in a process that has done nothing else on the
    device — 10M x 768 f32 rows in pageable host memory -> ten ah_dataset_upload_vectors calls (pinned ring, PCIe) -> finalize
    -> the first 100-tree build of the dataset (which also makes the binary16 / int8 copies).  The parent starts it BEFORE it
    touches the GPU itself: what `cold.total_s` means is "first process on the box", not "behind whatever ran before" (round-5
    review: 2.07 s vs 3.17 s between two runs, depending on how much recently released HBM the driver was still wiping).
    Prints one JSON object."""
    import numpy as np

    from arroy_amd import Dataset, distances, shard
    n = args.build_items
    seeds = shard.tree_seeds(SEED, range(100))
    t_rows = time.perf_counter()
    host_vecs, why = host_rows_10m(n)
    t_rows = time.perf_counter() - t_rows
    if host_vecs is None:
        print(json.dumps({"skipped": why}), flush=True)
        return
    chunk = 1_000_000
    tc = time.perf_counter()
    ds = Dataset(distances.Cosine, DIMS, n, device=0)
    t0 = time.perf_counter()
    # `Writer::build` knows its tree count before it collects the items: the device memory of the first build is obtained on a
    # helper thread while the records travel (fresh HBM is not free: ah_dataset_reserve_build)
    ds.reserve_build(len(seeds))
    for lo in range(0, n, chunk):
        ds.upload_vectors(np.arange(lo, min(n, lo + chunk), dtype=np.uint32), host_vecs[lo:lo + chunk])
    t1 = time.perf_counter()
    ds.finalize()
    t2 = time.perf_counter()
    f = ds.build_forest(seeds)  # the first build of the dataset
    t3 = time.perf_counter()
    st = f.stats
    dig = f.digest()[0]
    f.close()
    t4 = time.perf_counter()
    g = ds.build_forest(seeds)  # ... and the second, for the same process's warm figure
    t5 = time.perf_counter()
    out = {"workload": f"{n}x{DIMS} cosine staged from pageable host memory (ten 1M-row ah_dataset_upload_vectors calls), then the "
                       f"first {len(seeds)}-tree build of the dataset (it makes the binary16 / int8 copies)",
           "means": "a process of its own, started by bench.py before the parent touched the device: the first work this box's GPU sees",
           "create_s": t0 - tc, "staging_s": t2 - t0, "staging_calls_s": t1 - t0, "staging_gb_per_s": n * DIMS * 4 / (t2 - t0) / 1e9,
           "first_build_s": t3 - t2, "total_s": t3 - t0, "first_build_library_s": st["seconds_total"],
           "first_build_device_s": st["seconds_device"], "first_build_setup_s": st.get("seconds_setup"),
           "reserve_thread_s": st.get("seconds_reserve"), "first_build_waited_for_reserve_s": st.get("seconds_reserve_wait"),
           "second_build_s": t5 - t4, "digest": f"{dig:016x}", "host_rows_s": t_rows}
    g.close()
    ds.close()
    print(json.dumps(out), flush=True)


def run_cold_child(args):
    """Start `bench.py --cold-child` and wait for it (the parent has not touched the GPU yet).  Returns its JSON or a reason."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cold-child", "--build-items", str(args.build_items)]
    t0 = time.perf_counter()
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    except (subprocess.SubprocessError, OSError) as e:
        return {"skipped": f"cold child failed: {type(e).__name__}"}
    for ln in reversed(p.stdout.strip().splitlines()):
        try:
            out = json.loads(ln)
            out["child_process_s"] = time.perf_counter() - t0
            return out
        except ValueError:
            continue
    return {"skipped": f"cold child rc={p.returncode}: {p.stderr.strip()[-300:]}"}


def timed_builds(ds, seeds, mode, reps, rank, sync, host_threads=0, tree_keys=None, keyed_out=None, hash_trees=None):
    """`reps` builds of `seeds` (barrier + max over ranks each); returns (per-rank seconds, max-over-ranks seconds,
    stats, digest) of the last one.  tree_keys + keyed_out: the last forest's per-tree digests keyed by the trees' indices in
    the whole index (the same whichever share builds a tree) are stored as keyed_out[tree index] = digest.
    hash_trees: {position in `seeds`: None} -> filled with the content hash of that whole tree of the last forest (the form the
    oracle's tree is hashed in: oracle.tree_hash — the checker, after the timed region)."""
    samples, owns, st, dig = [], [], {}, None
    for _rep in range(reps):
        sync.barrier(rank)
        t0 = time.perf_counter()
        forest = ds.build_forest(seeds, margin_mode=mode, max_host_threads=host_threads) if seeds else None
        owns.append(time.perf_counter() - t0)
        sync.barrier(rank)
        samples.append(sync.max(time.perf_counter() - t0, rank))
        if forest is not None:
            st = forest.stats
            if _rep == reps - 1:
                dig = forest.digest()[0]
                if tree_keys is not None and keyed_out is not None:
                    for t, d in zip(tree_keys, forest.digest_keyed(tree_keys)):
                        keyed_out[int(t)] = int(d)
                if hash_trees:
                    from oracle import oracle as O
                    for t in hash_trees:
                        hash_trees[t] = O.tree_hash(forest, t, 4, 4 * DIMS)
            forest.close()
    return owns, samples, st, dig


class _NoSync:
    """timed_builds for one rank on its own (the reference build of --check-union)."""

    def barrier(self, _i=0):
        pass

    def max(self, x, _i=0):
        return x


def union_digest(per_tree):
    """One 64-bit value over {tree index: keyed digest} in tree order: equal at every N (and for every sharding) iff every
    tree is the same tree."""
    h = 0xCBF29CE484222325
    for t in sorted(per_tree):
        h = ((h ^ (int(per_tree[t]) & 0xFFFFFFFFFFFFFFFF)) * 0x100000001B3 + t) & 0xFFFFFFFFFFFFFFFF
    return h


def build_entry(samples, st):
    ms = st.get("seconds_margin", 0.0)
    met8 = st.get("screen8_pairs", 0)
    return {"seconds": sorted(samples)[len(samples) // 2], "seconds_samples": samples,
            "seconds_spread": (max(samples) - min(samples)) / min(samples) if samples and min(samples) > 0 else None,
            "seconds_library": st.get("seconds_total"), "seconds_device": st.get("seconds_device"),
            # outside the kernels (last sample): entry -> first launch, last launch -> return; output blobs recycled from the pool
            "seconds_setup": st.get("seconds_setup"), "seconds_after_device": st.get("seconds_after_device"),
            "host_blob_recycled": st.get("host_blob_recycled"),
            # groups of trees the last big level and what followed it ran in (their ids and normals left the device under the next
            # group's kernels; 0: every level for all trees, the ids after the last launch)
            "tail_groups": st.get("tail_groups"),
            "seconds_margin_kernel": ms, "margin_evaluations": st.get("margin_evaluations"),
            "levels": st.get("levels"), "split_nodes": st.get("split_nodes"),
            "margin_effective_gb_per_s": st.get("margin_evaluations", 0) * 4 * DIMS / ms / 1e9 if ms else None,
            "margin_row_major_passes": st.get("margin_row_passes"),
            "screen_fallbacks": st.get("screen_fallbacks"), "margin_mode_launches": st.get("margin_mode_launches"),
            "dense_mfma_levels": st.get("dense_launches"), "dense_mfma_columns": st.get("dense_columns"),
            # schedule variants of the row-major pass (launches): one XCD per tree group / non-temporal rows / extra launches
            "rows_xcd_launches": st.get("rows_xcd_launches"), "rows_nt_launches": st.get("rows_nt_launches"),
            "rows_split_launches": st.get("rows_split_launches"),
            # int8 first stage of the node-major levels: pairs it met, share it decided
            "screen8_pairs": met8, "screen8_decided_frac": st.get("screen8_decided", 0) / met8 if met8 else None,
            "screen_unavailable": st.get("screen_unavailable")}


def build_10m(args, rank, world, device, sync, ds, result):
    """BASELINE configs[2]: 10M x 768 cosine, n_trees = 100, this device's share of the trees (all 100 at N = 1).
    Timed in both arithmetics — default (certified screens) and f32 only (AH_MARGIN_EXACT_ONLY) — and the two forests are
    COMPARED in the run (content digest).  At N = 1 the rows are staged from host memory first, so that the cold
    end-to-end time (staging + first build, which makes the shadow copies) is part of the line, and the same host copy
    feeds the configs[2] CPU baseline."""
    import numpy as np

    from arroy_amd import Dataset, distances, shard
    from arroy_amd import _lib as ahlib
    n = args.build_items
    trees = shard.trees_for_rank(100, rank, world)
    seeds = shard.tree_seeds(SEED, trees)
    # N > 1: the devices' builds share the host (src/writer.rs:538-548 gives ONE build the whole rayon pool): every build gets
    # its share of the cores for its output path instead of the default eight threads each
    host_threads = max(1, usable_cpus()[0] // world) if world > 1 else 0
    keyed = {}
    out, cold, host_vecs = {}, result.get("cold_10m"), None
    if ds is None:
        # What the legs before this one left in the library's device cache goes back to the driver here (the driver wipes
        # released HBM in the background, ~33 GB/s, DESIGN.md §3 — the CPU leg below gives it the time).
        ahlib.device_cache_trim()
        ds = Dataset(distances.Cosine, DIMS, n, device=device)
        why = "--no-e2e" if args.no_e2e else ("one host copy per rank would not fit" if world > 1 else None)
        if why is None and not args.no_cpu:
            host_vecs, why = host_rows_10m(n)
        if host_vecs is not None:
            # the CPU leg of configs[2] (and the oracle's whole trees) from a host copy of the rows; the copy is dropped before any
            # GPU build is timed (round-3 review: the driver's box punished 30.7 GB of idle rows next to the builds)
            result["cpu_10m"] = cpu_baseline_10m(args, host_vecs, result.get("cpu"))
            host_vecs = None
        # the same rows, generated in HBM (bit-identical to the host generator's: tests/test_gpu_structured.py); the staged,
        # cold build is the `cold` child process that ran before this process touched the device
        ds.fill_synthetic(SEED, 1, n)
        ds.finalize()
        if cold is None:
            cold = {"skipped": why or "no cold child (see --no-e2e / --virtual / N > 1)"}
    if seeds:
        ds.build_forest(seeds[:1]).close()  # warm-up (shadow copies of the rows, buffers)
    digests = {}
    # whole trees of THIS build against the oracle's (built on the host cores from the same rows in the CPU leg above)
    want_whole = (result.get("cpu_10m") or {}).get("oracle_trees") if world == 1 else None
    gpu_whole = {t: None for t in want_whole} if want_whole else None
    for key, mode, reps in (("screened", 0, 3), ("f32_only", ahlib.MARGIN_EXACT_ONLY, 1)):
        owns, samples, st, digests[key] = timed_builds(ds, seeds, mode, reps, rank, sync, host_threads, trees,
                                                       keyed if key == "screened" else None,
                                                       gpu_whole if key == "screened" else None)
        if rank == 0:
            out[key] = build_entry(samples, st)
        result.setdefault("build_10m_seconds_per_device", {}).setdefault(key, {})[rank] = sorted(owns)[len(owns) // 2]
        if key == "screened":
            # what a slow device of an N-GPU run looks like from one line: its own wall time (median and samples), the kernels,
            # the host-side head and tail of the build, the host threads it was allowed
            result.setdefault("build_10m_per_device", {})[rank] = {
                "device": device, "trees": len(trees), "seconds": sorted(owns)[len(owns) // 2], "seconds_samples": owns,
                "seconds_device": st.get("seconds_device"), "seconds_setup": st.get("seconds_setup"),
                "seconds_after_device": st.get("seconds_after_device"), "max_host_threads": host_threads or 8,
                "host_blob_recycled": st.get("host_blob_recycled")}
    if world == 1 and rank == 0 and seeds:
        # A/B in this run: every level for all trees to the end (AH_BUILD_TAIL_GROUPS=0, the build of rounds 1-5: the 4 GB of ids
        # and the last level's normals leave the device after the last launch)
        with ahlib.tuning(AH_BUILD_TAIL_GROUPS=0):
            _o, samples_l, st_l, dig_l = timed_builds(ds, seeds, 0, 2, rank, _NoSync(), host_threads, trees, None, None)
        out["screened"]["level_by_level"] = {"seconds": min(samples_l), "seconds_samples": samples_l,
                                             "seconds_device": st_l.get("seconds_device"),
                                             "seconds_after_device": st_l.get("seconds_after_device"),
                                             "tail_groups": st_l.get("tail_groups"), "identical": dig_l == digests["screened"]}
        if dig_l != digests["screened"]:
            print("FOREST MISMATCH: the build with the tail in groups of trees differs from the level-by-level build", file=sys.stderr)
            sys.exit(5)
    stream = None
    if world == 1 and rank == 0:
        # the same build through ah_build_forest_stream: split planes per level and item ids handed to a sink from the pinned
        # ring while the build runs, nothing materialised (the sink here only counts: what a consumer does with a batch is the
        # consumer's time — arroy would NodeCodec-encode into its TmpNodes files, src/parallel.rs:130-147)
        cnt = {"nodes": 0, "bytes": 0, "calls": 0}

        def sink(b):
            cnt["nodes"] += int(b.n_nodes)
            cnt["bytes"] += int(b.payload_len)
            cnt["calls"] += 1
            return 0
        ss, st_s = [], {}
        for _rep in range(3):
            cnt.update(nodes=0, bytes=0, calls=0)
            t0 = time.perf_counter()
            _roots, st_s, _c = ds.build_forest_stream(seeds, sink=sink)
            ss.append(time.perf_counter() - t0)
        stream = {"seconds": sorted(ss)[1], "seconds_samples": ss, "seconds_spread": (max(ss) - min(ss)) / min(ss),
                  "seconds_library": st_s.get("seconds_total"), "seconds_device": st_s.get("seconds_device"),
                  "seconds_after_device": st_s.get("seconds_after_device"), "sink_calls": cnt["calls"], "nodes": cnt["nodes"],
                  "gb_handed_over": cnt["bytes"] / 1e9,
                  "sink": "counts nodes and bytes (Python callback per batch); host memory held by the library: 64 MiB pinned ring + "
                          "the node table, instead of the 9.4 GB of the materialised forest"}
    identical = digests["screened"] == digests["f32_only"]
    result.setdefault("build_10m_identical_per_device", {})[rank] = bool(identical)
    # The shares put together: every tree's digest keyed by its index in the whole index (the same whichever device built it).
    # `union.digest` is one value over all 100 of them — equal at N = 1, 2, 4, 8 iff every run built the same forest; with
    # --check-union (always under --virtual) rank 0 also builds all 100 trees itself and compares tree by tree.
    all_keyed = {}
    for part in sync.gather(keyed, rank):
        all_keyed.update(part)
    union = None
    if rank == 0:
        union = {"trees": len(all_keyed), "digest": f"{union_digest(all_keyed):016x}", "complete": sorted(all_keyed) == list(range(100))}
    if world > 1 and (args.check_union or args.virtual):
        ref = {}
        if rank == 0:
            every = list(range(100))
            timed_builds(ds, shard.tree_seeds(SEED, every), 0, 1, 0, _NoSync(), host_threads, every, ref)
            bad = sorted(t for t in every if all_keyed.get(t) != ref.get(t))
            union.update(checked_against_one_device_build=True, identical=not bad, differing_trees=bad[:10])
            result["build_10m_union_ok"] = not bad
        sync.barrier(rank)
    if rank == 0:
        result["build_10m_union"] = union
    share = None
    if world == 1 and rank == 0:
        # the 13 trees GPU 0 of an 8-GPU node builds (t = 0 mod 8), on this GPU: the one-GPU proxy of the 8-GPU build time
        s13 = shard.tree_seeds(SEED, shard.trees_for_rank(100, 0, 8))
        _o, samples13, st13, _d = timed_builds(ds, s13, 0, 3, rank, sync)
        share = build_entry(samples13, st13)
        share["trees"] = len(s13)
        # the proxy of the 8-GPU speed-up from like quantities: wall minima of both (the steady state of either build),
        # the medians, and the device seconds (what no host hiccup touches)
        share["speedup_100_trees_over_share"] = min(out["screened"]["seconds_samples"]) / min(share["seconds_samples"])
        share["speedup_from_medians"] = out["screened"]["seconds"] / share["seconds"]
        share["speedup_from_device_seconds"] = out["screened"]["seconds_device"] / share["seconds_device"]
    ds.close()
    oracle_tree = None
    if want_whole:
        same = {t: gpu_whole[t] == want_whole[t]["hash"] for t in want_whole}
        oracle_tree = {"trees": sorted(want_whole), "identical": all(same.values()), "per_tree": same,
                       "hashes": {t: gpu_whole[t] for t in gpu_whole}, "oracle": want_whole,
                       "oracle_seconds": result["cpu_10m"].get("oracle_trees_seconds"),
                       "what": "whole trees of the 100-tree build (two_means, every split plane, every side, every Descendants "
                               "list: the canonical form hashed, oracle.tree_hash) against the trees the CPU oracle builds from the "
                               "same 10M x 768 rows and seeds (src/writer.rs:1167-1261, src/distance/mod.rs:126-171)"}
        result.setdefault("build_10m_identical_per_device", {})["oracle_tree"] = oracle_tree["identical"]
    other_data = {}
    if world == 1 and rank == 0:
        # the same build on rows that are not uniform: ~N(0,1) (long-tailed data is what the
        # quantised copies of the screen have to survive) and CLUSTERED rows (4096 skewed clusters, centre + N(0,1)/16, one row in
        # 61 an exact copy of its centre — the shape imported embeddings have; margins crowd the planes, imbalance retries and the
        # random fallback fire, the screens decide least).  Both arithmetics again, compared by digest.
        for name, dist, desc in (("normal", 2, "synthetic ~N(0,1) (sum of twelve uniforms, arroy_hip_policy.h AH_SYNTH_NORMAL), seed 42"),
                                 ("clustered", 4, "synthetic CLUSTERED (arroy_hip_policy.h AH_SYNTH_CLUSTERED: 4096 centres ~N(0,1) of "
                                                  "skewed sizes, rows = centre + N(0,1)/16, 1 row in 61 an exact duplicate of its centre), seed 42")):
            dn = Dataset(distances.Cosine, DIMS, n, device=device)
            dn.fill_synthetic(SEED, dist, n)
            dn.finalize()
            dn.build_forest(seeds[:1]).close()
            _o, sn, stn, dig_n = timed_builds(dn, seeds, 0, 3, rank, sync)
            _o, sx, stx, dig_x = timed_builds(dn, seeds, ahlib.MARGIN_EXACT_ONLY, 1, rank, sync)
            e = build_entry(sn, stn)
            e["data"] = desc
            e["f32_only_seconds"] = sx[0]
            e["identical"] = bool(dig_n == dig_x)
            e["digest"] = f"{dig_n:016x}"
            e["retries"], e["dummy_normals"] = stn.get("retries"), stn.get("dummy_normals")
            ev = stn.get("margin_evaluations") or 0
            e["screen_fallbacks_frac"] = stn.get("screen_fallbacks", 0) / ev if ev else None
            result.setdefault("build_10m_identical_per_device", {})[name] = e["identical"]
            other_data[name] = e
            dn.close()
    if rank == 0:
        res = dict(out["screened"])
        res["workload"] = f"{n}x{DIMS} cosine, n_trees=100, trees on device 0: {len(trees)} (t = device mod {world})"
        res["arithmetic"] = ("certified screens (int8 first on the node-major levels, binary16 — as one MFMA product on the top "
                             "levels — second) decide the side of a margin when |screen| > proven error bound, f32 reference "
                             "arithmetic for the rest (screen_fallbacks pairs); `identical` = the content digest of this "
                             "forest equals that of the f32_only build, compared in this run")
        res["identical"] = bool(identical)
        res["digest"] = f"{digests['screened']:016x}"
        res["digest_f32_only"] = f"{digests['f32_only']:016x}"
        res["f32_only"] = out["f32_only"]
        if stream is not None:
            res["stream"] = stream
        res["cold"] = cold
        if share is not None:
            res["share_13"] = share
        res.update(other_data)
        if oracle_tree is not None:
            res["oracle_tree"] = oracle_tree
        ev = out["screened"].get("margin_evaluations") or 0
        res["screen_fallbacks_frac"] = (out["screened"].get("screen_fallbacks") or 0) / ev if ev else None
        res["scaling"] = "strong"
        result["build_10m"] = res


def cpu_baseline_10m(args, vecs, cpu_1m):
    """configs[2] on the host cores: 8 of the 100 trees over all 10M rows — one tree per core when
    the --cpu-seconds budget allows (estimated from the configs[1] margin rate) — scaled to 100 trees, and said so."""
    import ctypes as C

    import numpy as np

    from oracle import oracle as O
    L = O.lib()
    cores, _host = usable_cpus()
    L.ao_set_num_threads(cores)
    n = len(vecs)
    data = O.Data(O.COSINE, vecs)
    rate = (cpu_1m or {}).get("build_margins_per_s") or 60e6
    per_tree = n * 14.05 / rate  # seconds per tree with all cores busy (14.05 margin evaluations per row and tree: GPU side)
    k = min(100, cores) if cores >= 8 and cores * per_tree <= args.cpu_seconds else 8
    seeds = np.arange(1, k + 1, dtype=np.uint64)
    t0 = time.perf_counter()
    evals = L.ao_build_forest_count(data.c(), 0, seeds.ctypes.data_as(C.c_void_p), k)
    el = time.perf_counter() - t0
    # ... and WHOLE trees of the headline build for the parity check of build_10m (`oracle_tree`): trees 0 and 99 of the 100, with
    # bench.py's own seeds, margin loops on all cores; only their content hashes are kept
    from arroy_amd import shard
    whole = {}
    t0 = time.perf_counter()
    for t in ORACLE_TREES:
        tr = data.build_tree(0, shard.tree_seeds(SEED, [t])[0])
        whole[t] = {"hash": O.tree_hash(tr.as_forest(data), 0, 4, 4 * DIMS), "nodes": len(tr.nodes), "retries": tr.retries,
                    "dummy_normals": tr.dummy_normals, "margin_evaluations": tr.margin_evals}
        del tr
    whole_s = time.perf_counter() - t0
    return {"build_seconds_config_2": el * 100.0 / k, "build_seconds_measured": el, "build_trees_measured": k,
            "oracle_trees": whole, "oracle_trees_seconds": whole_s,
            "build_margins_per_s": evals / el, "cores": cores, "kind": "port",
            "sample": f"{k} of the 100 trees of configs[2], each over all {n}x{DIMS} rows (data in RAM), OpenMP on {cores} threads "
                      f"({'one tree per thread' if k >= cores else 'trees one after the other, margin loops parallel'}): "
                      f"{el:.1f} s, scaled by 100/{k}; C restatement of arroy's path (not arroy)"}


def _dig(d, *path):
    for k in path:
        if not isinstance(d, dict) or d.get(k) is None:
            return None
        d = d[k]
    return d


def _r(x, nd=4):
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        return float(f"{x:.{nd}g}") if abs(x) < 1e4 else round(x, 1)
    return x


def flat_scalars(line):
    """The numbers this project is judged on, as flat top-level scalars of the JSON line (round-5 review: the nested line is
    > 8 KB, the driver keeps top-level scalars and a short tail)."""
    b, c = line.get("build_10m") or {}, _dig(line, "cpu_baseline", "build_10m") or {}
    out = {
        "build_1m_seconds": _dig(line, "build", "seconds"),
        "build_10m_seconds": b.get("seconds"), "build_10m_seconds_device": b.get("seconds_device"),
        "build_10m_seconds_after_device": b.get("seconds_after_device"), "build_10m_tail_groups": b.get("tail_groups"),
        "build_10m_level_by_level_seconds": _dig(b, "level_by_level", "seconds"),
        "build_10m_f32_only_seconds": _dig(b, "f32_only", "seconds"),
        "build_10m_identical": b.get("identical"), "build_10m_screen_fallbacks_frac": b.get("screen_fallbacks_frac"),
        "build_10m_screen8_decided_frac": b.get("screen8_decided_frac"),
        "build_10m_oracle_tree_identical": _dig(b, "oracle_tree", "identical"),
        "build_10m_normal_seconds": _dig(b, "normal", "seconds"), "build_10m_normal_identical": _dig(b, "normal", "identical"),
        "build_10m_clustered_seconds": _dig(b, "clustered", "seconds"),
        "build_10m_clustered_seconds_device": _dig(b, "clustered", "seconds_device"),
        "build_10m_clustered_identical": _dig(b, "clustered", "identical"),
        "build_10m_clustered_screen8_decided_frac": _dig(b, "clustered", "screen8_decided_frac"),
        "build_10m_clustered_screen_fallbacks_frac": _dig(b, "clustered", "screen_fallbacks_frac"),
        "build_10m_clustered_f32_only_seconds": _dig(b, "clustered", "f32_only_seconds"),
        "build_10m_clustered_retries": _dig(b, "clustered", "retries"),
        "build_10m_clustered_dummy_normals": _dig(b, "clustered", "dummy_normals"),
        "share_13_seconds": _dig(b, "share_13", "seconds"), "share_13_seconds_device": _dig(b, "share_13", "seconds_device"),
        "share_13_speedup": _dig(b, "share_13", "speedup_100_trees_over_share"),
        "stream_seconds": _dig(b, "stream", "seconds"), "stream_seconds_after_device": _dig(b, "stream", "seconds_after_device"),
        "cold_total_s": _dig(b, "cold", "total_s"), "cold_staging_gb_per_s": _dig(b, "cold", "staging_gb_per_s"),
        "cold_first_build_s": _dig(b, "cold", "first_build_s"), "cold_reserve_thread_s": _dig(b, "cold", "reserve_thread_s"),
        "cpu_build_10m_seconds": c.get("build_seconds_config_2"), "cpu_build_1m_seconds": _dig(line, "cpu_baseline", "build_seconds_config_1"),
        "rerank_callers_1_qps": _dig(line, "rerank", "callers_1", "queries_per_s"),
        "rerank_callers_1_frac": _dig(line, "rerank", "callers_1", "frac_of_hbm_peak"),
        "rerank_callers_1_binary16_qps": _dig(line, "rerank", "callers_1_binary16", "queries_per_s"),
        "rerank_callers_1_survivors_per_query": _dig(line, "rerank", "callers_1", "survivors_per_query"),
        "rerank_callers_4_qps": _dig(line, "rerank", "callers_4", "queries_per_s"),
        "rerank_f32_qps": _dig(line, "rerank", "callers_1_f32_only", "queries_per_s"),
        "rerank_f32_frac": _dig(line, "rerank", "callers_1_f32_only", "frac_of_hbm_peak"),
        "rerank_one_submission_qps": _dig(line, "rerank", "one_submission", "queries_per_s"),
        "rerank_callers_1_sync_wait_share": None,
        "bq_scan_frac": _dig(line, "bq_scan", "roofline", "frac"),
        "search_clustered_qps": _dig(line, "search", "callers_1", "queries_per_s"),
        "search_distinct_qps": _dig(line, "search", "callers_1_distinct_items", "queries_per_s"),
        "search_filter_half_qps": _dig(line, "search", "callers_1_filter_half", "queries_per_s"),
        "search_nq1_p50_us": _dig(line, "search", "latency", "search_nq_1", "p50_us"),
        "search_nq1_p99_us": _dig(line, "search", "latency", "search_nq_1", "p99_us"),
        "search_nq8_p50_us": _dig(line, "search", "latency", "search_nq_8", "p50_us"),
        "rerank_by_vector_p50_us": _dig(line, "search", "latency", "rerank_by_vector", "p50_us"),
        "search_verified": _dig(line, "search", "verified"),
        "replicate_10m_gb_per_s_total": _dig(line, "replicate_10m", "gb_per_s_total"),
        "build_10m_union_digest": _dig(line, "build_10m_union", "digest"),
        "build_10m_union_identical": _dig(line, "build_10m_union", "identical"),
    }
    ph = _dig(line, "rerank", "callers_1", "seconds_by_phase")
    if ph and ph.get("wall"):
        out["rerank_callers_1_sync_wait_share"] = ph.get("sync_wait", 0.0) / ph["wall"]
    per = line.get("build_10m_per_device") or {}
    for r, v in sorted(per.items(), key=lambda kv: int(kv[0])):
        if len(per) > 1:  # an N > 1 line: every device's own seconds
            out[f"build_10m_device_{r}_seconds"] = v.get("seconds")
            out[f"build_10m_device_{r}_seconds_device"] = v.get("seconds_device")
    return {k: v for k, v in out.items() if v is not None}


def summary_line(line, flat):
    """< 1.9 KB (the driver keeps a 2 KB tail of stdout): the contract keys, `roofline` / `cpu_baseline` cut to their contract fields, the flat scalars rounded."""
    keep = ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype")
    out = {"metric": "distances/sec, Q=1 768-dim cosine scan; build seconds at 10M", "summary": True}
    out.update({k: _r(line.get(k), 6) for k in keep})
    out["data"] = "synthetic"
    out["config"] = {"workload": "1M x 768 cosine Q=1 scan (configs[1])"}
    roof = line.get("roofline") or {}
    out["roofline"] = {k: _r(roof.get(k), 5) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
    cpu = line.get("cpu_baseline") or {}
    if cpu:
        out["cpu_baseline"] = {"value": _r(cpu.get("value"), 4), "unit": cpu.get("unit"), "cores": cpu.get("cores"), "kind": cpu.get("kind"),
                               "sample": "Q=1 scan of all 1M rows, OpenMP"}
    short = {"build_10m": "b10m", "clustered": "clu", "seconds": "s", "identical": "same", "screen": "scr", "fallbacks": "fb",
             "decided": "dec", "callers_": "c", "search": "srch", "rerank": "rr", "device": "dev", "oracle_tree": "otree"}
    for k, v in flat.items():
        kk = k
        for a, b in short.items():
            kk = kk.replace(a, b)
        out[kk] = _r(v)
    # under 1.9 KB whatever a later round adds: the least telling scalars go first (never reached at today's key count)
    drop = ["b10m_union_digest", "cpu_build_1m_s", "rr_one_submission_qps", "srch_filter_half_qps", "stream_s_after_dev",
            "b10m_clu_dummy_normals", "b10m_clu_retries", "b10m_clu_f32_only_s", "cold_reserve_thread_s", "rr_by_vector_p50_us"]
    while len(json.dumps(out, separators=(",", ":"))) > 1900 and drop:
        out.pop(drop.pop(0), None)
    return out


def dry_run_work(args, rank, world, sync, result):
    """Control-path test: no device work, fixed fake durations."""
    from arroy_amd import shard
    my_trees = shard.trees_for_rank(args.trees, rank, world)
    sync.barrier(rank)
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    sync.barrier(rank)
    result["elapsed"] = sync.max(time.perf_counter() - t0, rank)
    result["kernel_ms"] = result["elapsed"] * 1e3 / max(args.steps, 1)
    b = sync.max(0.001 * len(my_trees), rank)
    # the per-device figures of an N > 1 line travel by sync.gather (rank order, every rank gets all of them)
    per_rank = sync.gather({"rank": rank, "trees": len(my_trees)}, rank)
    if rank == 0:
        result["device"] = "dry-run (cpu)"
        result["build"] = {"trees": args.trees, "trees_this_rank": len(my_trees), "seconds": b, "per_rank": per_rank}


def main():
    args = parse_args()
    if args.build_items is None:
        args.build_items = 1_000_000 if args.virtual else 10_000_000
    if args.scan_only:
        scan_only(args)
        return
    if args.cold_child:
        cold_child(args)
        return
    env_world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    result = {}
    cold_pre = None
    if (args.gpus == 1 and env_world <= 1 and not (args.dry_run or args.virtual or args.no_e2e or args.no_build or args.no_build_10m)
            and args.items == N_ITEMS):
        cold_pre = run_cold_child(args)  # BEFORE this process touches the device (cold_child's docstring)
        result["cold_10m"] = cold_pre
    if env_world > 0:
        # one rank per process (python -m torch.distributed.run): the launcher's world must be the --gpus asked for
        world, rank, local_rank = env_world, int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
        if world != args.gpus:
            print(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}", file=sys.stderr)
            sys.exit(2)
        mode = "one process per GPU (torch.distributed)" if world > 1 else "single process"
        sync = RankSync(args, rank, world, local_rank)
        if args.dry_run:
            dry_run_work(args, rank, world, sync, result)
        else:
            import arroy_amd
            if args.virtual:
                local_rank = 0  # every rank on device 0: the N > 1 path of this process model on one GPU
            if arroy_amd.device_count() <= local_rank:
                print(f"bench.py: rank {rank} needs device {local_rank}, {arroy_amd.device_count()} visible", file=sys.stderr)
                sys.exit(3)
            sync.local_rank = local_rank
            sync.torch.cuda.set_device(local_rank)
            device_work(args, rank, world, local_rank, sync, None, result)
            if not args.no_build and not args.no_build_10m and args.items == N_ITEMS:
                build_10m(args, rank, world, local_rank, sync, None, result)
        if not args.dry_run and world > 1:
            for key in ("build_10m_per_device", "build_seconds_per_device", "build_10m_identical_per_device"):
                merged = {}
                for part in sync.gather(result.get(key, {}), rank):
                    merged.update(part)
                if merged:
                    result[key] = merged
        sync.close()
        n_used = world
    else:
        # this process drives all --gpus devices: one host thread per device, dataset staged once and replicated over xGMI
        world, rank = args.gpus, 0
        mode = f"single process, {world} host threads (one per device), dataset replicated device to device" if world > 1 \
            else "single process"
        sync = ThreadSync(args, world)
        if not args.dry_run:
            import arroy_amd
            have = arroy_amd.device_count()
            if have < (1 if args.virtual else world):
                print(f"bench.py: --gpus {world} but only {have} device(s) visible", file=sys.stderr)
                sys.exit(3)
        results = [dict() for _ in range(world)]
        if cold_pre is not None:
            results[0]["cold_10m"] = cold_pre
        errors = []

        def replicas(n_items):
            """configs dataset on device 0 + device-to-device replicas on the others (None per device at N = 1)."""
            if world == 1:
                return [None], None
            from arroy_amd import Dataset, distances
            d0 = Dataset(distances.Cosine, DIMS, n_items, device=0)
            d0.fill_synthetic(SEED, 1, n_items)
            d0.finalize()
            t0 = time.perf_counter()
            out = [d0] + [None] * (world - 1)

            def rep(i):
                out[i] = d0.replicate(0 if args.virtual else i)
            ths = [threading.Thread(target=rep, args=(i,)) for i in range(1, world)]
            [t.start() for t in ths]
            [t.join() for t in ths]
            el = time.perf_counter() - t0
            return out, {"seconds": el, "replicas": world - 1,
                         "gb_per_s_total": (world - 1) * n_items * DIMS * 4 / el / 1e9}

        def run_threads(fn, datasets):
            def body(i):
                try:
                    fn(i, datasets[i])
                except BaseException as e:  # noqa: BLE001 — a failed thread must not leave the others in a barrier
                    errors.append(repr(e))
                    sync.bar.abort()
            ths = [threading.Thread(target=body, args=(i,)) for i in range(world)]
            [t.start() for t in ths]
            [t.join() for t in ths]
            if errors:
                print("bench.py: device thread failed: " + "; ".join(errors), file=sys.stderr)
                sys.exit(4)

        if args.dry_run:
            run_threads(lambda i, _d: dry_run_work(args, i, world, sync, results[i]), [None] * world)
        else:
            dev_of = (lambda i: 0) if args.virtual else (lambda i: i)
            dsets, rep = replicas(args.items)
            run_threads(lambda i, d: device_work(args, i, world, dev_of(i), sync, d, results[i]), dsets)
            if rep:
                results[0]["replicate_1m"] = rep
            if not args.no_build and not args.no_build_10m and args.items == N_ITEMS:
                dsets, rep = replicas(args.build_items)
                run_threads(lambda i, d: build_10m(args, i, world, dev_of(i), sync, d, results[i]), dsets)
                if rep:
                    results[0]["replicate_10m"] = rep
        result = results[0]
        for key in ("build_seconds_per_device",):
            merged = {}
            for r in results:
                merged.update(r.get(key, {}))
            if merged:
                result[key] = merged
        per = {}
        for r in results:
            for k, v in r.get("build_10m_seconds_per_device", {}).items():
                per.setdefault(k, {}).update(v)
        if per:
            result["build_10m_seconds_per_device"] = per
        same = {}
        for r in results:
            same.update(r.get("build_10m_identical_per_device", {}))
        if same:
            result["build_10m_identical_per_device"] = same
        perdev = {}
        for r in results:
            perdev.update(r.get("build_10m_per_device", {}))
        if perdev:
            result["build_10m_per_device"] = perdev
        n_used = world

    srch = None
    if rank == 0:
        n = args.items
        elapsed, kernel_ms = result["elapsed"], result["kernel_ms"]
        ms_per_step = elapsed * 1e3 / args.steps
        value = n_used * n * args.steps / elapsed
        achieved = n * BYTES_PER_DISTANCE / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_src = (args.traffic_bytes, "--traffic-bytes") if args.traffic_bytes else \
            (measured_traffic(n) if not args.dry_run else (None, "dry-run"))
        stored = {"traffic": traffic, "source": traffic_src}
        if not args.dry_run and not args.no_live_pmc and not args.traffic_bytes and n_used == 1 and env_world <= 1:
            # the counters of THIS run (round-3 review: a stored figure is not a measurement of the driver's run); the stored,
            # hash-stamped figure stays next to it and is what is quoted if the child runs cannot be made
            try:  # the children are other processes: give them the HBM this one only keeps cached
                from arroy_amd import _lib as ahlib
                ahlib.device_cache_trim()
            except Exception:  # noqa: BLE001
                pass
            live, how = live_traffic(n)
            if live is not None:
                traffic, traffic_src = live, how
            else:
                stored["live_attempt"] = how
        read_gbs = result.get("read_gbs")
        line = {
            "metric": "distances/sec, Q=1 batched 768-dim cosine scan (GB/s vs HBM roofline in `roofline`); tree-build "
                      "seconds at 10M vectors in `build_10m`",
            "value": value, "unit": "distances/s", "n_gpus": n_used, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic uniform[-1,1) (counter-based generator, seed 42), generated in HBM",
            "config": {"workload": f"{n}x{DIMS} cosine Q=1 distance scan, one replica per GPU (BASELINE configs[1])",
                       "items": n, "dims": DIMS, "metric": "cosine", "device": result.get("device"), "launch": mode},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_stored": stored,
                         "kernel": "ah::k_distances_f32<COSINE,false>", "kernel_ms": kernel_ms,
                         "kernel_source_sha16": scan_source_hash(),
                         "measured_d2d_copy_gb_per_s": result.get("copy_gbs"), "measured_read_only_gb_per_s": read_gbs,
                         "frac_of_measured_read_ceiling": achieved / read_gbs if read_gbs else None,
                         "algorithmic_bytes_per_launch": n * BYTES_PER_DISTANCE},
            "cpu_baseline": dict(result["cpu"], build_10m=result.get("cpu_10m")) if result.get("cpu") else None,
            "build": result.get("build"),
            "build_10m": result.get("build_10m"),
            # runtime switches this process ran under (GPU_MAX_HW_QUEUES: set at the top of this file unless the caller set it)
            "env": {k: os.environ[k] for k in ("GPU_MAX_HW_QUEUES", "HSA_ENABLE_IPC_MODE_LEGACY") if k in os.environ},
        }
        if args.virtual:
            line["config"]["virtual_devices"] = (f"{n_used} ranks / device threads, all on device 0: the N > 1 code path on one GPU — "
                                                 "a correctness and diagnosability run, not a throughput figure")
        for key in ("build_seconds_per_device", "build_10m_seconds_per_device", "build_10m_identical_per_device",
                    "build_10m_per_device", "build_10m_union", "replicate_1m", "replicate_10m"):
            if result.get(key):
                line[key] = result[key]
        extra = result.get("extra") or {}
        for key in ("rerank", "bq_scan", "search"):
            if key in extra:
                line[key] = extra.pop(key)
        if extra:
            line["extra"] = extra
        if not args.dry_run and not args.no_live_pmc and n_used == 1 and env_world <= 1:
            # the other two roofline entries: their kernels' HBM traffic measured in this run as well
            for key in ("rerank", "bq_scan"):
                roof = (line.get(key) or {}).get("roofline")
                if roof is None:
                    continue
                live, how = live_traffic(N_ITEMS, steps=3, which=key)
                roof["traffic_stored"] = {"traffic": roof.get("traffic"), "source": roof.get("traffic_source")}
                if live is not None:
                    roof["traffic"], roof["traffic_source"] = live, how
                else:
                    roof["traffic_stored"]["live_attempt"] = how
        srch = line.get("search")
        flat = flat_scalars(line)
        # top-level scalars — what the driver's record keeps of a line this long (`parsed`) — appended LAST, so that a tail of
        # the one stdout line shows them too (round-5 review: share_13 / stream / cold were cut off a > 8 KB line)
        line.update(flat)
        print(json.dumps(line), flush=True)
        # the same in short on stderr (stdout stays ONE JSON line, as the contract says): the headline keys, `roofline` and
        # `cpu_baseline` abridged, every flat scalar, under 1.9 KB
        print(json.dumps(summary_line(line, flat), separators=(",", ":")), file=sys.stderr, flush=True)
    # a screened forest that differs from the f32-only forest is a wrong result, not a slow one
    same = result.get("build_10m_identical_per_device") or {}
    if same and not all(same.values()):
        print(f"bench.py: screened and f32-only forests differ: {same}", file=sys.stderr)
        sys.exit(5)
    if result.get("build_10m_union_ok") is False:
        print(f"bench.py: the union of the devices' shares differs from the one-device build: {result.get('build_10m_union')}",
              file=sys.stderr)
        sys.exit(6)
    # ... and so is a search whose answers differ from the oracle's
    if srch is not None and srch.get("verified") is False:
        print(f"bench.py: on-device search differs from the oracle: {srch.get('mismatches')}", file=sys.stderr)
        sys.exit(5)


if __name__ == "__main__":
    main()
