"""GPU parity of EVERY margin-kernel family of the forest build (src/writer.rs:1193-1233) against the CPU oracle.

`ah_build_options.margin_mode` pins one kernel family for all levels where it is legal: node-major, row-major with
2 / 4 / 8 / 16 trees per pass, row-major with the normals of 8 / 16 trees resident in LDS, the dense screen on the
matrix units (AH_MARGIN_DENSE_MFMA) — each with and without the certified binary16 screen (AH_MARGIN_EXACT_ONLY).  Whole forests must equal the oracle's node for node at the
dimensions the headline builds use (768 / 1536: the 8-unrolled main loops) and at short rows (128 / 256)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import test_gpu_parity as P
from conftest import ROOT
from oracle import oracle as O
from test_gpu_parity import check_forest_valid, make_data

pytestmark = pytest.mark.gpu

from arroy_amd import _lib  # noqa: E402
from arroy_amd import distances as D  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def _imports():
    import arroy_amd
    assert arroy_amd.device_count() >= 1, "no GPU visible: these tests must run on an MI355X"
    P.D, P.O = D, O  # make_data / check_forest_valid live in test_gpu_parity and use its lazily imported modules

MODES = [("auto", _lib.MARGIN_AUTO), ("node_major", _lib.MARGIN_NODE_MAJOR), ("rows_tc2", 2), ("rows_tc4", 4), ("rows_tc8", 8),
         ("rows_tc16", 16), ("rows_lds_tc8", 0x108), ("rows_lds_tc16", 0x110), ("dense_mfma", 0x200)]
SHAPES = [("cosine768", D.Cosine, 768, 24_000), ("dot1536", D.DotProduct, 1536, 12_000), ("euclid128", D.Euclidean, 128, 30_000),
          ("manhattan256", D.Manhattan, 256, 20_000)]
N_TREES = 16
_cache = {}


def shape_fixture(name):
    """Dataset + oracle trees of a shape, built once per module run (the oracle is the slow side)."""
    if name not in _cache:
        _, cls, dims, n = next(s for s in SHAPES if s[0] == name)
        ds, oracle, vecs, ids = make_data(cls, n, dims, seed=len(name) * 1000 + dims)
        seeds = [int(x) for x in np.random.default_rng(dims).integers(0, 2**63, N_TREES)]
        ref = [oracle.build_tree(0, s).canonical() for s in seeds]
        evals = sum(oracle.build_tree(0, s).margin_evals for s in seeds[:2])
        _cache[name] = (ds, seeds, ref, evals, n)
    return _cache[name]


@pytest.mark.parametrize("exact_only", [False, True], ids=["screened", "f32_only"])
@pytest.mark.parametrize("mode", MODES, ids=[m[0] for m in MODES])
@pytest.mark.parametrize("shape", [s[0] for s in SHAPES])
def test_forest_equals_oracle_in_every_margin_mode(shape, mode, exact_only):
    ds, seeds, ref, _, n = shape_fixture(shape)
    flag = _lib.MARGIN_EXACT_ONLY if exact_only else 0
    forest = ds.build_forest(seeds, margin_mode=mode[1] | flag)
    check_forest_valid(forest, n)
    for t in range(N_TREES):
        assert forest.canonical(t) == ref[t], f"tree {t} differs from the oracle in mode {mode[0]}"
    st = forest.stats
    if mode[1] in _lib.MODE_LAUNCH_INDEX:  # the pinned kernel family really ran
        assert st["margin_mode_launches"][_lib.MODE_LAUNCH_INDEX[mode[1]]] > 0, st["margin_mode_launches"]
    if mode[1] == _lib.MARGIN_DENSE_MFMA and not exact_only:  # the matrix-unit screen ran (it needs the screen)
        assert st["dense_launches"] > 0 and st["dense_columns"] >= st["dense_launches"], st
    elif mode[1] != _lib.MARGIN_AUTO:
        assert st["dense_launches"] == 0
    if mode[1] == _lib.MARGIN_NODE_MAJOR or (mode[1] == _lib.MARGIN_DENSE_MFMA and exact_only):
        assert sum(st["margin_mode_launches"][1:7]) == 0
    if exact_only:
        assert st["screened_launches"] == 0 and st["screen_fallbacks"] == 0
    else:
        assert st["screened_launches"] > 0
        assert st["screen_fallbacks"] < 0.2 * st["margin_evaluations"]  # the screen decides the bulk of the pairs
    assert st["screen_violations"] == 0
    forest.close()


def test_margin_mode_is_validated():
    ds, seeds, *_ = shape_fixture("euclid128")
    with pytest.raises(_lib.ArroyHipError):
        ds.build_forest(seeds[:2], margin_mode=3)
    with pytest.raises(_lib.ArroyHipError):
        ds.build_forest(seeds[:2], margin_mode=0x2000)


VERIFY_SCRIPT = r"""
import sys, json
import numpy as np
sys.path.insert(0, sys.argv[1])
sys.path.insert(0, sys.argv[1] + "/tests")
from arroy_amd import Dataset, distances as D
out = []
rng = np.random.default_rng(5)
for cls, dims, n in [(D.Cosine, 768, 20000), (D.Euclidean, 96, 20000), (D.Manhattan, 200, 12000), (D.DotProduct, 512, 12000),
                     (D.Cosine, 1536, 6000)]:
    for scale, shift in [(1.0, 0.0), (1e-3, 0.0), (300.0, 0.0), (1e5, 0.0), (1e-7, 0.0), (1.0, 5.0)]:
        vecs = (rng.standard_normal((n, dims)) * scale + shift).astype(np.float32)
        vecs[7] = 0.0
        vecs[9] = vecs[8]
        ds = Dataset(cls, dims, n)
        ds.upload_vectors(np.arange(n, dtype=np.uint32), vecs)
        if cls is D.DotProduct:
            ds.preprocess_dot()
        ds.finalize()
        for mode in (0, 1, 4, 16, 0x110, 0x200):
            f = ds.build_forest([1, 2, 3, 4, 5, 6, 7, 8], margin_mode=mode)
            st = f.stats
            out.append({"metric": cls.name, "dims": dims, "scale": scale, "shift": shift, "mode": mode,
                        "evals": st["margin_evaluations"], "fallbacks": st["screen_fallbacks"],
                        "violations": st["screen_violations"], "screened": st["screened_launches"],
                        "dense": st["dense_launches"]})
            f.close()
        ds.close()
print(json.dumps(out))
"""


def test_screen_bound_holds_for_every_pair():
    """AH_SCREEN_VERIFY=1: the screened kernels also evaluate the reference arithmetic for EVERY (item, node) pair and
    count the pairs a screen stage (int8, binary16, the dense MFMA product) decided differently.  That count must be 0 — for
    data at scales where binary16 is exact enough, where it underflows (1e-7) and where it overflows (1e5: inf)."""
    import json
    env = dict(os.environ, AH_SCREEN_VERIFY="1", AH_SCREEN8="1")  # int8 first stage kept even where it decides little
    res = subprocess.run([sys.executable, "-c", VERIFY_SCRIPT, ROOT], capture_output=True, text=True, env=env, timeout=1500)
    assert res.returncode == 0, res.stderr[-2000:]
    rows = json.loads(res.stdout.strip().splitlines()[-1])
    assert len(rows) == 5 * 6 * 6
    assert all(r["dense"] > 0 for r in rows if r["mode"] == 0x200)
    for r in rows:
        assert r["violations"] == 0, r
        assert r["screened"] > 0, r
    # unit-scale data: the screen decides almost everything; far outside the binary16 range: nothing, and that is fine
    unit = [r for r in rows if r["scale"] == 1.0 and r["shift"] == 0.0 and r["metric"] == "cosine" and r["dims"] == 768]
    assert all(r["fallbacks"] < 0.05 * r["evals"] for r in unit), unit
    # binary16 cannot hold values of 1e5 x N(0, 1) (rows overflow -> measured error inf): the row-order passes and the
    # dense product decide nothing there, and that is fine.  The node-major screen has an int8 first stage whose single
    # scale follows the data, so it still decides most pairs (DotProduct has no int8 stage: the margin needs the row's
    # header anyway).
    huge = [r for r in rows if r["scale"] == 1e5]
    assert all(r["fallbacks"] >= 0.99 * r["evals"] for r in huge if r["metric"] == "dot-product"), huge[:3]
    assert all(r["fallbacks"] < 0.9 * r["evals"] for r in huge if r["mode"] == 1 and r["metric"] != "dot-product"), huge[:6]


def test_baseline_config_2_two_full_size_trees_equal_oracle():
    """BASELINE configs[1] at full size: 1M x 768 cosine, n_trees = 50, default (AUTO, screened) build — the kernels and
    the per-level choices of the headline number.  Trees 0 and 37 are compared with the CPU oracle node for node, normals
    and descendant lists bit for bit (one oracle tree streams ~34 GB; the two run on separate host threads)."""
    from concurrent.futures import ThreadPoolExecutor

    from arroy_amd import Dataset, shard
    n, dims, trees = 1_000_000, 768, 50
    ds = Dataset(D.Cosine, dims, n)
    ds.fill_synthetic(42, 1, n)
    ds.finalize()
    seeds = shard.tree_seeds(42, range(trees))
    forest = ds.build_forest(seeds)
    assert forest.n_trees == trees and forest.stats["screen_violations"] == 0
    vecs = O.synth(42, 1, n, dims)
    oracle = O.Data(O.COSINE, vecs)
    picks = [0, 37]
    with ThreadPoolExecutor(2) as pool:
        ref = list(pool.map(lambda t: oracle.build_tree(0, seeds[t]), picks))
    for t, r in zip(picks, ref):
        assert forest.canonical(t) == r.canonical(), f"tree {t} of the 1M x 768 build differs from the oracle"
    # the same two trees from the f32-only build of the same seeds
    f2 = ds.build_forest([seeds[t] for t in picks], margin_mode=_lib.MARGIN_EXACT_ONLY)
    for i, t in enumerate(picks):
        assert f2.canonical(i) == forest.canonical(t)
    forest.close()
    f2.close()
    ds.close()


@pytest.mark.parametrize("metric", [D.Cosine, D.Euclidean])
def test_very_long_vectors(metric):
    """12 000 dimensions: the node's normals (f32 + binary16) no longer fit the default dynamic-LDS limit of the
    node-major kernels, and the two-means of `create_split` uses 144 KB of LDS.  Same forests as the oracle."""
    dims, n = 12_000, 700
    ds, oracle, vecs, ids = make_data(metric, n, dims, seed=5)
    for mode in (0, _lib.MARGIN_NODE_MAJOR, _lib.MARGIN_NODE_MAJOR | _lib.MARGIN_EXACT_ONLY, 4):
        forest = ds.build_forest([3, 4, 5], split_after=60, margin_mode=mode)
        for t, seed in enumerate([3, 4, 5]):
            assert forest.canonical(t) == oracle.build_tree(60, seed).canonical(), (mode, t)
        forest.close()
    ds.close()


def test_baseline_config_3_split_sides_equal_oracle_at_10m():
    """BASELINE configs[2] at full size (10M x 768 cosine, default build, 20 of the 100 trees so that every row-major
    group size occurs): the children the forest recorded for the ROOT of two trees and for sampled split nodes at every
    depth equal the sides the ORACLE computes for the recorded normal (`D::side`, src/writer.rs:1201-1207) over the node's
    items — the margin kernels checked at full size against the reference arithmetic, not against another GPU kernel."""
    from arroy_amd import Dataset, shard
    from test_gpu_parity import subtree_items
    n, dims, trees = 10_000_000, 768, 20
    ds = Dataset(D.Cosine, dims, n)
    ds.fill_synthetic(42, 1, n)
    ds.finalize()
    f = ds.build_forest(shard.tree_seeds(42, range(trees)))
    launches = f.stats["margin_mode_launches"]
    # node-major (deep levels), row-major (middle), and the top levels: dense MFMA screen (or LDS-resident row-major groups)
    assert launches[0] > 0 and sum(launches[1:5]) > 0 and launches[5] + launches[6] + f.stats["dense_launches"] > 0, launches
    assert f.stats["dense_launches"] > 0, f.stats
    assert f.stats["screened_launches"] > 0 and f.stats["screen_violations"] == 0
    vecs = O.synth(42, 1, n, dims)
    oracle = O.Data(O.COSINE, vecs)
    nodes = f.nodes
    rng = np.random.default_rng(3)
    checked = {}
    for t in (0, 17):
        root = int(f.roots[t])
        picks = [root]
        of_tree = np.flatnonzero((nodes["kind"] == 2) & (nodes["tree"] == t) & (nodes["has_normal"] == 1))
        by_depth = {}
        for i in of_tree[rng.permutation(of_tree.size)]:
            by_depth.setdefault(int(nodes[i]["depth"]), []).append(int(i))
        for depth, cand in sorted(by_depth.items()):
            if depth > 0:
                picks += cand[:2]
        for i in picks:
            nd = nodes[i]
            left, right = subtree_items(f, nd["left"]), subtree_items(f, nd["right"])
            rows = np.sort(np.concatenate([left, right]))  # ids are 0..n-1: row == id
            hdr, vec = f.normal_of(i)
            sides, n_left, _ = oracle.split_sides(vec, hdr, rows)
            assert n_left == left.size, (t, i, int(nd["depth"]))
            assert np.array_equal(rows[sides == 0], left) and np.array_equal(rows[sides == 1], right), (t, i)
            checked[int(nd["depth"])] = checked.get(int(nd["depth"]), 0) + 1
    assert checked[0] == 2 and len(checked) >= 13, checked
    f.close()
    ds.close()


@pytest.mark.parametrize("shape", [(D.Cosine, 768, 30_000, 100), (D.Euclidean, 128, 40_000, 64), (D.DotProduct, 256, 30_000, 90)],
                         ids=["cosine768", "euclid128", "dot256"])
def test_dense_levels_of_several_tiles_and_both_ways_of_reading_the_exact_pairs_equal_the_oracle(shape):
    """20 trees make dense levels of 320 and 640 columns (several 256-column tiles, the last one partly padding).  Round 6:
    `k_forest_exact_pairs` asks for every line of a pair's row and normal before its first multiply-add (`AH_EXACT_WIDE`, the same
    chains in the same order as the streamed reduction of rounds 2-5).  Whole forests against the oracle with the screen's
    self-check on, and the same bits either way."""
    cls, dims, n, split_after = shape
    ds, oracle, _vecs, _ids = make_data(cls, n, dims, seed=dims + 17)
    seeds = [int(x) for x in np.random.default_rng(dims + 1).integers(0, 2**63, 20)]
    ref = [oracle.build_tree(split_after, s).canonical() for s in seeds]
    digests = set()
    for knobs in (dict(), dict(AH_EXACT_WIDE=0)):
        with _lib.tuning(AH_SCREEN_VERIFY=1, **knobs):
            forest = ds.build_forest(seeds, split_after=split_after, margin_mode=_lib.MARGIN_DENSE_MFMA)
        st = forest.stats
        assert st["dense_launches"] >= 6 and st["screen_violations"] == 0, (knobs, st)
        for t in range(len(seeds)):
            assert forest.canonical(t) == ref[t], f"tree {t} differs from the oracle with {knobs}"
        digests.add(forest.digest()[0])
        forest.close()
    assert len(digests) == 1
    ds.close()
